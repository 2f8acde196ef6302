// Microbenchmark (diagnostic, not part of the product): HBM read bandwidth of the two ways a frame tile can be
// fetched on gfx950 - ordinary 16-byte vector loads and LDS-DMA (global_load_lds_dwordx4, what the lane kernels
// use) - over buffers large enough and rotated so that nothing comes from the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/read_bw.hip -o tools/micro/read_bw && tools/micro/read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

// tile = TILE bytes per wave per step; persistent waves, like the lane kernels
__device__ unsigned long long g_issue[2]; // [0] cycles spent issuing, [1] tiles
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int TILE>
__global__ __launch_bounds__(128) void read_vec(const unsigned char* __restrict__ x, float* out, long n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned long long issue = 0, tiles = 0;
    for (long t = (long)blockIdx.x * 2 + wave; t < n_tiles; t += (long)gridDim.x * 2) {
        const f32x4* src = (const f32x4*)(x + t * TILE);
        f32x4 r[TILE / 1024];
        const unsigned long long t0 = stamp();
#pragma unroll
        for (int c = 0; c < TILE / 1024; ++c) r[c] = __builtin_nontemporal_load(src + c * 64 + lane);
        asm volatile("" ::: "memory");
        issue += stamp() - t0; ++tiles;
#pragma unroll
        for (int c = 0; c < TILE / 1024; ++c) acc += r[c];
    }
    if (lane == 0) { atomicAdd(&g_issue[0], issue); atomicAdd(&g_issue[1], tiles); }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

template <int TILE>
__global__ __launch_bounds__(128) void read_dma(const unsigned char* __restrict__ x, float* out, long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* tile = smem + wave * TILE;
    float acc = 0.f;
    unsigned long long issue = 0, tiles = 0;
    for (long t = (long)blockIdx.x * 2 + wave; t < n_tiles; t += (long)gridDim.x * 2) {
        const unsigned char* src = x + t * TILE;
        const unsigned long long t0 = stamp();
#pragma unroll
        for (int c = 0; c < TILE / 1024; ++c)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (c * 64 + lane) * 16), (lptr_t)(tile + c * 1024), 16, 0, 0);
        issue += stamp() - t0; ++tiles;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += ((const float*)tile)[lane * (TILE / 256)];   // one read per lane: the tile is "consumed"
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) { atomicAdd(&g_issue[0], issue); atomicAdd(&g_issue[1], tiles); }
    if (acc == 12345.f) out[0] = 1.f;
}

int main() {
    constexpr int TILE = 16384;                 // ~ the 22-atom tile (16896 B)
    const long bytes = 272l << 20, n_tiles = bytes / TILE;
    const int nbuf = 5;
    std::vector<unsigned char*> bufs(nbuf);
    for (auto& b : bufs) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 1, bytes)); }
    float* out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind)
        for (int bpc : {2, 4, 8}) {
            const int grid = 256 * bpc;
            if (kind == 1 && bpc * 2 * TILE > 160 * 1024) continue;
            for (int rep = 0; rep < 2; ++rep) {
                unsigned long long z[2] = {0, 0};
                CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_issue), z, sizeof(z)));
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < 20; ++i) {
                    if (kind == 0) hipLaunchKernelGGL(read_vec<TILE>, dim3(grid), dim3(128), 0, 0, bufs[i % nbuf], out, n_tiles);
                    else hipLaunchKernelGGL(read_dma<TILE>, dim3(grid), dim3(128), 2 * TILE, 0, bufs[i % nbuf], out, n_tiles);
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_issue), sizeof(z)));
                if (rep == 1) printf("%s  blocks/CU %d (waves/CU %d): %.1f us per %ld MiB = %.2f TB/s; issuing a tile's %d loads blocks the wave %.0f ticks of s_memtime (tile period %.0f)\n",
                                     kind == 0 ? "vector loads" : "LDS-DMA     ", bpc, 2 * bpc, ms / 20 * 1e3, bytes >> 20,
                                     bytes / (ms / 20 * 1e-3) / 1e12, TILE / 1024, (double)z[0] / z[1],
                                     ms / 20 * 1e-3 * 1e8 / ((double)n_tiles / (grid * 2)));
            }
        }
    return 0;
}
