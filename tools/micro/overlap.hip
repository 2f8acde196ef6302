// Microbenchmark (diagnostic): a per-tile pipeline shaped like the lane kernel - prefetch the next 9 KB tile, then
// ALU work on the current one - with the prefetch issued as LDS-DMA or as vector loads (registers, written to LDS
// when they have arrived).  Reports the time with the memory side only, the ALU side only, and both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
constexpr int NW = 9, FRAME = 264, TILE_SRC = 64 * FRAME, TILE_LDS = NW * 1024;

__device__ __forceinline__ int slot_source(int i, int lane) {
    const int s = 64 * i + lane, frame = s / NW, k = s - frame * NW;
    return frame * FRAME + 16 * k + (k >= 5 ? 32 : 0) + 12;   // some 16-byte windows of the frame
}

template <int KIND, bool MEM, bool ALU>   // KIND 0: LDS-DMA, 1: vector loads
__global__ __launch_bounds__(128) void pipe(const unsigned char* __restrict__ x, float* out, long n_tiles, int work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* tile = smem + wave * TILE_LDS;
    int goff[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) goff[i] = slot_source(i, lane);
    const long t0 = (long)blockIdx.x * 2 + wave, step = (long)gridDim.x * 2;
    f32x4 r[NW];
    float acc[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    auto issue = [&](long t) {
        const unsigned char* src = x + t * TILE_SRC;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (KIND == 0) __builtin_amdgcn_global_load_lds((gptr_t)(src + goff[i]), (lptr_t)(tile + i * 1024), 16, 0, 0);
            else r[i] = *(const f32x4*)(src + goff[i]);
        }
    };
    if (MEM && t0 < n_tiles) issue(t0);
    for (long t = t0; t < n_tiles; t += step) {
        f32x4 w[NW];
        if (MEM) {
            if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < NW; ++i) *(f32x4*)(tile + i * 1024 + lane * 16) = r[i];   // waits for the loads
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) w[k] = ((const f32x4*)tile)[lane * NW + k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + step < n_tiles) issue(t + step);
#pragma unroll
            for (int k = 0; k < NW; ++k) acc[k & 7] += w[k][0] + w[k][3];
        }
        if (ALU) {
            for (int it = 0; it < work; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(acc[j], 1.0001f, 0.5f);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j];
    if (s == 12345.f) out[0] = s;
}

template <int KIND, bool MEM, bool ALU>
float run(const std::vector<unsigned char*>& bufs, float* out, long n_tiles, int work) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL((pipe<KIND, MEM, ALU>), dim3(1024), dim3(128), 2 * TILE_LDS, 0, bufs[i % bufs.size()], out, n_tiles, work);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms / 10 < best ? ms / 10 : best;
    }
    return best * 1e3f;
}

int main() {
    const long n_tiles = 16384;   // 1 M frames of 264 B
    std::vector<unsigned char*> bufs(5);
    for (auto& b : bufs) { CHECK(hipMalloc(&b, n_tiles * TILE_SRC + 4096)); CHECK(hipMemset(b, 0, n_tiles * TILE_SRC + 4096)); }
    float* out; CHECK(hipMalloc(&out, 4));
    for (int work : {200, 400, 600}) {
        printf("ALU work %d: alu only %.1f us | DMA: mem only %.1f, both %.1f | vector loads: mem only %.1f, both %.1f\n", work,
               run<0, false, true>(bufs, out, n_tiles, work), run<0, true, false>(bufs, out, n_tiles, work), run<0, true, true>(bufs, out, n_tiles, work),
               run<1, true, false>(bufs, out, n_tiles, work), run<1, true, true>(bufs, out, n_tiles, work));
    }
    return 0;
}
