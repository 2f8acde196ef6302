// Diagnostic (not part of the product): can gfx950 fetch LESS than a whole 128-byte line for a 16-byte gather?
//
// frames_ring_kernel gathers one 16-byte window per touched atom of a 60 000-byte AoS frame (C4: one C-alpha per
// 192 bytes) and is at the HBM roof *for whole 128-byte lines* (43.9 KB fetched per frame for 5.9 KB needed).  This
// program reads ONE 16-byte (or 12 / 4-byte) piece every STRIDE bytes of a buffer far larger than the Infinity Cache
// with each load form the kernel could use, and prints the time per access; run under rocprofv3 --pmc with
// TCC_EA0_RDREQ_sum / _32B / _64B / _128B / TCC_BUBBLE_sum it shows the size of the requests L2 sent to memory.
// Every variant is its own kernel NAME (variant and stride in the name: `sub<V,STRIDE>`), so counters can be read per row.
//
//   V 0  global_load_dwordx4 -> registers, default policy        V 4  global_load_lds_dwordx4, default policy
//   V 1  global_load_dwordx4 sc0 sc1 nt -> registers             V 5  global_load_lds_dwordx4 nt
//   V 2  global_load_dwordx3 -> registers                        V 6  global_load_lds_dwordx4 sc0 sc1 nt
//   V 3  global_load_dword   -> registers                        V 7  buffer_load_dwordx4 offen sc0 sc1 nt -> registers
//   V 8  global_load_dwordx4 nt -> registers                     V 9  buffer_load_dwordx4 ... lds sc0 sc1 nt  (LDS-DMA, buffer form)
//
//   hipcc --offload-arch=gfx950 -O3 -o subline subline.hip && ./subline [GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BATCH = 8;          // loads in flight per wave between two waits
constexpr int WAVES_PER_BLOCK = 8;
constexpr int BLOCKS = 256 * 2;   // 16 waves per CU, 128 loads (of 64 lanes) in flight per CU

template <int V, int STRIDE>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void sub(const unsigned char* __restrict__ buf, long n_access, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[WAVES_PER_BLOCK * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long gw = (long)blockIdx.x * WAVES_PER_BLOCK + wave, nw = (long)gridDim.x * WAVES_PER_BLOCK;
    const unsigned voff = (unsigned)lane * STRIDE;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)(lds + wave * 1024));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long n_groups = n_access / 64;   // one group = one wave-instruction = 64 accesses
    // consecutive groups go to consecutive waves: at any moment the chip reads one contiguous region, as the ring kernel does
    for (long g0 = gw; g0 < n_groups; g0 += nw * BATCH) {
        f32x4 r[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            long g = g0 + (long)b * nw;
            if (g >= n_groups) g = gw;       // wave-uniform: re-read an early group instead of branching around the asm
            const unsigned char* base = buf + g * (64l * STRIDE);
            if constexpr (V == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[b]) : "v"(voff), "s"(base) : "memory");
            if constexpr (V == 1) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1 nt" : "=v"(r[b]) : "v"(voff), "s"(base) : "memory");
            if constexpr (V == 8) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r[b]) : "v"(voff), "s"(base) : "memory");
            if constexpr (V == 2) {
                float __attribute__((ext_vector_type(3))) t;
                asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(t) : "v"(voff), "s"(base) : "memory");
                r[b] = (f32x4){t[0], t[1], t[2], 0.f};
            }
            if constexpr (V == 3) {
                float t;
                asm volatile("global_load_dword %0, %1, %2" : "=v"(t) : "v"(voff), "s"(base) : "memory");
                r[b] = (f32x4){t, 0.f, 0.f, 0.f};
            }
            if constexpr (V == 4 || V == 5 || V == 6) {
                unsigned keep;
                if constexpr (V == 4)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_base) : "memory");
                if constexpr (V == 5)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_base) : "memory");
                if constexpr (V == 6)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc0 sc1 nt\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_base) : "memory");
                r[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (V == 7 || V == 9) {
                // raw buffer resource over the group's window: base, stride 0, num_records = 64 STRIDE, dword3 = raw 32-bit format
                const unsigned long long ba = (unsigned long long)base;
                const unsigned r0 = (unsigned)ba, r1 = (unsigned)(ba >> 32) & 0xffffu, r2 = 64u * STRIDE, r3 = 0x00020000u;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 rs = {r0, r1, r2, r3};
                rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
                if constexpr (V == 7) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc0 sc1 nt" : "=v"(r[b]) : "v"(voff), "s"(rs) : "memory");
                if constexpr (V == 9) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc0 sc1 nt lds\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_base) : "memory");
                    r[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            // the asm outputs are only valid after the wait: tie them to it
            asm volatile("" : "+v"(r[b]));
            acc += r[b];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) sink[gw * 64 + lane] = acc[0];   // never true: keeps the loads
    if (V >= 4 && V != 7 && sink == nullptr) sink[0] = ((float*)lds)[lane];
}

template <int V, int STRIDE>
static void run(const char* what, const unsigned char* buf, size_t bytes, float* sink) {
    const long n_access = (long)(bytes / STRIDE) / 64 * 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((sub<V, STRIDE>), dim3(BLOCKS), dim3(WAVES_PER_BLOCK * 64), 0, 0, buf, n_access, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (rep) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double t = ms[ms.size() / 2] * 1e-3;
    const int piece = V == 2 ? 12 : V == 3 ? 4 : 16;
    const double lines = STRIDE >= 128 ? (double)n_access : (double)bytes / 128.0;
    printf("V%d %-44s stride %4d  %8.3f ms  %7.2f G access/s  useful %6.2f TB/s  whole-line %5.2f TB/s  half-line(64B) %5.2f TB/s\n", V, what,
           STRIDE, t * 1e3, n_access / t * 1e-9, n_access * (double)piece / t * 1e-12, lines * 128.0 / t * 1e-12,
           (STRIDE >= 64 ? (double)n_access : (double)bytes / 64.0) * 64.0 / t * 1e-12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int V>
static void run_strides(const char* what, const unsigned char* buf, size_t bytes, float* sink) {
    run<V, 16>(what, buf, bytes, sink);     // dense: every byte (piece 16) - the whole-line reference
    run<V, 64>(what, buf, bytes, sink);     // one piece per 64-byte half line: every line touched twice
    run<V, 128>(what, buf, bytes, sink);    // one piece per line
    run<V, 192>(what, buf, bytes, sink);    // the C4 pattern: one C-alpha per 16 atoms
    run<V, 256>(what, buf, bytes, sink);    // every second line
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 3.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / (64 * 768) * (64 * 768);   // a multiple of 64 x every stride
    unsigned char* buf; float* sink;
    if (hipMalloc(&buf, bytes + 4096) != hipSuccess || hipMalloc(&sink, (size_t)BLOCKS * WAVES_PER_BLOCK * 64 * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes + 4096);
    hipDeviceSynchronize();
    printf("buffer %.2f GiB, %d blocks x %d waves, %d loads in flight per wave\n", bytes / 1073741824.0, BLOCKS, WAVES_PER_BLOCK, BATCH);
    run_strides<0>("global_load_dwordx4", buf, bytes, sink);
    run_strides<1>("global_load_dwordx4 sc0 sc1 nt", buf, bytes, sink);
    run_strides<8>("global_load_dwordx4 nt", buf, bytes, sink);
    run_strides<2>("global_load_dwordx3", buf, bytes, sink);
    run_strides<3>("global_load_dword", buf, bytes, sink);
    run_strides<4>("global_load_lds_dwordx4", buf, bytes, sink);
    run_strides<5>("global_load_lds_dwordx4 nt", buf, bytes, sink);
    run_strides<6>("global_load_lds_dwordx4 sc0 sc1 nt", buf, bytes, sink);
    run_strides<7>("buffer_load_dwordx4 offen sc0 sc1 nt", buf, bytes, sink);
    run_strides<9>("buffer_load_dwordx4 offen sc0 sc1 nt lds", buf, bytes, sink);
    hipFree(buf); hipFree(sink);
    return 0;
}
