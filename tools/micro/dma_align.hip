// Microbenchmark / probe (diagnostic): does global_load_lds_dwordx4 need a 16-byte aligned global address on
// gfx950, or is dword alignment enough (as for ordinary global loads)?  Copies 64 x 16 B per wave from
// src + shift (shift = 0, 4, 8, 12 bytes) with per-lane gather addresses (stride 264 B, like 22-atom frames).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const unsigned char* src, unsigned* out, int shift, int stride) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[1024];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + shift + (size_t)lane * stride), (lptr_t)tile, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = ((const unsigned*)tile)[lane * 4 + i];
}
int main() {
    const int n = 64 * 264 + 64;
    std::vector<unsigned> h(n / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;   // dword i holds i
    unsigned char* d; unsigned* o;
    hipMalloc(&d, n); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int stride : {16, 264}) for (int shift : {0, 4, 8, 12}) {
        hipMemset(o, 0xff, 1024);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, shift, stride);
        unsigned r[256];
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) bad += r[l * 4 + i] != (unsigned)((shift + l * stride) / 4 + i);
        printf("stride %3d shift %2d: %s (%d wrong dwords; lane 1 got %u %u %u %u, want %u..)\n", stride, shift, bad ? "WRONG" : "ok", bad,
               r[4], r[5], r[6], r[7], (unsigned)((shift + stride) / 4));
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
