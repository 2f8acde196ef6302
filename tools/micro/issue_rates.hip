// Microbenchmark (diagnostic, not part of the product): cycles per wave-instruction on gfx950 for the
// instruction classes frames_lane_kernel is made of, at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP 64
#define ITERS 200
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int KIND>
__global__ void bench(float* out, unsigned long long* cyc, float seed, int idx) {
    float a[8]; double d[8]; f32x4 acc[4]; f32x16 vx;
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; }
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    for (int i = 0; i < 16; ++i) vx[i] = seed * i + threadIdx.x;
    const float b = seed * 1.0001f; const double bd = b;
    int vreg = threadIdx.x * 7;
    const unsigned long long t0 = stamp();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (KIND == 0) a[r & 7] = fmaf(a[r & 7], b, 1.0f);                 // fp32 fma, 8 independent chains
            if (KIND == 1) a[0] = fmaf(a[0], b, 1.0f);                         // fp32 fma, dependent
            if (KIND == 2) d[r & 7] = fma(d[r & 7], bd, 1.0);                  // fp64 fma, 8 chains
            if (KIND == 3) d[0] = fma(d[0], bd, 1.0);                          // fp64 dependent
            if (KIND == 4) a[r & 7] = __builtin_amdgcn_exp2f(a[r & 7]);        // transcendental, 8 chains
            if (KIND == 5) { const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vreg, (r + it) & 63)); a[r & 7] = fmaf(a[r & 7], s, 1.0f); } // readlane -> sgpr -> fma
            if (KIND == 6) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b, acc[r & 3], 0, 0, 0); // mfma, 4 chains
            if (KIND == 7) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b, acc[0], 0, 0, 0);         // mfma dependent
            if (KIND == 8) a[r & 7] += vx[(idx + r) & 15];                     // uniform-indexed register read + add
            if (KIND == 9) a[r & 7] = (float)(d[r & 7] * bd);                  // fp64 mul + cvt
        }
    }
    const unsigned long long t1 = stamp();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int KIND> void run(const char* name, int instr_per_rep) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 16 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * 16 * 4 * sizeof(unsigned long long));
    printf("%-34s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD = waves per block / 4, one block per CU
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps;
        const int blocks = 256 * ((64 * 4 * wps + threads - 1) / threads);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f, 3);
        hipDeviceSynchronize();
        const int nw = blocks * threads / 64;
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += v;
        printf("  %dw/SIMD: %6.2f cyc/instr", wps, sum / nw / (double)(ITERS * REP * instr_per_rep));
    }
    printf("\n"); hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("fp32 fma 8 chains", 1); run<1>("fp32 fma dependent", 1);
    run<2>("fp64 fma 8 chains", 1); run<3>("fp64 fma dependent", 1);
    run<4>("v_exp_f32 8 chains", 1); run<5>("readlane->sgpr->fma (2 instr)", 2);
    run<6>("mfma f32 16x16x4, 4 chains", 1); run<7>("mfma f32 16x16x4 dependent", 1);
    run<8>("indexed vreg read + add", 1); run<9>("fp64 mul + cvt_f32 (2 instr)", 2);
    return 0;
}
