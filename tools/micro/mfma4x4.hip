// Microbenchmark + layout check (diagnostic): v_mfma_f32_4x4x1_16b_f32, the only f32 MFMA whose blocks stay inside
// their own 4 lanes: D_b[i][j] += A_b[i] * B_b[j] with b = lane / 4.  Hypothesis checked here: A_b[i] is read from
// lane 4b+i, B_b[j] from lane 4b+j, D_b[i][j] lands in VGPR i of lane 4b+j - i.e. with lane = frame,
// B = the lane's own activation and A = 4 weights replicated over the 16 blocks, each lane receives 4 outputs
// of ITS OWN frame: a per-lane matvec with shared weights, no transposition through LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__global__ void layout(float* out) {
    const int l = threadIdx.x;
    const float a = 1.0f + l, b = 100.0f + 3.0f * l;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[4 * l + i] = d[i];
}
// MODE 0: 8 independent accumulators, back to back   1: one accumulator (dependent chain)
// MODE 2: 8 accumulators, 2 v_fma between MFMAs       3: 8 accumulators, 1 v_exp + 1 v_fma between MFMAs
template <int MODE>
__global__ void rate(float* out, unsigned long long* cyc, float seed) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 0.5f, v0 = seed, v1 = seed + 1.f;
    const unsigned long long t0 = stamp();
    for (int it = 0; it < 400; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            acc[MODE == 1 ? 0 : (r & 7)] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[MODE == 1 ? 0 : (r & 7)], 0, 0, 0);
            if (MODE == 2) { v0 = fmaf(v0, b, 1.0f); v1 = fmaf(v1, b, 1.0f); }
            if (MODE == 3) { v0 = __builtin_amdgcn_exp2f(v0); v1 = fmaf(v1, b, 1.0f); }
        }
    }
    const unsigned long long t1 = stamp();
    float s = v0 + v1;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
    printf("%-28s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, 1.0f);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(4 * wps * 256);
        (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += v;
        printf("  %dw: %6.2f cyc/mfma/wave, %4.0f us", wps, sum / h.size() / (400.0 * 32), ms * 1e3);
    }
    printf("\n");
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * sizeof(float));
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
    float h[256]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const float want = (1.0f + (4 * (l / 4) + i)) * (100.0f + 3.0f * l);
            if (h[4 * l + i] != want) { if (bad < 8) printf("lane %d vgpr %d: got %g want %g\n", l, i, h[4 * l + i], want); ++bad; }
        }
    printf("layout hypothesis (D[vgpr i][lane l] = A[lane 4(l/4)+i] * B[lane l]): %s\n", bad ? "WRONG" : "confirmed");
    run<0>("8 accumulators"); run<1>("1 accumulator (dependent)"); run<2>("8 acc + 2 fma / mfma"); run<3>("8 acc + exp + fma / mfma");
    return 0;
}
