// Microbenchmark (diagnostic, not part of the product): does a long straight-line VALU body (the shape of the
// plan-specialised lane kernel: ~2000 instructions per tile, no reuse inside a tile) run at the issue rate of a
// short loop, or is it bound by instruction fetch?  Same instruction count in every variant:
//   LONG   2048 instructions per loop iteration (16 KB of VOP3)     SHORT 64 per iteration
//   fma    v_fma_f32 (8-byte VOP3)   fmac  v_fmac_f32 (4-byte VOP2)   pk  v_pk_fma_f32   dep2  two dependent chains
// at 1 / 2 / 3 / 4 waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define CL "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
#define BODY8_FMA  "v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v48, v49\n v_fma_f32 v42, v42, v48, v49\n v_fma_f32 v43, v43, v48, v49\n" \
                   "v_fma_f32 v44, v44, v48, v49\n v_fma_f32 v45, v45, v48, v49\n v_fma_f32 v46, v46, v48, v49\n v_fma_f32 v47, v47, v48, v49\n"
#define BODY8_FMAC "v_fmac_f32 v40, v48, v49\n v_fmac_f32 v41, v48, v49\n v_fmac_f32 v42, v48, v49\n v_fmac_f32 v43, v48, v49\n" \
                   "v_fmac_f32 v44, v48, v49\n v_fmac_f32 v45, v48, v49\n v_fmac_f32 v46, v48, v49\n v_fmac_f32 v47, v48, v49\n"
#define BODY8_PK   "v_pk_fma_f32 v[40:41], v[40:41], v[48:49], v[50:51]\n v_pk_fma_f32 v[42:43], v[42:43], v[48:49], v[50:51]\n" \
                   "v_pk_fma_f32 v[44:45], v[44:45], v[48:49], v[50:51]\n v_pk_fma_f32 v[46:47], v[46:47], v[48:49], v[50:51]\n" \
                   "v_pk_fma_f32 v[52:53], v[52:53], v[48:49], v[50:51]\n v_pk_fma_f32 v[54:55], v[54:55], v[48:49], v[50:51]\n" \
                   "v_pk_fma_f32 v[40:41], v[40:41], v[48:49], v[50:51]\n v_pk_fma_f32 v[42:43], v[42:43], v[48:49], v[50:51]\n"
#define BODY8_DEP2 "v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v48, v49\n v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v48, v49\n" \
                   "v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v48, v49\n v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v48, v49\n"
#define BODY8_DEP1 "v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v40, v40, v48, v49\n" \
                   "v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v40, v40, v48, v49\n"
#define BODY8_DEP4 "v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v48, v49\n v_fma_f32 v42, v42, v48, v49\n v_fma_f32 v43, v43, v48, v49\n" \
                   "v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v48, v49\n v_fma_f32 v42, v42, v48, v49\n v_fma_f32 v43, v43, v48, v49\n"
#define BODY8_EXP  "v_exp_f32 v40, v40\n v_exp_f32 v41, v41\n v_exp_f32 v42, v42\n v_exp_f32 v43, v43\n v_exp_f32 v44, v44\n v_exp_f32 v45, v45\n v_exp_f32 v46, v46\n v_exp_f32 v47, v47\n"
#define BODY8_MIX  "v_exp_f32 v40, v40\n v_fma_f32 v41, v41, v48, v49\n v_fma_f32 v42, v42, v48, v49\n v_fma_f32 v43, v43, v48, v49\n v_exp_f32 v44, v44\n v_fma_f32 v45, v45, v48, v49\n v_fma_f32 v46, v46, v48, v49\n v_fma_f32 v47, v47, v48, v49\n"

// tanh as the kernel computes it (mul, exp, add, rcp, fma), as 1 / 2 / 4 interleaved dependent chains (40 instructions per body)
#define TANH(r) "v_mul_f32 " r ", " r ", v48\n v_exp_f32 " r ", " r "\n v_add_f32 " r ", " r ", v49\n v_rcp_f32 " r ", " r "\n v_fma_f32 " r ", " r ", v48, v49\n"
#define T5(a, b, c, d, e, r0, r1, r2, r3) \
    "v_mul_f32 " r0 ", " r0 ", v48\n v_mul_f32 " r1 ", " r1 ", v48\n v_mul_f32 " r2 ", " r2 ", v48\n v_mul_f32 " r3 ", " r3 ", v48\n" \
    "v_exp_f32 " r0 ", " r0 "\n v_exp_f32 " r1 ", " r1 "\n v_exp_f32 " r2 ", " r2 "\n v_exp_f32 " r3 ", " r3 "\n" \
    "v_add_f32 " r0 ", " r0 ", v49\n v_add_f32 " r1 ", " r1 ", v49\n v_add_f32 " r2 ", " r2 ", v49\n v_add_f32 " r3 ", " r3 ", v49\n" \
    "v_rcp_f32 " r0 ", " r0 "\n v_rcp_f32 " r1 ", " r1 "\n v_rcp_f32 " r2 ", " r2 "\n v_rcp_f32 " r3 ", " r3 "\n" \
    "v_fma_f32 " r0 ", " r0 ", v48, v49\n v_fma_f32 " r1 ", " r1 ", v48, v49\n v_fma_f32 " r2 ", " r2 ", v48, v49\n v_fma_f32 " r3 ", " r3 ", v48, v49\n"
#define BODY40_TANH1 TANH("v40") TANH("v40") TANH("v40") TANH("v40") TANH("v40") TANH("v40") TANH("v40") TANH("v40")
#define BODY40_TANH4 T5(0,0,0,0,0,"v40","v41","v42","v43") T5(0,0,0,0,0,"v40","v41","v42","v43")
// an MFMA whose B operand is the end of a fresh tanh chain, as in the fused MLP's second layer: 4 frame blocks in turn
#define MT(acc, r) TANH(r) "s_nop 1\n v_mfma_f32_16x16x4_f32 " acc ", v50, " r ", " acc "\n"
#define BODY_MFMA_TANH MT("v[56:59]","v40") MT("v[60:63]","v41") MT("v[64:67]","v42") MT("v[68:71]","v43")
// the same work with the four tanh chains interleaved ahead of the four MFMAs
#define BODY_MFMA_TANH4 T5(0,0,0,0,0,"v40","v41","v42","v43") "s_nop 1\n v_mfma_f32_16x16x4_f32 v[56:59], v50, v40, v[56:59]\n v_mfma_f32_16x16x4_f32 v[60:63], v50, v41, v[60:63]\n" \
    "v_mfma_f32_16x16x4_f32 v[64:67], v50, v42, v[64:67]\n v_mfma_f32_16x16x4_f32 v[68:71], v50, v43, v[68:71]\n"
#define BODY_MFMA4 "v_mfma_f32_16x16x4_f32 v[56:59], v50, v40, v[56:59]\n v_mfma_f32_16x16x4_f32 v[60:63], v50, v41, v[60:63]\n" \
    "v_mfma_f32_16x16x4_f32 v[64:67], v50, v42, v[64:67]\n v_mfma_f32_16x16x4_f32 v[68:71], v50, v43, v[68:71]\n"
// mixes: what besides plain VALU the lane kernel's stream is made of (per body: 8 v_fma + the extra instructions)
#define BODY_SALU3 BODY8_FMA "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n"
#define BODY_SALU8 BODY8_FMA "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
#define BODY_NOP BODY8_FMA "s_nop 1\n s_nop 1\n"
#define BODY_EXEC BODY8_FMA "v_cmp_gt_f32 vcc, v40, v49\n s_and_saveexec_b64 s[24:25], vcc\n v_fma_f32 v41, v41, v48, v49\n s_or_b64 exec, exec, s[24:25]\n"
#define BODY_CND BODY8_FMA "v_cmp_gt_f32 vcc, v40, v49\n v_cndmask_b32 v42, v42, v43, vcc\n v_cmp_gt_f32 vcc, v41, v49\n v_cndmask_b32 v44, v44, v45, vcc\n"
#define BODY_BR BODY8_FMA "s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n"
#define BODY_WAIT BODY8_FMA "s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n"
#undef CL
#define CL "s20","s21","s22","s23","s24","s25","vcc","scc", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"

#define KERNEL(NAME, BODY, REPT, ITERS)                                                                   \
    __global__ void NAME(float* out, unsigned long long* cyc, float seed) {                                \
        asm volatile("v_mov_b32 v48, %0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, %0\n v_mov_b32 v51, 1.0\n"    \
                     "v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n"      \
                     "v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"      \
                     "v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n" ::"v"(seed) : CL); \
        const unsigned long long t0 = stamp();                                                            \
        for (int it = 0; it < ITERS; ++it) asm volatile(".rept " #REPT "\n" BODY ".endr\n" ::: CL);        \
        const unsigned long long t1 = stamp();                                                            \
        float r;                                                                                          \
        asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v47" : "=v"(r)::CL);                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                   \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;          \
    }
KERNEL(long_fma, BODY8_FMA, 256, 50)
KERNEL(short_fma, BODY8_FMA, 8, 1600)
KERNEL(long_fmac, BODY8_FMAC, 256, 50)
KERNEL(short_fmac, BODY8_FMAC, 8, 1600)
KERNEL(long_pk, BODY8_PK, 256, 50)
KERNEL(short_pk, BODY8_PK, 8, 1600)
KERNEL(short_dep1, BODY8_DEP1, 8, 1600)
KERNEL(short_dep2, BODY8_DEP2, 8, 1600)
KERNEL(short_dep4, BODY8_DEP4, 8, 1600)
KERNEL(long_dep2, BODY8_DEP2, 256, 50)
KERNEL(short_exp, BODY8_EXP, 8, 1600)
KERNEL(short_mix, BODY8_MIX, 8, 1600)
KERNEL(huge_fma, BODY8_FMA, 1024, 12)   // 64 KB body
KERNEL(mix_salu3, BODY_SALU3, 8, 1600)
KERNEL(mix_salu8, BODY_SALU8, 8, 1600)
KERNEL(mix_nop, BODY_NOP, 8, 1600)
KERNEL(mix_exec, BODY_EXEC, 8, 1600)
KERNEL(mix_cnd, BODY_CND, 8, 1600)
KERNEL(mix_wait, BODY_WAIT, 8, 1600)
KERNEL(tanh_dep1, BODY40_TANH1, 8, 320)
KERNEL(tanh_dep4, BODY40_TANH4, 8, 320)
KERNEL(mfma_tanh, BODY_MFMA_TANH, 8, 400)     // 24 instr (4 mfma) per body
KERNEL(mfma_tanh4, BODY_MFMA_TANH4, 8, 400)   // 25 instr (4 mfma)
KERNEL(mfma4, BODY_MFMA4, 8, 400)             // 4 mfma

typedef void (*kern_t)(float*, unsigned long long*, float);
void run(const char* name, kern_t k, double n_instr) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 16 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * 16 * 4 * sizeof(unsigned long long));
    printf("%-12s", name);
    for (int wps = 1; wps <= 4; ++wps) {           // waves per SIMD: one block of 4*wps waves per CU
        const int threads = 64 * 4 * wps, blocks = 256;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(b, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        const int nw = blocks * threads / 64;
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += v;
        // cycles per instruction as one wave sees it, and the SIMD's instructions per cycle implied by it
        const double cpi = sum / nw / n_instr;
        printf("  %dw: %5.2f c/i (%4.2f i/c/SIMD, %.0f us)", wps, cpi, wps / cpi, ms * 1e3);
    }
    printf("\n"); hipFree(out); hipFree(cyc);
}
int main() {
    const double N = 2048.0 * 50;
    // per body of 8 v_fma + extras: cycles per BODY (n = bodies)
    run("fma8 alone", short_fma, N / 8); run("+3 salu", mix_salu3, N / 8); run("+8 salu", mix_salu8, N / 8); run("+2 s_nop 1", mix_nop, N / 8);
    run("+exec mask", mix_exec, N / 8); run("+2 cmp/cnd", mix_cnd, N / 8); run("+2 waitcnt", mix_wait, N / 8);
    return 0;
    run("long_fma", long_fma, N); run("short_fma", short_fma, N); run("huge_fma", huge_fma, 8192.0 * 12);
    run("long_fmac", long_fmac, N); run("short_fmac", short_fmac, N);
    run("long_pk", long_pk, N); run("short_pk", short_pk, N);
    run("short_dep1", short_dep1, N); run("short_dep2", short_dep2, N); run("short_dep4", short_dep4, N); run("long_dep2", long_dep2, N);
    run("short_exp", short_exp, N); run("short_mix", short_mix, N);
    run("tanh_dep1", tanh_dep1, N); run("tanh_dep4", tanh_dep4, N);
    // per MFMA (4 per body): cycles per (tanh + mfma) group
    run("mfma_tanh", mfma_tanh, 4.0 * 8 * 400); run("mfma_tanh4", mfma_tanh4, 4.0 * 8 * 400); run("mfma4", mfma4, 4.0 * 8 * 400);
    return 0;
}
