// molann_kernels.hip - gfx950 (MI355X / CDNA4) kernels + C ABI for molann's per-frame forward path.
//
//   x[N, n_inp, 3] -> AlignmentLayer (Kabsch, ann.py:157-199) -> FeatureLayer (ann.py:454-474)
//                  -> MLP (create_sequential_nn, ann.py:37-67) -> y[N, d_out]
//
// Kernel families (DESIGN.md has the byte/flop accounting for each).  The first three are compiled ahead of time
// in this file; each has a plan-specialised sibling compiled at plan creation with hipRTC from the .inc files
// next to it (jit_sources.gen.h embeds their text), which is what normally runs:
//
//  * frames_lane_kernel   small frames (22-atom class).  One LANE per frame, one wave = a tile of 64
//    consecutive frames = one contiguous span of HBM, copied to the wave's private LDS region by
//    LDS-DMA (global_load_lds, 16 B per lane, no VGPRs).  Each lane then reads its atoms from LDS,
//    solves its own Kabsch rotation, evaluates the feature table and runs the small MLP on the fp32 MFMA
//    (computed transposed: a layer's accumulator is the next layer's B operand).  No barriers: waves never
//    share data.  Specialised: molann_lane_jit.inc (tables as constants, compact staging of the touched
//    16-byte windows only); backward: molann_lane_bwd.inc.
//
//  * frames_wave_kernel   large frames (5000-atom class).  One WAVE per frame: lanes gather only the
//    atoms the plan touches, the 3x3 covariance is a wave reduction (DPP), every lane solves the
//    same rotation, lanes then split the feature table.  Backward: frames_wave_bwd_kernel.
//
//  * mlp_mfma_kernel      wide MLPs.  One wave per 16-frame row block, activations in the wave's LDS
//    region, weights streamed from L2 as MFMA B-fragments (fp32-input MFMA 16x16x4, exact fp32; or
//    bf16 MFMA 16x16x32 with fp32 accumulate).  Specialised (bf16): molann_mlp_jit.inc (activations chained
//    through the accumulators, weights streamed once per block through LDS slabs).
//
// gfx950 only: wave = 64, LDS-DMA, DPP row ops, v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_bf16.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <new>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/molann_hip.h"
#include "molann_math.h"
#include "jit_sources.gen.h"

using namespace molann;

namespace {

// ---------------------------------------------------------------------------------------------
// address-space helpers
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Plan data never changes during a launch: reading it through the constant address space lets the
// compiler use scalar loads (s_load_dword*) for wave-uniform addresses.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* as_const(const T* p) {
    return (const __attribute__((address_space(4))) T*)(p);
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 4, 0, 0);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ItemDev { // one expanded feature item, 32 bytes
    int type;    // ItemType
    int col;     // first output column
    int idx[4];  // atom positions inside the n_inp axis (unused entries repeat idx[0])
    int pad[2];
};

constexpr int FB_STRIDE = 65; // feature/activation staging [col][FB_STRIDE]: lane-contiguous, odd stride
constexpr int LANE_MLP_MAX_WIDTH = 32;  // widths (and the feature dim) the fused MFMA MLP covers
constexpr int LANE_MLP_MAX_LAYERS = 4;
constexpr int LANE_MAX_COLS = 128; // widest feature / output row the lane kernel stages in LDS

// ---------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------
struct PreArgs {
    long n_frames;
    int n_inp;
    int frame_dw;     // 3 * n_inp
    int mode;         // 0: features (+ fused MLP)   1: aligned coordinates
    int n_align;
    int n_items;
    int n_slots;      // touched atoms of the register-resident mode
    int out_cols;     // columns written per frame in mode 0
    int step_f;       // 64 / out_cols
    int step_c;       // 64 % out_cols
    int lds_per_wave; // bytes (lane kernel)
    int fbuf_off;     // byte offset of the staging buffer inside the wave's region
    int x_wide;       // x is 16-byte aligned
    int out_wide;     // out is 16-byte aligned
    int out_vec4;     // fused MLP: out rows can be written with 16-byte stores
    int n_layers;     // fused MLP (lane kernel only); 0 = none
    int act;
    int dims[MOLANN_MAX_LAYERS + 1];
    int ablate;       // diagnostic only (MOLANN_DEBUG_ABLATE): skip stages to price them; 0 in production
};

// ---------------------------------------------------------------------------------------------
// wave reduction (sum over the 64 lanes, result broadcast) with DPP row operations
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                                                 BANK_MASK, BOUND));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v); // row_half_mirror
    v += dpp_mov<0x140>(v); // row_mirror       -> every lane holds its row's (16-lane) sum
    v += dpp_mov<0x142, 0xa>(v); // row_bcast15 : lane 15 -> row 1, lane 47 -> row 3
    v += dpp_mov<0x143, 0xc>(v); // row_bcast31 : lane 31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double wave_sum(double v) {
    // same butterfly on the two 32-bit halves of each partner value
    auto step = [](double x, auto mover) {
        const long long b = __builtin_bit_cast(long long, x);
        const int lo = mover((int)(b & 0xffffffffll)), hi = mover((int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    };
    v += step(v, [](int w) { return __builtin_amdgcn_update_dpp(0, w, 0xB1, 0xf, 0xf, true); });
    v += step(v, [](int w) { return __builtin_amdgcn_update_dpp(0, w, 0x4E, 0xf, 0xf, true); });
    v += step(v, [](int w) { return __builtin_amdgcn_update_dpp(0, w, 0x141, 0xf, 0xf, true); });
    v += step(v, [](int w) { return __builtin_amdgcn_update_dpp(0, w, 0x140, 0xf, 0xf, true); });
    v += step(v, [](int w) { return __builtin_amdgcn_update_dpp(0, w, 0x142, 0xa, 0xf, true); });
    v += step(v, [](int w) { return __builtin_amdgcn_update_dpp(0, w, 0x143, 0xc, 0xf, true); });
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// ---------------------------------------------------------------------------------------------
// activation over a register array (switch hoisted out of the unrolled loop)
// ---------------------------------------------------------------------------------------------
// CHEAP: only the activations that are a handful of instructions (the lane kernel unrolls this over
// up to 32 registers; ELU / Softplus / GELU pull in libm-sized code and go through the MFMA MLP kernel).
template <int W, bool CHEAP>
__device__ __forceinline__ void activate(int act, float (&h)[W]) {
    switch (act) {
    case MOLANN_ACT_TANH:
#pragma unroll
        for (int j = 0; j < W; ++j) h[j] = act_tanh(h[j]);
        break;
    case MOLANN_ACT_RELU:
#pragma unroll
        for (int j = 0; j < W; ++j) h[j] = apply_activation(MOLANN_ACT_RELU, h[j]);
        break;
    case MOLANN_ACT_SIGMOID:
#pragma unroll
        for (int j = 0; j < W; ++j) h[j] = act_sigmoid(h[j]);
        break;
    case MOLANN_ACT_SILU:
#pragma unroll
        for (int j = 0; j < W; ++j) h[j] = h[j] * act_sigmoid(h[j]);
        break;
    case MOLANN_ACT_LEAKY_RELU:
#pragma unroll
        for (int j = 0; j < W; ++j) h[j] = apply_activation(MOLANN_ACT_LEAKY_RELU, h[j]);
        break;
    case MOLANN_ACT_IDENTITY:
        break;
    default:
        if constexpr (!CHEAP) {
#pragma unroll
            for (int j = 0; j < W; ++j) h[j] = apply_activation(act, h[j]);
        }
        break;
    }
}

// ---------------------------------------------------------------------------------------------
// diagnostic phase stamps (MOLANN_DEBUG_ABLATE bit 32): per-phase shader-clock sums of every wave,
// added to this array by lane 0 at kernel end and read back by molann_debug_read_stamps.  Shares of a
// wave's time, not a timing of the production kernel (the stamps drain LDS/scalar queues).
// ---------------------------------------------------------------------------------------------
__device__ unsigned long long g_stamps[16];

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// ---------------------------------------------------------------------------------------------
// lane-kernel helpers
// ---------------------------------------------------------------------------------------------
// Plan constants (align table, feature table) are held one record per LANE in VGPRs for the whole
// kernel and broadcast with v_readlane when a loop needs record i: no memory access, no latency.
__device__ __forceinline__ int bcast(int v, int i) { return __builtin_amdgcn_readlane(v, i); }
__device__ __forceinline__ double bcast(double v, int i) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), i), hi = __builtin_amdgcn_readlane((int)(b >> 32), i);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// HBM -> LDS for one tile of up to 64 WHOLE frames (contiguous in HBM): LDS-DMA, 16 B per lane when x is
// 16-byte aligned, else 4 B per lane.  LDS image = the HBM bytes: [frame][atom][xyz].  Used when every
// atom is needed (AlignmentLayer.forward).  Returns without waiting; the consumer waits on vmcnt.
__device__ __forceinline__ void stage_tile_dense(const float* __restrict__ x, float* tile, long t, const PreArgs& a, int lane) {
    // WAR: every LDS read of the tile's previous contents has returned before the DMA may overwrite it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int frame_bytes = a.frame_dw * 4;
    const long rem = a.n_frames - t * 64;
    const int nfr = rem < 64 ? (int)rem : 64;
    const int valid_bytes = nfr * frame_bytes;
    const unsigned char* gsrc = (const unsigned char*)x + t * ((long)frame_bytes * 64);
    if (a.x_wide) {
        const int nchunk = valid_bytes >> 4;
        for (int c0 = 0; c0 < nchunk; c0 += 64) {
            const int c = c0 + lane;
            if (c < nchunk) glds16(gsrc + (size_t)c * 16, (unsigned char*)tile + (size_t)c0 * 16);
        }
        const int rem_dw = (valid_bytes & 15) >> 2;
        if (lane < rem_dw) glds4(gsrc + (size_t)nchunk * 16 + lane * 4, (unsigned char*)tile + (size_t)nchunk * 16);
    } else {
        const int ndw = valid_bytes >> 2;
        for (int c0 = 0; c0 < ndw; c0 += 64) {
            const int c = c0 + lane;
            if (c < ndw) glds4(gsrc + (size_t)c * 4, (unsigned char*)tile + (size_t)c0 * 4);
        }
    }
}

// atom k of this lane's frame row in the LDS tile
__device__ __forceinline__ V3 lds_atom(const float* fr, int k) { return v3(fr[3 * k], fr[3 * k + 1], fr[3 * k + 2]); }

template <int N>
__device__ __forceinline__ f32x4 mfma_chain(const float (&wa)[8], int w0, const float* b, f32x4 acc) {
#pragma unroll
    for (int k = 0; k < N; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[w0 + k], b[k], acc, 0, 0, 0);
    return acc;
}

// =============================================================================================
// frames_lane_kernel: one lane per frame, 64-frame tiles staged through LDS by LDS-DMA
// =============================================================================================
// MODE 1 (align-out): AlignmentLayer.forward, aligned frames written back.
// MODE 0 (features) : features (+ fused MLP when NL > 0), atoms read from the LDS tile where needed.
// MODE 2 (features, register-resident): at most 16 touched atoms ("slots", align atoms first).  As soon
//   as the tile lands each lane copies its frame's touched atoms into 48 registers (static fill; later
//   reads index the register file with a wave-uniform index, s_set_gpr_idx) and the tile buffer goes
//   straight back to the DMA engine, so the next tile's HBM latency runs under ALL of this tile's work.
// (Staging only the touched atoms with per-lane-address 12-byte LDS-DMA was measured and rejected: it
// halves the LDS footprint but each such instruction touches 64 different 128-B lines, ~9 us per staged
// atom per 1M frames against 66 us for the whole dense tile.)
//
// Fused MLP (NL Linear layers, every width <= 32, feature dim <= 32) on the fp32-input MFMA
// v_mfma_f32_16x16x4_f32, computed TRANSPOSED: D[unit][frame] = W[unit][k] . X[k][frame].
//   * A operand = weights: lane (i = l&15, q = l>>4) holds W[16ub + i][k(q)] - loaded ONCE per kernel
//     into registers (wfrag, laid out per lane by pack_lane_kernel), never re-read.
//   * B operand = activations: lane (frame = l&15, q) holds X[k(q)][frame].  Layer 0 reads the features
//     from the wave's staging buffer, k(q) = 4 ks + q.  For the following layers the k-steps are
//     enumerated as (ub', r) with k(q) = 16 ub' + 4 q + r, which is exactly the (row 4q + r, col frame)
//     element the previous layer's accumulator register r of unit block ub' holds in that same lane:
//     the accumulator IS the next B operand, no shuffle and no LDS round trip between layers.
//   * The last layer's accumulator holds out[frame][16ub + 4q + r], r = 0..3: one 16-byte store per
//     lane, whole output rows per 16-lane group.
// fp32 MFMA is an exact k-ordered fmaf chain (cdna_hip_programming.md section 3), so this is fp32 math.
template <int NL, int MODE>
__global__ __launch_bounds__(256) void frames_lane_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          const int* __restrict__ align_tbl_g,
                                                          const float* __restrict__ ref_g,    // [a*3] + consts
                                                          const double* __restrict__ ref64_g, // same, fp64
                                                          const ItemDev* __restrict__ items_g,
                                                          const int* __restrict__ slots_g, // MODE 2: slot -> atom
                                                          const float* __restrict__ wfrag_g, PreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool ALIGN_OUT = (MODE == 1);
    constexpr bool REGS = (MODE == 2);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    unsigned char* wreg = smem + (size_t)wave * a.lds_per_wave;
    float* tile = (float*)wreg;
    float* fbuf = (float*)(wreg + a.fbuf_off);
    const auto ref = as_const(ref_g);
    const auto ref64 = as_const(ref64_g);
    int slot_atom = 0; // MODE 2: lane u holds the atom index of slot u (0 beyond n_slots: a valid atom)
    if (REGS && lane < a.n_slots) slot_atom = slots_g[lane];

    const long n_tiles = (a.n_frames + 63) >> 6;
    const long t_first = (long)blockIdx.x * wpb + wave;
    const long t_step = (long)gridDim.x * wpb;

    // ---- per-kernel constants, one record per lane (broadcast later with v_readlane) -----------
    if (t_first < n_tiles) stage_tile_dense(x, tile, t_first, a, lane); // first tile in flight while constants load
    const bool has_align = a.n_align > 0;
    int al_idx = 0;
    double al_rx = 0., al_ry = 0., al_rz = 0.;
    if (has_align && lane < a.n_align) {
        al_idx = align_tbl_g[lane];
        al_rx = ref64_g[3 * lane]; al_ry = ref64_g[3 * lane + 1]; al_rz = ref64_g[3 * lane + 2];
    }
    int it_type = 0, it_col = 0, it_i0 = 0, it_i1 = 0, it_i2 = 0, it_i3 = 0;
    if (!ALIGN_OUT && lane < a.n_items) {
        const int4 d0 = ((const int4*)items_g)[2 * lane];
        const int2 d1 = ((const int2*)items_g)[4 * lane + 2];
        it_type = d0.x; it_col = d0.y; it_i0 = d0.z; it_i1 = d0.w; it_i2 = d1.x; it_i3 = d1.y;
    }
    float wA[NL > 0 ? NL : 1][2][8];
    f32x4 wB[NL > 0 ? NL : 1][2];
    if constexpr (NL > 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) wA[l][ub][ks] = wfrag_g[((l * 2 + ub) * 8 + ks) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) wB[l][ub][r] = wfrag_g[NL * 1024 + ((l * 2 + ub) * 4 + r) * 64 + lane];
            }
        // feature rows beyond the feature dim feed zero weights: keep them finite
        for (int c = a.dims[0]; c < ((a.dims[0] + 3) & ~3); ++c) fbuf[c * FB_STRIDE + lane] = 0.f;
    }
    const int i16 = lane & 15, q4 = lane >> 4;

    // read-out position of this lane (no fused MLP): element e = it*64 + lane of the [64][out_cols] tile
    int ro_f0 = 0, ro_c0 = 0;
    if (!ALIGN_OUT && NL == 0) {
        ro_f0 = lane / a.out_cols;
        ro_c0 = lane - ro_f0 * a.out_cols;
    }

    const bool stamps = (a.ablate & 32) != 0;
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, ts = 0;
#define MOLANN_STAMP(i)                            \
    if (stamps) {                                  \
        const unsigned long long now_ = stamp();   \
        ph[i] += now_ - ts;                        \
        ts = now_;                                 \
    }
    if (stamps) ts = stamp();
    for (long t = t_first; t < n_tiles; t += t_step) {
        const long rem = a.n_frames - t * 64;
        const int nfr = rem < 64 ? (int)rem : 64;
        // ---- 1. the tile was requested one iteration ago (or above): wait for the LDS-DMA -------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MOLANN_STAMP(0) // waiting for DMA + previous stores

        // lanes past the end of the batch recompute the last valid frame (their stores are masked)
        const int fl = lane < nfr ? lane : nfr - 1;
        const float* fr = tile + fl * a.frame_dw;

        f32x16 ax, ay, az; // MODE 2: the touched atoms of this lane's frame
        if constexpr (REGS) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const V3 p = lds_atom(fr, bcast(slot_atom, u));
                ax[u] = p.x; ay[u] = p.y; az[u] = p.z;
            }
            // the tile is dead: hand its buffer back to the DMA engine for the next tile right now
            if (t + t_step < n_tiles) stage_tile_dense(x, tile, t + t_step, a, lane);
        }
        MOLANN_STAMP(1) // register fill + DMA issue

        // ---- 2. Kabsch (ann.py:179-195) --------------------------------------------------------
        float R[9];
        V3 c0 = v3(0.f, 0.f, 0.f), dl = v3(0.f, 0.f, 0.f);
        if (has_align) {
            float sx = 0.f, sy = 0.f, sz = 0.f, g = 0.f;
            double h[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
            const int na = (a.ablate & 8) ? 1 : a.n_align;
            auto accumulate = [&](const V3 p, const double rx, const double ry, const double rz) {
                sx += p.x; sy += p.y; sz += p.z;
                g = fmaf(p.x, p.x, fmaf(p.y, p.y, fmaf(p.z, p.z, g)));
                const double px = p.x, py = p.y, pz = p.z;
                h[0] = fma(px, rx, h[0]); h[1] = fma(px, ry, h[1]); h[2] = fma(px, rz, h[2]);
                h[3] = fma(py, rx, h[3]); h[4] = fma(py, ry, h[4]); h[5] = fma(py, rz, h[5]);
                h[6] = fma(pz, rx, h[6]); h[7] = fma(pz, ry, h[7]); h[8] = fma(pz, rz, h[8]);
            };
            if constexpr (REGS) {
                c0 = v3(ax[0], ay[0], az[0]); // provisional centre: first align atom = slot 0
#pragma unroll
                for (int i = 0; i < 16; ++i) // align atom i IS slot i (plan guarantees it)
                    if (i < na) accumulate(v3(ax[i], ay[i], az[i]) - c0, bcast(al_rx, i), bcast(al_ry, i), bcast(al_rz, i));
            } else {
                c0 = lds_atom(fr, bcast(al_idx, 0)); // provisional centre: first align atom
#pragma unroll 4
                for (int i = 0; i < na; ++i)
                    accumulate(lds_atom(fr, bcast(al_idx, i)) - c0, bcast(al_rx, i), bcast(al_ry, i), bcast(al_rz, i));
            }
            // constants after the reference coordinates: sum ref (3), sum |ref|^2, 1/a, a
            const int cb = 3 * a.n_align;
            const double srx = ref64[cb], sry = ref64[cb + 1], srz = ref64[cb + 2], gref = ref64[cb + 3];
            const float inv_a = ref[cb + 4], fa = ref[cb + 5];
            dl = v3(sx * inv_a, sy * inv_a, sz * inv_a); // centroid = c0 + dl (ann.py:181)
            // H = sum (p - dl) ref^T = sum p ref^T - dl (sum ref)^T   (ann.py:183-187)
            const double dx = dl.x, dy = dl.y, dz = dl.z;
            h[0] = fma(-dx, srx, h[0]); h[1] = fma(-dx, sry, h[1]); h[2] = fma(-dx, srz, h[2]);
            h[3] = fma(-dy, srx, h[3]); h[4] = fma(-dy, sry, h[4]); h[5] = fma(-dy, srz, h[5]);
            h[6] = fma(-dz, srx, h[6]); h[7] = fma(-dz, sry, h[7]); h[8] = fma(-dz, srz, h[8]);
            const float gp = fmaxf(g - fa * dot(dl, dl), 0.f);
            MOLANN_STAMP(2) // covariance accumulation
            if (a.ablate & 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = (float)h[i];
            } else {
                kabsch_rotation(h, 0.5 * ((double)gp + gref) * 1.0001, R);
            }
            MOLANN_STAMP(3) // rotation solve
        }

        if constexpr (ALIGN_OUT) {
            // ---- 3a. AlignmentLayer.forward: every atom, in place, then LDS -> HBM ------------
            const int frame_bytes = a.frame_dw * 4;
            const int valid_bytes = nfr * frame_bytes;
            float* frw = tile + fl * a.frame_dw;
#pragma unroll 2
            for (int k = 0; k < a.n_inp; ++k) {
                const V3 p = (v3(frw[3 * k], frw[3 * k + 1], frw[3 * k + 2]) - c0) - dl;
                const V3 y = rotate(p, R); // ann.py:197
                if (lane < nfr) {
                    frw[3 * k] = y.x; frw[3 * k + 1] = y.y; frw[3 * k + 2] = y.z;
                }
            }
            unsigned char* gdst = (unsigned char*)out + t * ((long)frame_bytes * 64);
            if (a.out_wide) {
                const int nchunk = valid_bytes >> 4;
                for (int c = lane; c < nchunk; c += 64) ((float4*)gdst)[c] = ((const float4*)tile)[c];
                const int done_dw = nchunk << 2, ndw = valid_bytes >> 2;
                if (done_dw + lane < ndw) ((float*)gdst)[done_dw + lane] = tile[done_dw + lane];
            } else {
                const int ndw = valid_bytes >> 2;
                for (int c = lane; c < ndw; c += 64) ((float*)gdst)[c] = tile[c];
            }
            if (t + t_step < n_tiles) stage_tile_dense(x, tile, t + t_step, a, lane);
        } else {
            // ---- 3b. feature table (ann.py:323-354, 473) -> fbuf[col][lane] -------------------
            const int n_items = (a.ablate & 2) ? 0 : a.n_items;
            for (int it = 0; it < n_items; ++it) {
                int type, col, i0, i1, i2, i3;
                if (it < 64) {
                    type = bcast(it_type, it); col = bcast(it_col, it);
                    i0 = bcast(it_i0, it); i1 = bcast(it_i1, it); i2 = bcast(it_i2, it); i3 = bcast(it_i3, it);
                } else { // long tables (identity features over many atoms): the rest comes from memory
                    const auto items = as_const((const int*)items_g);
                    type = items[8 * it]; col = items[8 * it + 1];
                    i0 = items[8 * it + 2]; i1 = items[8 * it + 3]; i2 = items[8 * it + 4]; i3 = items[8 * it + 5];
                }
                // unused entries of an item repeat its first atom: always four reads in one batch
                V3 p0, p1, p2, p3;
                if constexpr (REGS) { // wave-uniform register index (s_set_gpr_idx)
                    p0 = v3(ax[i0], ay[i0], az[i0]); p1 = v3(ax[i1], ay[i1], az[i1]);
                    p2 = v3(ax[i2], ay[i2], az[i2]); p3 = v3(ax[i3], ay[i3], az[i3]);
                } else {
                    p0 = lds_atom(fr, i0); p1 = lds_atom(fr, i1);
                    p2 = lds_atom(fr, i2); p3 = lds_atom(fr, i3);
                }
                if (has_align) { // features see the ALIGNED frame (ann.py:565)
                    p0 = rotate((p0 - c0) - dl, R);
                    if (type != IT_POSITION) {
                        p1 = rotate((p1 - c0) - dl, R);
                        p2 = rotate((p2 - c0) - dl, R);
                        p3 = rotate((p3 - c0) - dl, R);
                    }
                }
                float v[3];
                const int w = eval_item(type, p0, p1, p2, p3, v);
                fbuf[col * FB_STRIDE + lane] = v[0];
                if (w > 1) fbuf[(col + 1) * FB_STRIDE + lane] = v[1];
                if (w > 2) fbuf[(col + 2) * FB_STRIDE + lane] = v[2];
            }

            // MODE 0: the tile is dead from here on: start the next tile's LDS-DMA now, so that its HBM
            // latency runs under the MLP and the stores of this tile (MODE 2 did it right after the fill).
            if (!REGS && t + t_step < n_tiles) stage_tile_dense(x, tile, t + t_step, a, lane);
            MOLANN_STAMP(4) // feature table

            if constexpr (NL > 0) {
                // ---- 4. fused MLP on the fp32 MFMA, four blocks of 16 frames ------------------
                const int ks0 = (a.dims[0] + 3) >> 2;
                float* gout = out + (t * 64) * (long)a.out_cols;
#pragma unroll 2
                for (int fb = 0; fb < 4; ++fb) {
                    f32x4 D[2];
                    D[1] = wB[0][1];
                    {   // layer 0: B operand from the staging buffer, k = 4 ks + q
                        float bin[8];
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks)
                            bin[ks] = ks < ks0 ? fbuf[(4 * ks + q4) * FB_STRIDE + 16 * fb + i16] : 0.f;
#pragma unroll
                        for (int ub = 0; ub < 2; ++ub) {
                            if (ub == 0 || a.dims[1] > 16) {
                                f32x4 acc = wB[0][ub];
                                switch (ks0) {
                                case 1: acc = mfma_chain<1>(wA[0][ub], 0, bin, acc); break;
                                case 2: acc = mfma_chain<2>(wA[0][ub], 0, bin, acc); break;
                                case 3: acc = mfma_chain<3>(wA[0][ub], 0, bin, acc); break;
                                case 4: acc = mfma_chain<4>(wA[0][ub], 0, bin, acc); break;
                                case 5: acc = mfma_chain<5>(wA[0][ub], 0, bin, acc); break;
                                case 6: acc = mfma_chain<6>(wA[0][ub], 0, bin, acc); break;
                                case 7: acc = mfma_chain<7>(wA[0][ub], 0, bin, acc); break;
                                default: acc = mfma_chain<8>(wA[0][ub], 0, bin, acc); break;
                                }
                                D[ub] = acc;
                            }
                        }
                    }
#pragma unroll
                    for (int l = 1; l < NL; ++l) {
                        if (a.ablate & 4) break;
                        const bool wide_in = a.dims[l] > 16;
                        float h0[4] = {D[0][0], D[0][1], D[0][2], D[0][3]};
                        float h1[4] = {D[1][0], D[1][1], D[1][2], D[1][3]};
                        activate<4, true>(a.act, h0);
                        if (wide_in) activate<4, true>(a.act, h1);
#pragma unroll
                        for (int ub = 0; ub < 2; ++ub) {
                            f32x4 acc = wB[l][ub];
                            if (ub == 0 || a.dims[l + 1] > 16) {
                                acc = mfma_chain<4>(wA[l][ub], 0, h0, acc);
                                if (wide_in) acc = mfma_chain<4>(wA[l][ub], 4, h1, acc);
                            }
                            D[ub] = acc;
                        }
                    }
                    // D[ub][r] = out[frame 16 fb + i16][16 ub + 4 q + r]
                    const int frame = 16 * fb + i16;
                    if (frame < nfr && !(a.ablate & 16)) {
                        float* orow = gout + (long)frame * a.out_cols;
#pragma unroll
                        for (int ub = 0; ub < 2; ++ub) {
                            const int u0 = 16 * ub + 4 * q4;
                            if (u0 + 3 < a.out_cols && a.out_vec4) {
                                *(f32x4*)(orow + u0) = D[ub];
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (u0 + r < a.out_cols) orow[u0 + r] = D[ub][r];
                            }
                        }
                    }
                }
            } else {
                // ---- 5. read-out: fbuf[col][frame] -> out[t*64 + frame][col], contiguous per wave
                float* gdst = out + t * 64 * (long)a.out_cols;
                const int n_valid = nfr * a.out_cols;
                int f = ro_f0, c = ro_c0;
                for (int e = lane; e < n_valid; e += 64) {
                    gdst[e] = fbuf[c * FB_STRIDE + f];
                    c += a.step_c;
                    f += a.step_f;
                    if (c >= a.out_cols) { c -= a.out_cols; ++f; }
                }
            }
        }
        MOLANN_STAMP(5) // MLP / read-out + stores
    }
#undef MOLANN_STAMP
    if (stamps && lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[i], ph[i]);
        atomicAdd(&g_stamps[7], 1ull);
    }
}

// =============================================================================================
// frames_wave_kernel: one wave per frame, atoms gathered from HBM
// =============================================================================================
__device__ __forceinline__ V3 load_atom(const float* __restrict__ xf, int k) {
    return v3(xf[3 * k], xf[3 * k + 1], xf[3 * k + 2]);
}

// PRE: rounds of 64 feature items whose records live in registers for the whole kernel and whose atoms are
// loaded at the top of each frame, TOGETHER with the alignment atoms: one HBM round trip per frame instead of
// two dependent ones, and a 128-B line that holds both kinds of atom is fetched while it is still in L2 (with
// 16 waves x 60 KB per CU in flight, the alignment phase's lines are long evicted by the time a second phase
// would ask for them again).  Items beyond 64 PRE take the two-phase loop.
template <int PRE>
__global__ __launch_bounds__(256) void frames_wave_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          const int* __restrict__ align_idx,
                                                          const float* __restrict__ ref,
                                                          const double* __restrict__ ref64,
                                                          const ItemDev* __restrict__ items, PreArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const auto refc = as_const(ref);
    const bool has_align = a.n_align > 0;
    constexpr int PR = PRE > 0 ? PRE : 1;
    int pt[PR], pc[PR], pi[PR][4];
    bool pon[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int it = lane + 64 * r;
        pon[r] = PRE > 0 && a.mode != 1 && it < a.n_items;
        pt[r] = 0; pc[r] = 0; pi[r][0] = pi[r][1] = pi[r][2] = pi[r][3] = 0;
        if (pon[r]) {
            const int4 d0 = ((const int4*)items)[2 * it];
            const int2 d1 = ((const int2*)items)[4 * it + 2];
            pt[r] = d0.x; pc[r] = d0.y; pi[r][0] = d0.z; pi[r][1] = d0.w; pi[r][2] = d1.x; pi[r][3] = d1.y;
        }
    }

    for (long f = (long)blockIdx.x * wpb + wave; f < a.n_frames; f += (long)gridDim.x * wpb) {
        const float* xf = x + f * (long)a.frame_dw;
        float R[9];
        V3 c0 = v3(0.f, 0.f, 0.f), dl = v3(0.f, 0.f, 0.f);
        V3 pa[PR][4]; // raw atoms of this lane's items (atom 0 for lanes without an item: a harmless L1 hit)
        if constexpr (PRE > 0) {
#pragma unroll
            for (int r = 0; r < PR; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) pa[r][j] = load_atom(xf, pi[r][j]);
        }
        if (has_align) {
            const int k0 = as_const(align_idx)[0];
            c0 = load_atom(xf, k0);
            float sx = 0.f, sy = 0.f, sz = 0.f, g = 0.f;
            double h[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
#pragma unroll 4
            for (int i = lane; i < a.n_align; i += 64) {
                const int k = align_idx[i];
                const double rx = ref64[3 * i], ry = ref64[3 * i + 1], rz = ref64[3 * i + 2];
                const V3 p = load_atom(xf, k) - c0;
                sx += p.x; sy += p.y; sz += p.z;
                g = fmaf(p.x, p.x, fmaf(p.y, p.y, fmaf(p.z, p.z, g)));
                const double px = p.x, py = p.y, pz = p.z;
                h[0] = fma(px, rx, h[0]); h[1] = fma(px, ry, h[1]); h[2] = fma(px, rz, h[2]);
                h[3] = fma(py, rx, h[3]); h[4] = fma(py, ry, h[4]); h[5] = fma(py, rz, h[5]);
                h[6] = fma(pz, rx, h[6]); h[7] = fma(pz, ry, h[7]); h[8] = fma(pz, rz, h[8]);
            }
            sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); g = wave_sum(g);
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = wave_sum(h[i]);
            const int cb = 3 * a.n_align;
            const auto r64c = as_const(ref64);
            const double srx = r64c[cb], sry = r64c[cb + 1], srz = r64c[cb + 2], gref = r64c[cb + 3];
            const float inv_a = refc[cb + 4], fa = refc[cb + 5];
            dl = v3(sx * inv_a, sy * inv_a, sz * inv_a);
            const double dx = dl.x, dy = dl.y, dz = dl.z;
            h[0] = fma(-dx, srx, h[0]); h[1] = fma(-dx, sry, h[1]); h[2] = fma(-dx, srz, h[2]);
            h[3] = fma(-dy, srx, h[3]); h[4] = fma(-dy, sry, h[4]); h[5] = fma(-dy, srz, h[5]);
            h[6] = fma(-dz, srx, h[6]); h[7] = fma(-dz, sry, h[7]); h[8] = fma(-dz, srz, h[8]);
            const float gp = fmaxf(g - fa * dot(dl, dl), 0.f);
            kabsch_rotation(h, 0.5 * ((double)gp + gref) * 1.0001, R);
        }

        if (a.mode == 1) { // AlignmentLayer.forward: all atoms, coalesced 12 B per lane
            float* of = out + f * (long)a.frame_dw;
#pragma unroll 4
            for (int k = lane; k < a.n_inp; k += 64) {
                const V3 y = rotate((load_atom(xf, k) - c0) - dl, R);
                of[3 * k] = y.x; of[3 * k + 1] = y.y; of[3 * k + 2] = y.z;
            }
            continue;
        }

        float* of = out + f * (long)a.out_cols;
        if constexpr (PRE > 0) {
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                if (!pon[r]) continue;
                V3 p0 = pa[r][0], p1 = pa[r][1], p2 = pa[r][2], p3 = pa[r][3];
                if (has_align) align_item_atoms(pt[r], p0, p1, p2, p3, c0, dl, R);
                float v[3];
                const int w = eval_item(pt[r], p0, p1, p2, p3, v);
                of[pc[r]] = v[0];
                if (w > 1) of[pc[r] + 1] = v[1];
                if (w > 2) of[pc[r] + 2] = v[2];
            }
        }
#pragma unroll 2
        for (int it = lane + 64 * PRE; it < a.n_items; it += 64) {
            const int4 d0 = ((const int4*)items)[2 * it];
            const int2 d1 = ((const int2*)items)[4 * it + 2];
            const int type = d0.x, col = d0.y, i0 = d0.z, i1 = d0.w, i2 = d1.x, i3 = d1.y;
            V3 p0 = load_atom(xf, i0), p1 = load_atom(xf, i1), p2 = load_atom(xf, i2), p3 = load_atom(xf, i3);
            if (has_align) align_item_atoms(type, p0, p1, p2, p3, c0, dl, R);
            float v[3];
            const int w = eval_item(type, p0, p1, p2, p3, v);
            of[col] = v[0];
            if (w > 1) of[col + 1] = v[1];
            if (w > 2) of[col + 2] = v[2];
        }
    }
}

// =============================================================================================
// frames_ring_kernel: features (+ alignment) of LARGE frames, loader / consumer waves around an LDS ring
// =============================================================================================
// frames_wave_kernel (above) lets every lane fetch its own atoms from global memory: a 128-byte line that holds
// both an alignment atom and an item's atom, or atoms of items that sit in different 64-item rounds, is requested
// several times, and what misses the L1 is fetched again from L2 / HBM (C5: 519 line fetches per frame for 392
// distinct lines touched; the whole 60 000-byte frame is 469).  Here each frame is staged ONCE: the 16-byte windows
// that hold a touched atom (the same greedy cover the specialised lane kernel uses, computed at plan creation) are
// gathered by LDS-DMA into a compact image - lane l of instruction i copies window 64 i + l - by loader waves that
// do nothing else, and consumer waves (one wave per frame, the arithmetic of frames_wave_kernel unchanged) read
// their atoms from the image at the LDS positions the plan precomputed.  Hand-off as in molann_lane_jit.inc:
// monotonic counters ready[s] / done[s] per slot, tiles taken in order from a block-wide counter; a consumer keeps
// its slot for the whole frame (it computes from the image) and hands it back when the features are stored.
// ND = LDS-DMA instructions per frame (a compile-time count: the loader's counted vmcnt waits need an immediate);
// windows beyond the plan's count re-copy window 0 into the image's padding.
struct RingArgs {
    long n_frames;
    int frame_bytes;   // 12 n_inp
    int n_align, n_items, out_cols;
    int n_win;         // windows staged per frame (<= 64 ND)
    int n_slot, n_cons, n_load, depth;
    int nt;            // loaders' gathers non-temporal
};
constexpr int RING_HEADER = 256;

template <int K>
__device__ __forceinline__ void ring_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K < 64 ? K : 0) : "memory"); }
template <int ND>
__device__ __forceinline__ void ring_wait_frames(int k) { // at most k frames (ND operations each) of this wave in flight
    switch (k) {
    case 0: ring_wait_vmcnt<0>(); break;
    case 1: ring_wait_vmcnt<ND>(); break;
    case 2: ring_wait_vmcnt<2 * ND>(); break;
    case 3: ring_wait_vmcnt<3 * ND>(); break;
    case 4: ring_wait_vmcnt<4 * ND>(); break;
    case 5: ring_wait_vmcnt<5 * ND>(); break;
    default: ring_wait_vmcnt<6 * ND>(); break;
    }
}
// LDS-DMA gather of 16 bytes per lane, saddr form: wave-uniform 64-bit base + the lane's zero-extended 32-bit offset ->
// LDS (wave-uniform address in M0) + 16 * lane.  Inline asm: hipcc picks the vaddr form for `base + offset[i]` (a 64-bit
// vector add per instruction on the loader's issue path); M0 is written where it is read and restored.  The loader
// counts these operations itself (vmcnt).
// NT: non-temporal (the frame is read once: tools/micro/subline.hip moves whole 128-byte lines 12 % faster with it).
template <bool NT>
__device__ __forceinline__ void ring_dma16(const void* base_uniform, unsigned lane_off, unsigned lds_uniform) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(base_uniform), "s"(lds_uniform)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(lane_off), "s"(base_uniform), "s"(lds_uniform)
                     : "memory");
}
typedef __attribute__((address_space(3))) volatile int* ring_word_t;
__device__ __forceinline__ int ring_peek(ring_word_t w) { return __builtin_amdgcn_readfirstlane(*w); }

__device__ __forceinline__ V3 img_atom(const float* img, int pos) { return v3(img[pos], img[pos + 1], img[pos + 2]); }

// (Measured and dropped: consumers that first copy their lanes' atoms to registers and hand the slot back at once -
// 63 more VGPRs beside the fp64 covariance; C4 9.7 against 7.4 us per 1000 frames, C5 12.0 against 9.6.)
template <int ND>
__global__ __launch_bounds__(1024) void frames_ring_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           const int* __restrict__ win_off, const int* __restrict__ align_pos,
                                                           const float* __restrict__ ref, const double* __restrict__ ref64,
                                                           const ItemDev* __restrict__ items_pos, RingArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ring_word_t ready = (ring_word_t)(smem);        // [n_slot <= 16]
    ring_word_t done = (ring_word_t)(smem + 64);
    ring_word_t next = (ring_word_t)(smem + 128);
    unsigned char* const ring = smem + RING_HEADER;
    constexpr int IMG_BYTES = ND * 1024;

    // frames of this block: f(n) = blockIdx.x + n * gridDim.x
    const long f_step = (long)gridDim.x;
    const int n_b = (long)blockIdx.x < a.n_frames ? (int)((a.n_frames - 1 - (long)blockIdx.x) / f_step) + 1 : 0;
    if (threadIdx.x < 33) ((volatile int*)smem)[threadIdx.x] = 0;
    __syncthreads();

    if (wave >= a.n_cons) {
        // =========================== loader ==========================================================
        __builtin_amdgcn_s_setprio(3);
        unsigned goff[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) { const int w = 64 * i + lane; goff[i] = (unsigned)win_off[w < a.n_win ? w : 0]; }
        const int jl = wave - a.n_cons;
        int n_issue = jl, n_pub = jl, inflight = 0;
        while (n_pub < n_b) {
            bool can_issue = n_issue < n_b && inflight <= a.depth;
            if (can_issue) {
                const int s = n_issue % a.n_slot, gen = n_issue / a.n_slot;
                if (gen > 0 && ring_peek(done + s) < gen) {
                    can_issue = false;
                    if (inflight == 0) { __builtin_amdgcn_s_sleep(1); continue; }
                }
            }
            if (can_issue) {
                const long f = (long)blockIdx.x + (long)n_issue * f_step;
                const unsigned char* gsrc = (const unsigned char*)x + f * (long)a.frame_bytes;
                unsigned char* slot = ring + (size_t)(n_issue % a.n_slot) * IMG_BYTES;
                const unsigned slot_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)slot);
                if (a.nt) {
#pragma unroll
                    for (int i = 0; i < ND; ++i) ring_dma16<true>(gsrc, goff[i], slot_lds + i * 1024);
                } else {
#pragma unroll
                    for (int i = 0; i < ND; ++i) ring_dma16<false>(gsrc, goff[i], slot_lds + i * 1024);
                }
                n_issue += a.n_load;
                ++inflight;
            } else {
                ring_wait_frames<ND>(inflight - 1);
                ready[n_pub % a.n_slot] = n_pub / a.n_slot + 1;
                n_pub += a.n_load;
                --inflight;
            }
        }
        return;
    }

    // =============================== consumers: one wave per frame ====================================
    const auto refc = as_const(ref);
    const bool has_align = a.n_align > 0;
    for (;;) {
        int n = 0;
        if (lane == 0) n = __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        n = __builtin_amdgcn_readfirstlane(n);
        if (n >= n_b) break;
        const long f = (long)blockIdx.x + (long)n * f_step;
        const int s = n % a.n_slot, gen = n / a.n_slot;
        while (ring_peek(ready + s) < gen + 1) __builtin_amdgcn_s_sleep(1);
        const float* img = (const float*)(ring + (size_t)s * IMG_BYTES);

        V3 c0 = v3(0.f, 0.f, 0.f), dl = v3(0.f, 0.f, 0.f);
        if (has_align) c0 = img_atom(img, as_const(align_pos)[0]);
        float R[9];
        if (has_align) {
            float sx = 0.f, sy = 0.f, sz = 0.f, g = 0.f;
            double h[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
            auto acc = [&](int i, V3 atom) {
                const double rx = ref64[3 * i], ry = ref64[3 * i + 1], rz = ref64[3 * i + 2];
                const V3 p = atom - c0;
                sx += p.x; sy += p.y; sz += p.z;
                g = fmaf(p.x, p.x, fmaf(p.y, p.y, fmaf(p.z, p.z, g)));
                const double px = p.x, py = p.y, pz = p.z;
                h[0] = fma(px, rx, h[0]); h[1] = fma(px, ry, h[1]); h[2] = fma(px, rz, h[2]);
                h[3] = fma(py, rx, h[3]); h[4] = fma(py, ry, h[4]); h[5] = fma(py, rz, h[5]);
                h[6] = fma(pz, rx, h[6]); h[7] = fma(pz, ry, h[7]); h[8] = fma(pz, rz, h[8]);
            };
#pragma unroll 4
            for (int i = lane; i < a.n_align; i += 64) acc(i, img_atom(img, align_pos[i]));
            sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); g = wave_sum(g);
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = wave_sum(h[i]);
            const int cb = 3 * a.n_align;
            const auto r64c = as_const(ref64);
            const double srx = r64c[cb], sry = r64c[cb + 1], srz = r64c[cb + 2], gref = r64c[cb + 3];
            const float inv_a = refc[cb + 4], fa = refc[cb + 5];
            dl = v3(sx * inv_a, sy * inv_a, sz * inv_a);
            const double dx = dl.x, dy = dl.y, dz = dl.z;
            h[0] = fma(-dx, srx, h[0]); h[1] = fma(-dx, sry, h[1]); h[2] = fma(-dx, srz, h[2]);
            h[3] = fma(-dy, srx, h[3]); h[4] = fma(-dy, sry, h[4]); h[5] = fma(-dy, srz, h[5]);
            h[6] = fma(-dz, srx, h[6]); h[7] = fma(-dz, sry, h[7]); h[8] = fma(-dz, srz, h[8]);
            const float gp = fmaxf(g - fa * dot(dl, dl), 0.f);
            kabsch_rotation(h, 0.5 * ((double)gp + gref) * 1.0001, R);
        }
        float* of = out + f * (long)a.out_cols;
        auto item = [&](int type, int col, V3 p0, V3 p1, V3 p2, V3 p3) {
            if (has_align) align_item_atoms(type, p0, p1, p2, p3, c0, dl, R);
            float v[3];
            const int w = eval_item(type, p0, p1, p2, p3, v);
            of[col] = v[0];
            if (w > 1) of[col + 1] = v[1];
            if (w > 2) of[col + 2] = v[2];
        };
#pragma unroll 2
        for (int it = lane; it < a.n_items; it += 64) {
            const int4 d0 = ((const int4*)items_pos)[2 * it];
            const int2 d1 = ((const int2*)items_pos)[4 * it + 2];
            item(d0.x, d0.y, img_atom(img, d0.z), img_atom(img, d0.w), img_atom(img, d1.x), img_atom(img, d1.y));
        }
        // every lane's reads of the image have returned before the slot goes back to the loaders
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        done[s] = gen + 1;
    }
}

#include "molann_align_regs.inc"

// =============================================================================================
// float64 path: the same forward for `model.double()(x.double())` (the reference follows x.dtype, ann.py:187-197)
// =============================================================================================
// One wave per frame, any frame size, everything in double (covariance, quaternion solve, rotation, features, MLP):
// written for agreement with the reference's float64 run to rounding (tests: 1e-10), not for speed - float64 is the
// reference's validation mode, float32 its production mode.  mode 0: features -> out[f][out_cols]; mode 1: aligned
// coordinates -> out[f][n_inp][3].  The alignment reference comes from the plan's float64 copy (d_ref64: the centred
// coordinates, their sums and the two constants, see pack_ref_kernel / molann_plan_update_ref_f64).
struct F64Args {
    long n_frames;
    int n_inp, n_align, n_items, out_cols, mode;
};
__device__ __forceinline__ V3d load_atom_f64(const double* __restrict__ xf, int k) { return v3d(xf[3 * k], xf[3 * k + 1], xf[3 * k + 2]); }

__global__ __launch_bounds__(256) void frames_f64_kernel(const double* __restrict__ x, double* __restrict__ out,
                                                         const int* __restrict__ align_idx, const double* __restrict__ ref64,
                                                         const ItemDev* __restrict__ items, F64Args a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const long frame_dw = 3l * a.n_inp;
    for (long f = (long)blockIdx.x * wpb + wave; f < a.n_frames; f += (long)gridDim.x * wpb) {
        const double* xf = x + f * frame_dw;
        double R[9] = {1., 0., 0., 0., 1., 0., 0., 0., 1.};
        V3d c = v3d(0., 0., 0.);
        if (a.n_align > 0) {
            // centroid of the alignment atoms (ann.py:181), then H = sum (p - c) ref^T (ann.py:183-187), both as wave sums
            double sx = 0., sy = 0., sz = 0.;
            for (int i = lane; i < a.n_align; i += 64) { const V3d p = load_atom_f64(xf, align_idx[i]); sx += p.x; sy += p.y; sz += p.z; }
            const double inv_a = 1.0 / (double)a.n_align;
            c = v3d(wave_sum(sx) * inv_a, wave_sum(sy) * inv_a, wave_sum(sz) * inv_a);
            double h[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, g = 0.;
            for (int i = lane; i < a.n_align; i += 64) {
                const double rx = ref64[3 * i], ry = ref64[3 * i + 1], rz = ref64[3 * i + 2];
                const V3d p = load_atom_f64(xf, align_idx[i]) - c;
                g = fma(p.x, p.x, fma(p.y, p.y, fma(p.z, p.z, g)));
                h[0] = fma(p.x, rx, h[0]); h[1] = fma(p.x, ry, h[1]); h[2] = fma(p.x, rz, h[2]);
                h[3] = fma(p.y, rx, h[3]); h[4] = fma(p.y, ry, h[4]); h[5] = fma(p.y, rz, h[5]);
                h[6] = fma(p.z, rx, h[6]); h[7] = fma(p.z, ry, h[7]); h[8] = fma(p.z, rz, h[8]);
            }
            g = wave_sum(g);
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = wave_sum(h[i]);
            const double gref = ref64[3 * a.n_align + 3];
            kabsch_rotation_t<double, double>(h, 0.5 * (g + gref) * 1.0001, R);
        }
        if (a.mode == 1) {
            double* of = out + f * frame_dw;
            for (int k = lane; k < a.n_inp; k += 64) {
                const V3d y = rotate(load_atom_f64(xf, k) - c, R);
                of[3 * k] = y.x; of[3 * k + 1] = y.y; of[3 * k + 2] = y.z;
            }
            continue;
        }
        double* of = out + f * (long)a.out_cols;
        for (int it = lane; it < a.n_items; it += 64) {
            const ItemDev d = items[it];
            V3d p0 = load_atom_f64(xf, d.idx[0]), p1 = load_atom_f64(xf, d.idx[1]), p2 = load_atom_f64(xf, d.idx[2]), p3 = load_atom_f64(xf, d.idx[3]);
            if (a.n_align > 0) { p0 = rotate(p0 - c, R); p1 = rotate(p1 - c, R); p2 = rotate(p2 - c, R); p3 = rotate(p3 - c, R); }
            double v[3];
            const int w = eval_item_f64(d.type, p0, p1, p2, p3, v);
            of[d.col] = v[0];
            if (w > 1) of[d.col + 1] = v[1];
            if (w > 2) of[d.col + 2] = v[2];
        }
    }
}

// dL/dx of frames_f64_kernel's features, in double: the structure of frames_wave_bwd_kernel (one wave per frame, the frame's
// gradient row zeroed, every contribution a double atomic into it) on the float64 forward's own formulas (centroid, then
// H = sum (p - c) ref^T).  The reference differentiates its float64 forward with autograd.
__global__ __launch_bounds__(256) void frames_bwd_f64_kernel(const double* __restrict__ x, const double* __restrict__ gout, double* __restrict__ gx,
                                                             const int* __restrict__ align_idx, const double* __restrict__ ref64,
                                                             const ItemDev* __restrict__ items, F64Args a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const long frame_dw = 3l * a.n_inp;
    const bool has_align = a.n_align > 0;
    for (long f = (long)blockIdx.x * wpb + wave; f < a.n_frames; f += (long)gridDim.x * wpb) {
        const double* xf = x + f * frame_dw;
        double* gxf = gx + f * frame_dw;
        const double* gf = gout + f * (long)a.out_cols;
        for (long c = lane; c < frame_dw; c += 64) gxf[c] = 0.0;
        double R[9] = {1., 0., 0., 0., 1., 0., 0., 0., 1.};
        double h[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
        V3d c = v3d(0., 0., 0.);
        if (has_align) {
            double sx = 0., sy = 0., sz = 0.;
            for (int i = lane; i < a.n_align; i += 64) { const V3d p = load_atom_f64(xf, align_idx[i]); sx += p.x; sy += p.y; sz += p.z; }
            const double inv_a = 1.0 / (double)a.n_align;
            c = v3d(wave_sum(sx) * inv_a, wave_sum(sy) * inv_a, wave_sum(sz) * inv_a);
            double g = 0.;
            for (int i = lane; i < a.n_align; i += 64) {
                const double rx = ref64[3 * i], ry = ref64[3 * i + 1], rz = ref64[3 * i + 2];
                const V3d p = load_atom_f64(xf, align_idx[i]) - c;
                g = fma(p.x, p.x, fma(p.y, p.y, fma(p.z, p.z, g)));
                h[0] = fma(p.x, rx, h[0]); h[1] = fma(p.x, ry, h[1]); h[2] = fma(p.x, rz, h[2]);
                h[3] = fma(p.y, rx, h[3]); h[4] = fma(p.y, ry, h[4]); h[5] = fma(p.y, rz, h[5]);
                h[6] = fma(p.z, rx, h[6]); h[7] = fma(p.z, ry, h[7]); h[8] = fma(p.z, rz, h[8]);
            }
            g = wave_sum(g);
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = wave_sum(h[i]);
            kabsch_rotation_t<double, double>(h, 0.5 * (g + ref64[3 * a.n_align + 3]) * 1.0001, R);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the zero stores are acknowledged before the atomics below
        double GR[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
        V3d gsum = v3d(0., 0., 0.);
        for (int it = lane; it < a.n_items; it += 64) {
            const ItemDev d = items[it];
            V3d pc[4], y[4], gy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pc[j] = load_atom_f64(xf, d.idx[j]);
                if (has_align) pc[j] = pc[j] - c;
                y[j] = has_align ? rotate(pc[j], R) : pc[j];
                gy[j] = v3d(0., 0., 0.);
            }
            const int w = item_width(d.type);
            const double g3[3] = {gf[d.col], w > 1 ? gf[d.col + 1] : 0.0, w > 2 ? gf[d.col + 2] : 0.0};
            eval_item_backward_f64(d.type, y[0], y[1], y[2], y[3], g3, gy[0], gy[1], gy[2], gy[3]);
            const int na = item_atoms(d.type);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < na) {
                    const V3d g = gy[j];
                    V3d gp = g;
                    if (has_align) { // y = pc R :  G_R += pc^T g ,  g_p = g R^T
                        GR[0] = fma(pc[j].x, g.x, GR[0]); GR[1] = fma(pc[j].x, g.y, GR[1]); GR[2] = fma(pc[j].x, g.z, GR[2]);
                        GR[3] = fma(pc[j].y, g.x, GR[3]); GR[4] = fma(pc[j].y, g.y, GR[4]); GR[5] = fma(pc[j].y, g.z, GR[5]);
                        GR[6] = fma(pc[j].z, g.x, GR[6]); GR[7] = fma(pc[j].z, g.y, GR[7]); GR[8] = fma(pc[j].z, g.z, GR[8]);
                        gp = v3d(fma(g.z, R[2], fma(g.y, R[1], g.x * R[0])), fma(g.z, R[5], fma(g.y, R[4], g.x * R[3])),
                                 fma(g.z, R[8], fma(g.y, R[7], g.x * R[6])));
                        gsum = gsum + gp;
                    }
                    double* dst = gxf + 3 * d.idx[j];
                    atomicAdd(dst, gp.x); atomicAdd(dst + 1, gp.y); atomicAdd(dst + 2, gp.z);
                }
            }
        }
        if (has_align) {
#pragma unroll
            for (int i = 0; i < 9; ++i) GR[i] = wave_sum(GR[i]);
            gsum = v3d(wave_sum(gsum.x), wave_sum(gsum.y), wave_sum(gsum.z));
            double GH[9];
            kabsch_rotation_backward_t<double, double>(h, R, GR, GH);
            const double inv_a = 1.0 / (double)a.n_align;
            // H = sum_i (a_i - c) ref_i^T also depends on c through every p_i: - G_H (sum_j ref_j) / a per align atom.  The
            // reference centres ref_x in float32, so in double the sum is ~1e-7, not 0: visible at this path's 1e-9 bar.
            const double srx = ref64[3 * a.n_align], sry = ref64[3 * a.n_align + 1], srz = ref64[3 * a.n_align + 2];
            const V3d cen = v3d(inv_a * (gsum.x + fma(GH[2], srz, fma(GH[1], sry, GH[0] * srx))),
                                inv_a * (gsum.y + fma(GH[5], srz, fma(GH[4], sry, GH[3] * srx))),
                                inv_a * (gsum.z + fma(GH[8], srz, fma(GH[7], sry, GH[6] * srx))));
            for (int i = lane; i < a.n_align; i += 64) {
                const double rx = ref64[3 * i], ry = ref64[3 * i + 1], rz = ref64[3 * i + 2];
                double* dst = gxf + 3 * align_idx[i];
                atomicAdd(dst, fma(GH[2], rz, fma(GH[1], ry, GH[0] * rx)) - cen.x);
                atomicAdd(dst + 1, fma(GH[5], rz, fma(GH[4], ry, GH[3] * rx)) - cen.y);
                atomicAdd(dst + 2, fma(GH[8], rz, fma(GH[7], ry, GH[6] * rx)) - cen.z);
            }
        }
    }
}

// ann_layers in double: one wave per frame, the activations ping-pong between two LDS rows, lane j computes units j,
// j + 64, ... of a layer as one fma chain over the inputs (weights read from the caller's tensors as they are: the
// torch.nn.Linear layout W[J][K], b[J]).
struct F64Mlp {
    int n_layers, act, max_w;
    int dims[MOLANN_MAX_LAYERS + 1];
    const double* W[MOLANN_MAX_LAYERS];
    const double* b[MOLANN_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void mlp_f64_kernel(const double* __restrict__ in, double* __restrict__ out, long n_frames, F64Mlp m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    double* buf0 = (double*)smem + (size_t)wave * 2 * m.max_w;
    double* buf1 = buf0 + m.max_w;
    for (long f = (long)blockIdx.x * wpb + wave; f < n_frames; f += (long)gridDim.x * wpb) {
        for (int k = lane; k < m.dims[0]; k += 64) buf0[k] = in[f * (long)m.dims[0] + k];
        double* cur = buf0;
        double* nxt = buf1;
        for (int l = 0; l < m.n_layers; ++l) {
            const int K = m.dims[l], J = m.dims[l + 1];
            const bool last = l + 1 == m.n_layers;
            for (int j = lane; j < J; j += 64) {
                const double* w = m.W[l] + (long)j * K;
                double acc = m.b[l][j];
                for (int k = 0; k < K; ++k) acc = fma(w[k], cur[k], acc);
                if (last) out[f * (long)J + j] = acc;
                else nxt[j] = apply_activation_f64(m.act, acc);
            }
            double* t = cur; cur = nxt; nxt = t;
        }
    }
}

// =============================================================================================
// dst[i] += sum over rows of part[row][i]: the per-block parameter sums of molann_mlp_bwd (one writer per element).
// Block = 64 elements x 16 row groups, so a thread's chain of loads is n_rows / 16 long.
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ part, int n_rows, int n, float* __restrict__ dst) {
    __shared__ float acc[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + x;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int r = y;
        for (; r + 16 < n_rows; r += 32) { s0 += part[(long)r * n + i]; s1 += part[(long)(r + 16) * n + i]; }
        if (r < n_rows) s0 += part[(long)r * n + i];
    }
    acc[y][x] = s0 + s1;
    __syncthreads();
    if (y == 0 && i < n) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += acc[g][x];
        dst[i] += s;
    }
}

// frames_wave_bwd_kernel: dL/dx of frames_wave_kernel (features of large frames), one wave per frame
// =============================================================================================
// grad_out[f][d_feat] -> grad_x[f][n_inp][3].  The frame's gradient row is zeroed with coalesced stores, then
// every contribution is a float atomic into it (an atom may sit in several items and in the alignment set).
//   items (lanes):  y_j = ((p_j - c0) - dl) R ;  g_y from eval_item_backward ;  g_p = g_y R^T ;
//                   G_R += (p_j - c)^T g_y ;  g_sum += g_p
//   wave:           G_R, g_sum reduced ;  G_H = kabsch_rotation_backward(H, R, G_R)
//   align atoms:    g_p[i] += G_H ref_i - g_sum / a        (H = sum_i p_i ref_i^T, c = mean of the align atoms)
// The forward quantities (c, H, R) are recomputed exactly as frames_wave_kernel computes them.
__global__ __launch_bounds__(256) void frames_wave_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                              float* __restrict__ gx, const int* __restrict__ align_idx,
                                                              const float* __restrict__ ref, const double* __restrict__ ref64,
                                                              const ItemDev* __restrict__ items, PreArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const auto refc = as_const(ref);
    const bool has_align = a.n_align > 0;

    for (long f = (long)blockIdx.x * wpb + wave; f < a.n_frames; f += (long)gridDim.x * wpb) {
        const float* xf = x + f * (long)a.frame_dw;
        float* gxf = gx + f * (long)a.frame_dw;
        const float* gf = gout + f * (long)a.out_cols;
        // ---- 0. zero this frame's gradient row; the stores must have landed before the atomics below
        if (a.out_wide && (a.frame_dw & 3) == 0) {
            for (int c = lane; c < (a.frame_dw >> 2); c += 64) ((f32x4*)gxf)[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
            for (int c = lane; c < a.frame_dw; c += 64) gxf[c] = 0.f;
        }
        // ---- 1. forward recompute: centre, covariance, rotation
        float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        double h[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
        V3 c0 = v3(0.f, 0.f, 0.f), dl = v3(0.f, 0.f, 0.f);
        if (has_align) {
            const int k0 = as_const(align_idx)[0];
            c0 = load_atom(xf, k0);
            float sx = 0.f, sy = 0.f, sz = 0.f, g = 0.f;
#pragma unroll 4
            for (int i = lane; i < a.n_align; i += 64) {
                const int k = align_idx[i];
                const double rx = ref64[3 * i], ry = ref64[3 * i + 1], rz = ref64[3 * i + 2];
                const V3 p = load_atom(xf, k) - c0;
                sx += p.x; sy += p.y; sz += p.z;
                g = fmaf(p.x, p.x, fmaf(p.y, p.y, fmaf(p.z, p.z, g)));
                const double px = p.x, py = p.y, pz = p.z;
                h[0] = fma(px, rx, h[0]); h[1] = fma(px, ry, h[1]); h[2] = fma(px, rz, h[2]);
                h[3] = fma(py, rx, h[3]); h[4] = fma(py, ry, h[4]); h[5] = fma(py, rz, h[5]);
                h[6] = fma(pz, rx, h[6]); h[7] = fma(pz, ry, h[7]); h[8] = fma(pz, rz, h[8]);
            }
            sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); g = wave_sum(g);
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = wave_sum(h[i]);
            const int cb = 3 * a.n_align;
            const auto r64c = as_const(ref64);
            const double srx = r64c[cb], sry = r64c[cb + 1], srz = r64c[cb + 2], gref = r64c[cb + 3];
            const float inv_a = refc[cb + 4], fa = refc[cb + 5];
            dl = v3(sx * inv_a, sy * inv_a, sz * inv_a);
            const double dx = dl.x, dy = dl.y, dz = dl.z;
            h[0] = fma(-dx, srx, h[0]); h[1] = fma(-dx, sry, h[1]); h[2] = fma(-dx, srz, h[2]);
            h[3] = fma(-dy, srx, h[3]); h[4] = fma(-dy, sry, h[4]); h[5] = fma(-dy, srz, h[5]);
            h[6] = fma(-dz, srx, h[6]); h[7] = fma(-dz, sry, h[7]); h[8] = fma(-dz, srz, h[8]);
            const float gp = fmaxf(g - fa * dot(dl, dl), 0.f);
            kabsch_rotation(h, 0.5 * ((double)gp + gref) * 1.0001, R);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the zero stores are acknowledged

        // ---- 2. items
        float GR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        V3 gsum = v3(0.f, 0.f, 0.f);
        for (int it = lane; it < a.n_items; it += 64) {
            const int4 d0 = ((const int4*)items)[2 * it];
            const int2 d1 = ((const int2*)items)[4 * it + 2];
            const int type = d0.x, col = d0.y;
            const int idx[4] = {d0.z, d0.w, d1.x, d1.y};
            V3 pc[4], y[4], gy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pc[j] = load_atom(xf, idx[j]);
                if (has_align) pc[j] = (pc[j] - c0) - dl;
                y[j] = has_align ? rotate(pc[j], R) : pc[j];
                gy[j] = v3(0.f, 0.f, 0.f);
            }
            const int w = item_width(type);
            float g3[3] = {gf[col], w > 1 ? gf[col + 1] : 0.f, w > 2 ? gf[col + 2] : 0.f};
            eval_item_backward(type, y[0], y[1], y[2], y[3], g3, gy[0], gy[1], gy[2], gy[3]);
            const int na = item_atoms(type);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < na) {
                    const V3 g = gy[j];
                    V3 gp = g;
                    if (has_align) { // y = pc R :  G_R += pc^T g ,  g_p = g R^T
                        GR[0] = fmaf(pc[j].x, g.x, GR[0]); GR[1] = fmaf(pc[j].x, g.y, GR[1]); GR[2] = fmaf(pc[j].x, g.z, GR[2]);
                        GR[3] = fmaf(pc[j].y, g.x, GR[3]); GR[4] = fmaf(pc[j].y, g.y, GR[4]); GR[5] = fmaf(pc[j].y, g.z, GR[5]);
                        GR[6] = fmaf(pc[j].z, g.x, GR[6]); GR[7] = fmaf(pc[j].z, g.y, GR[7]); GR[8] = fmaf(pc[j].z, g.z, GR[8]);
                        gp = v3(fmaf(g.z, R[2], fmaf(g.y, R[1], g.x * R[0])), fmaf(g.z, R[5], fmaf(g.y, R[4], g.x * R[3])),
                                fmaf(g.z, R[8], fmaf(g.y, R[7], g.x * R[6])));
                        gsum = gsum + gp;
                    }
                    float* dst = gxf + 3 * idx[j];
                    atomicAdd(dst, gp.x); atomicAdd(dst + 1, gp.y); atomicAdd(dst + 2, gp.z);
                }
            }
        }
        // ---- 3. rotation and centring backward through the alignment atoms
        if (has_align) {
#pragma unroll
            for (int i = 0; i < 9; ++i) GR[i] = wave_sum(GR[i]);
            gsum = v3(wave_sum(gsum.x), wave_sum(gsum.y), wave_sum(gsum.z));
            float GH[9];
            kabsch_rotation_backward(h, R, GR, GH);
            const float inv_a = refc[3 * a.n_align + 4];
            const V3 gcen = inv_a * gsum;
            for (int i = lane; i < a.n_align; i += 64) {
                const int k = align_idx[i];
                const float rx = ref[3 * i], ry = ref[3 * i + 1], rz = ref[3 * i + 2];
                const V3 add = v3(fmaf(GH[2], rz, fmaf(GH[1], ry, GH[0] * rx)), fmaf(GH[5], rz, fmaf(GH[4], ry, GH[3] * rx)),
                                  fmaf(GH[8], rz, fmaf(GH[7], ry, GH[6] * rx))) - gcen;
                float* dst = gxf + 3 * k;
                atomicAdd(dst, add.x); atomicAdd(dst + 1, add.y); atomicAdd(dst + 2, add.z);
            }
        }
    }
}

// frames_wave_bwd_gather_kernel: the same gradient without atomics (they were 9/10 of frames_wave_bwd_kernel's time: C4 80 us per
// 1000 frames against 8 for the forward with the same scattered loads).  Items (lanes) leave their atoms' gradients g_y in the
// wave's LDS buffer; then every TOUCHED ATOM (lanes again) adds up its own contributions - the lists are made at plan creation
// (bw_atoms: touched atoms; bw_ptr / bw_list: CSR of (item, atom-of-item) pairs per touched atom; bw_align: its place in the
// alignment set or -1) - rotates the sum back once, adds the alignment terms and stores its three floats behind the zero-fill.
struct BwGatherArgs { int n_touched, lds_per_wave; };
__global__ __launch_bounds__(256) void frames_wave_bwd_gather_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                                     float* __restrict__ gx, const int* __restrict__ align_idx,
                                                                     const float* __restrict__ ref, const double* __restrict__ ref64,
                                                                     const ItemDev* __restrict__ items, const int* __restrict__ bw_atoms,
                                                                     const int* __restrict__ bw_ptr, const int* __restrict__ bw_list,
                                                                     const int* __restrict__ bw_align, PreArgs a, BwGatherArgs b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    float* gyb = (float*)(smem + (size_t)wave * b.lds_per_wave);   // [item][atom of item][xyz]
    const auto refc = as_const(ref);
    const bool has_align = a.n_align > 0;

    for (long f = (long)blockIdx.x * wpb + wave; f < a.n_frames; f += (long)gridDim.x * wpb) {
        const float* xf = x + f * (long)a.frame_dw;
        float* gxf = gx + f * (long)a.frame_dw;
        const float* gf = gout + f * (long)a.out_cols;
        // ---- 0. zero this frame's gradient row
        if (a.out_wide && (a.frame_dw & 3) == 0) {
            for (int c = lane; c < (a.frame_dw >> 2); c += 64) ((f32x4*)gxf)[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
            for (int c = lane; c < a.frame_dw; c += 64) gxf[c] = 0.f;
        }
        // ---- 1. forward recompute: centre, covariance, rotation (as frames_wave_bwd_kernel)
        float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        double h[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
        V3 c0 = v3(0.f, 0.f, 0.f), dl = v3(0.f, 0.f, 0.f);
        if (has_align) {
            const int k0 = as_const(align_idx)[0];
            c0 = load_atom(xf, k0);
            float sx = 0.f, sy = 0.f, sz = 0.f, g = 0.f;
#pragma unroll 4
            for (int i = lane; i < a.n_align; i += 64) {
                const int k = align_idx[i];
                const double rx = ref64[3 * i], ry = ref64[3 * i + 1], rz = ref64[3 * i + 2];
                const V3 p = load_atom(xf, k) - c0;
                sx += p.x; sy += p.y; sz += p.z;
                g = fmaf(p.x, p.x, fmaf(p.y, p.y, fmaf(p.z, p.z, g)));
                const double px = p.x, py = p.y, pz = p.z;
                h[0] = fma(px, rx, h[0]); h[1] = fma(px, ry, h[1]); h[2] = fma(px, rz, h[2]);
                h[3] = fma(py, rx, h[3]); h[4] = fma(py, ry, h[4]); h[5] = fma(py, rz, h[5]);
                h[6] = fma(pz, rx, h[6]); h[7] = fma(pz, ry, h[7]); h[8] = fma(pz, rz, h[8]);
            }
            sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); g = wave_sum(g);
#pragma unroll
            for (int i = 0; i < 9; ++i) h[i] = wave_sum(h[i]);
            const int cb = 3 * a.n_align;
            const auto r64c = as_const(ref64);
            const double srx = r64c[cb], sry = r64c[cb + 1], srz = r64c[cb + 2], gref = r64c[cb + 3];
            const float inv_a = refc[cb + 4], fa = refc[cb + 5];
            dl = v3(sx * inv_a, sy * inv_a, sz * inv_a);
            const double dx = dl.x, dy = dl.y, dz = dl.z;
            h[0] = fma(-dx, srx, h[0]); h[1] = fma(-dx, sry, h[1]); h[2] = fma(-dx, srz, h[2]);
            h[3] = fma(-dy, srx, h[3]); h[4] = fma(-dy, sry, h[4]); h[5] = fma(-dy, srz, h[5]);
            h[6] = fma(-dz, srx, h[6]); h[7] = fma(-dz, sry, h[7]); h[8] = fma(-dz, srz, h[8]);
            const float gp = fmaxf(g - fa * dot(dl, dl), 0.f);
            kabsch_rotation(h, 0.5 * ((double)gp + gref) * 1.0001, R);
        }
        // ---- 2. items: g_y of their atoms -> LDS;  G_R += p^T g_y,  sum of g_y
        float GR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        V3 gysum = v3(0.f, 0.f, 0.f);
        for (int it = lane; it < a.n_items; it += 64) {
            const int4 d0 = ((const int4*)items)[2 * it];
            const int2 d1 = ((const int2*)items)[4 * it + 2];
            const int type = d0.x, col = d0.y;
            const int idx[4] = {d0.z, d0.w, d1.x, d1.y};
            V3 pc[4], y[4], gy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pc[j] = load_atom(xf, idx[j]);
                if (has_align) pc[j] = (pc[j] - c0) - dl;
                y[j] = has_align ? rotate(pc[j], R) : pc[j];
                gy[j] = v3(0.f, 0.f, 0.f);
            }
            const int w = item_width(type);
            float g3[3] = {gf[col], w > 1 ? gf[col + 1] : 0.f, w > 2 ? gf[col + 2] : 0.f};
            eval_item_backward(type, y[0], y[1], y[2], y[3], g3, gy[0], gy[1], gy[2], gy[3]);
            const int na = item_atoms(type);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const V3 g = j < na ? gy[j] : v3(0.f, 0.f, 0.f);
                float* dst = gyb + (4 * it + j) * 3;
                dst[0] = g.x; dst[1] = g.y; dst[2] = g.z;
                if (has_align && j < na) {
                    GR[0] = fmaf(pc[j].x, g.x, GR[0]); GR[1] = fmaf(pc[j].x, g.y, GR[1]); GR[2] = fmaf(pc[j].x, g.z, GR[2]);
                    GR[3] = fmaf(pc[j].y, g.x, GR[3]); GR[4] = fmaf(pc[j].y, g.y, GR[4]); GR[5] = fmaf(pc[j].y, g.z, GR[5]);
                    GR[6] = fmaf(pc[j].z, g.x, GR[6]); GR[7] = fmaf(pc[j].z, g.y, GR[7]); GR[8] = fmaf(pc[j].z, g.z, GR[8]);
                    gysum = gysum + g;
                }
            }
        }
        // ---- 3. rotation backward; the centroid's share
        float GH[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        V3 gcen = v3(0.f, 0.f, 0.f);
        if (has_align) {
#pragma unroll
            for (int i = 0; i < 9; ++i) GR[i] = wave_sum(GR[i]);
            gysum = v3(wave_sum(gysum.x), wave_sum(gysum.y), wave_sum(gysum.z));
            kabsch_rotation_backward(h, R, GR, GH);
            const V3 gs = v3(fmaf(gysum.z, R[2], fmaf(gysum.y, R[1], gysum.x * R[0])), fmaf(gysum.z, R[5], fmaf(gysum.y, R[4], gysum.x * R[3])),
                             fmaf(gysum.z, R[8], fmaf(gysum.y, R[7], gysum.x * R[6])));   // sum of g_p = (sum of g_y) R^T
            gcen = refc[3 * a.n_align + 4] * gs;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // the zero stores are acknowledged, the g_y are in LDS
        // ---- 4. touched atoms: own contributions summed, rotated back, alignment terms, plain stores
        for (int t = lane; t < b.n_touched; t += 64) {
            V3 g = v3(0.f, 0.f, 0.f);
            const int k1 = bw_ptr[t + 1];
            for (int k = bw_ptr[t]; k < k1; ++k) {
                const float* src = gyb + 3 * bw_list[k];
                g = g + v3(src[0], src[1], src[2]);
            }
            V3 gp = g;
            if (has_align) {
                gp = v3(fmaf(g.z, R[2], fmaf(g.y, R[1], g.x * R[0])), fmaf(g.z, R[5], fmaf(g.y, R[4], g.x * R[3])),
                        fmaf(g.z, R[8], fmaf(g.y, R[7], g.x * R[6])));
                const int i = bw_align[t];
                if (i >= 0) {
                    const float rx = ref[3 * i], ry = ref[3 * i + 1], rz = ref[3 * i + 2];
                    gp = gp + (v3(fmaf(GH[2], rz, fmaf(GH[1], ry, GH[0] * rx)), fmaf(GH[5], rz, fmaf(GH[4], ry, GH[3] * rx)),
                                  fmaf(GH[8], rz, fmaf(GH[7], ry, GH[6] * rx))) - gcen);
                }
            }
            float* dst = gxf + 3 * bw_atoms[t];
            dst[0] = gp.x; dst[1] = gp.y; dst[2] = gp.z;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the LDS buffer is read before the next frame's items overwrite it
    }
}

// =============================================================================================
// mlp_mfma_kernel: wide MLP over precomputed features, one wave per 16-frame row block
// =============================================================================================
// Packed weights, layer l (fp32 path): Wp[Jp][Kp] row-major (torch.nn.Linear layout, zero padded to
// Jp = ceil16(J), Kp = ceil16(K)), then bias[Jp].  v_mfma_f32_16x16x4_f32 wants, per k-step, lane l
// to hold A[row l&15][k = l>>4] and B[k = l>>4][col l&15]; the k order inside a 16-wide group is
// free as long as A and B agree, so lane quad q = l>>4 takes the 4 CONTIGUOUS k's 4q..4q+3 of the
// group (one 16-byte load each for A from LDS and B from the weight row) and feeds them to 4 MFMAs.
struct MlpArgs {
    long n_frames;
    int n_layers;
    int act;
    int dims[MOLANN_MAX_LAYERS + 1];
    int kp[MOLANN_MAX_LAYERS];    // padded K per layer
    int jp[MOLANN_MAX_LAYERS];    // padded J per layer
    long woff[MOLANN_MAX_LAYERS]; // element offset of layer l in the packed buffer
    int ld[2];                    // LDS row strides (elements): buffer 0 feeds even layers, buffer 1 odd layers
    int lds_per_wave;             // bytes
    int in_stride;                // row stride of the input features (floats)
};

__device__ __forceinline__ unsigned short f2bf(float f) { // round-to-nearest-even, NaN kept
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

// NBW n-blocks (16 output columns each) advance together: one A fragment read from LDS feeds NBW MFMAs on
// NBW independent accumulator chains, their B fragments (weight rows, 16 B per lane) stream from L2.
template <bool BF16, int NBW, typename elem_t>
__device__ __forceinline__ void mlp_nblocks(const elem_t* __restrict__ W, const float* __restrict__ bias, const elem_t* cur,
                                            int ld_cur, elem_t* nxt, int ld_nxt, float* __restrict__ out, long frame0,
                                            int nrow, int n0, int Kp, int J, bool last, int act, int r16, int q) {
    constexpr int KG = BF16 ? 32 : 16; // k's consumed per group (bf16: one MFMA; f32: four MFMAs)
    constexpr int KQ = BF16 ? 8 : 4;   // contiguous k's per lane quad
    f32x4 acc[NBW];
    const elem_t* wrow[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const float bj = bias[n0 + 16 * b + r16]; // C[row 4q+reg][col r16]: bias depends on the column only
        acc[b] = (f32x4){bj, bj, bj, bj};
        wrow[b] = W + (long)(n0 + 16 * b + r16) * Kp + q * KQ;
    }
    const elem_t* arow = cur + r16 * ld_cur + q * KQ;
#pragma unroll 2
    for (int kg = 0; kg < Kp; kg += KG) {
        if constexpr (BF16) {
            const bf16x8 av = *(const bf16x8*)(arow + kg);
            bf16x8 bv[NBW];
#pragma unroll
            for (int b = 0; b < NBW; ++b) bv[b] = *(const bf16x8*)(wrow[b] + kg);
#pragma unroll
            for (int b = 0; b < NBW; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[b], acc[b], 0, 0, 0);
        } else {
            const f32x4 av = *(const f32x4*)(arow + kg);
            f32x4 bv[NBW];
#pragma unroll
            for (int b = 0; b < NBW; ++b) bv[b] = *(const f32x4*)(wrow[b] + kg);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int b = 0; b < NBW; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[b][s], acc[b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int col = n0 + 16 * b + r16;
        if (last) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q + r;
                if (row < nrow && col < J) out[(frame0 + row) * (long)J + col] = acc[b][r];
            }
        } else {
            float h[4] = {acc[b][0], acc[b][1], acc[b][2], acc[b][3]};
            activate<4, false>(act, h);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = col < J ? h[r] : 0.f; // padded columns feed zero into the next layer
                if constexpr (BF16) nxt[(4 * q + r) * ld_nxt + col] = f2bf(v);
                else nxt[(4 * q + r) * ld_nxt + col] = v;
            }
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void mlp_mfma_kernel(const float* __restrict__ feat, float* __restrict__ out,
                                                       const void* __restrict__ wpack_v, MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using elem_t = typename std::conditional<BF16, unsigned short, float>::type;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    elem_t* buf[2];
    buf[0] = (elem_t*)(smem + (size_t)wave * a.lds_per_wave);
    buf[1] = buf[0] + 16 * a.ld[0];
    const elem_t* wpack = (const elem_t*)wpack_v;
    const long n_blocks = (a.n_frames + 15) >> 4;

    for (long rb = (long)blockIdx.x * wpb + wave; rb < n_blocks; rb += (long)gridDim.x * wpb) {
        const long frame0 = rb << 4;
        const long remf = a.n_frames - frame0;
        const int nrow = remf < 16 ? (int)remf : 16;
        // ---- stage the 16 input rows into LDS (zero padded to kp[0]) -------------------------
        {
            const int K0 = a.dims[0], Kp0 = a.kp[0];
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                const float* src = feat + (frame0 + (r < nrow ? r : nrow - 1)) * (long)a.in_stride;
                for (int k = lane; k < Kp0; k += 64) {
                    const float v = k < K0 ? src[k] : 0.f;
                    if constexpr (BF16) buf[0][r * a.ld[0] + k] = f2bf(v);
                    else buf[0][r * a.ld[0] + k] = v;
                }
            }
        }
        for (int l = 0; l < a.n_layers; ++l) {
            const int Kp = a.kp[l], Jp = a.jp[l], J = a.dims[l + 1];
            const elem_t* W = wpack + a.woff[l];
            const float* bias = (const float*)(W + (long)Jp * Kp);
            const bool last = (l + 1 == a.n_layers);
            const elem_t* cur = buf[l & 1];
            elem_t* nxt = buf[(l + 1) & 1];
            const int ld_cur = a.ld[l & 1], ld_nxt = a.ld[(l + 1) & 1];
            int n0 = 0;
            for (; n0 + 64 <= Jp; n0 += 64)
                mlp_nblocks<BF16, 4>(W, bias, cur, ld_cur, nxt, ld_nxt, out, frame0, nrow, n0, Kp, J, last, a.act, r16, q);
            for (; n0 < Jp; n0 += 16)
                mlp_nblocks<BF16, 1>(W, bias, cur, ld_cur, nxt, ld_nxt, out, frame0, nrow, n0, Kp, J, last, a.act, r16, q);
            if (!last) { // zero the k padding of the next layer beyond Jp (kp[l+1] may exceed Jp)
                const int Kn = a.kp[l + 1];
                for (int k = Jp + lane; k < Kn; k += 64)
                    for (int r = 0; r < 16; ++r) nxt[r * ld_nxt + k] = (elem_t)0;
            }
        }
    }
}

// =============================================================================================
// pack kernels (live parameters -> plan-owned padded copies)
// =============================================================================================
struct PackArgs {
    const float* W[MOLANN_MAX_LAYERS];
    const float* b[MOLANN_MAX_LAYERS];
    int dims[MOLANN_MAX_LAYERS + 1];
    int n_layers;
    int fused;      // also write the lane kernel's per-lane MFMA fragments
    // mfma layout
    int kp[MOLANN_MAX_LAYERS];
    int jp[MOLANN_MAX_LAYERS];
    long moff[MOLANN_MAX_LAYERS];
    int bf16;
};

// Per-lane MFMA fragments of the fused MLP (frames_lane_kernel): for layer l, unit block ub, k-step ks,
// lane (i = lane&15, q = lane>>4):  A = W_l[16ub + i][k],  k = 4ks + q for layer 0 and
// k = 16(ks>>2) + 4q + (ks&3) for the following layers; then the bias fragments b_l[16ub + 4q + r].
__global__ void pack_lane_kernel(float* __restrict__ dst, PackArgs p) {
    const int NL = p.n_layers;
    const int base = NL * 1024 + NL * 512;
    const int total = base + 1024; // + layer 0's k-steps 8..15 (feature dims 33..64, specialised kernel only)
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, i = lane & 15, q = lane >> 4;
        float v = 0.f;
        if (e < NL * 1024) {
            const int ks = (e >> 6) & 7, ub = (e >> 9) & 1, l = e >> 10;
            const int K = p.dims[l], J = p.dims[l + 1];
            const int j = 16 * ub + i;
            const int k = l == 0 ? 4 * ks + q : 16 * (ks >> 2) + 4 * q + (ks & 3);
            if (j < J && k < K) v = p.W[l][(long)j * K + k];
        } else if (e < base) {
            const int f = e - NL * 1024;
            const int r = (f >> 6) & 3, ub = (f >> 8) & 1, l = f >> 9;
            const int j = 16 * ub + 4 * q + r;
            if (j < p.dims[l + 1]) v = p.b[l][j];
        } else {
            const int f = e - base;
            const int ks = 8 + ((f >> 6) & 7), ub = (f >> 9) & 1;
            const int K = p.dims[0], J = p.dims[1];
            const int j = 16 * ub + i, k = 4 * ks + q;
            if (j < J && k < K) v = p.W[0][(long)j * K + k];
        }
        dst[e] = v;
    }
}

// mfma layout, layer l: Wp[Jp][Kp] (zero padded; fp32 or bf16) then bias[Jp] (fp32)
__global__ void pack_mfma_kernel(void* __restrict__ dst_v, PackArgs p) {
    const int l = blockIdx.y;
    const int K = p.dims[l], J = p.dims[l + 1], Kp = p.kp[l], Jp = p.jp[l];
    const long nW = (long)Jp * Kp;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nW + Jp; e += (long)gridDim.x * blockDim.x) {
        if (e < nW) {
            const int j = (int)(e / Kp), k = (int)(e - (long)j * Kp);
            const float v = (j < J && k < K) ? p.W[l][(long)j * K + k] : 0.f;
            if (p.bf16) ((unsigned short*)dst_v)[p.moff[l] + e] = f2bf(v);
            else ((float*)dst_v)[p.moff[l] + e] = v;
        } else {
            const int j = (int)(e - nW);
            const float v = j < J ? p.b[l][j] : 0.f;
            if (p.bf16) ((float*)((unsigned short*)dst_v + p.moff[l] + nW))[j] = v;
            else ((float*)dst_v)[p.moff[l] + nW + j] = v;
        }
    }
}


// ---- chain MLP (molann_mlp_jit.inc): host mirror of the kernel text's constexpr geometry --------------------
struct ChainGeom {
    int nl;
    int bf16; // 1: bf16 MFMA 16x16x32 (32 k's per fragment, chunks of two 16-unit blocks); 0: fp32 MFMA 16x16x4 (16, one)
    int dims[MOLANN_MAX_LAYERS + 1];
    int cb() const { return bf16 ? 2 : 1; }
    int kw() const { return bf16 ? 32 : 16; }
    int ub(int l) const { return (dims[l + 1] + 15) / 16; }
    int ubp(int l) const { return (ub(l) + cb() - 1) / cb() * cb(); }
    int ks(int l) const { return l == 0 ? (dims[0] + kw() - 1) / kw() : ubp(l - 1) / cb(); }
    int npair() const { return (nl + 1) / 2; }
    bool has_c(int p) const { return 2 * p + 1 < nl; }
    int nchunk(int p) const { return ubp(2 * p) / cb(); }
    int slab_frags(int p) const { return cb() * ks(2 * p) + (has_c(p) ? ub(2 * p + 1) : 0); }
    int slab_max() const { int m = 0; for (int p = 0; p < npair(); ++p) m = std::max(m, slab_frags(p)); return m; }
    long total_frags() const { long s = 0; for (int p = 0; p < npair(); ++p) s += (long)nchunk(p) * slab_frags(p); return s; }
    int bias_off(int l) const { int s = 0; for (int i = 0; i < l; ++i) s += 16 * ubp(i); return s; }
    // registers (VGPR + AGPR) the kernel keeps live per 16-frame block: the pair's input operands, the C
    // layer's accumulators, the P chunk, and half of the next pair's operands while they are being formed
    int regs_per_fb() const {
        int m = 0;
        for (int p = 0; p < npair(); ++p) {
            const int ksin = ks(2 * p), ubc = has_c(p) ? ub(2 * p + 1) : 0, ksn = has_c(p) ? ubp(2 * p + 1) / cb() : 0;
            m = std::max(m, 4 * (ksin + ubc + cb() + (ksn + 1) / 2));
        }
        return m;
    }
};

struct ChainPackArgs {
    const float* W[MOLANN_MAX_LAYERS];
    const float* b[MOLANN_MAX_LAYERS];
    int dims[MOLANN_MAX_LAYERS + 1];
    int n_layers, npair, bf16;
    int pair_start[MOLANN_MAX_LAYERS / 2 + 1]; // first fragment of pair p (last entry: total)
    int slab_frags[MOLANN_MAX_LAYERS / 2];
    int ks_in[MOLANN_MAX_LAYERS / 2];
    int bias_off[MOLANN_MAX_LAYERS + 1];
    long stream_bytes;
};

// Fragment F of the stream = 64 lanes x 16 B: lane (i = l&15, q = l>>4) holds A[j = 16 ub + i][k slot s = 0..7];
// layer 0: k = 32 ks + 8q + s (the order of the feature row), later layers: k = 16(2ks + (s>>2)) + 4q + (s&3)
// (the order the previous layer's accumulators come in).  Order of fragments: see molann_mlp_jit.inc.
__global__ void pack_chain_kernel(unsigned char* __restrict__ dst, ChainPackArgs a) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;
    const long total = (long)a.pair_start[a.npair] * 64;
    for (long e = tid; e < total; e += nthreads) {
        const int F = (int)(e >> 6), lane = (int)(e & 63), i = lane & 15, q = lane >> 4;
        int p = 0;
        while (p + 1 < a.npair && F >= a.pair_start[p + 1]) ++p;
        const int rel = F - a.pair_start[p], c = rel / a.slab_frags[p], f = rel % a.slab_frags[p];
        const int cb = a.bf16 ? 2 : 1;
        int l, ub, ks;
        if (f < cb * a.ks_in[p]) { l = 2 * p; ks = f / cb; ub = cb * c + f % cb; }
        else { l = 2 * p + 1; ub = f - cb * a.ks_in[p]; ks = c; }
        const int K = a.dims[l], J = a.dims[l + 1], j = 16 * ub + i;
        uint4 w;
        if (a.bf16) {
            unsigned short v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int k = l == 0 ? 32 * ks + 8 * q + s : 16 * (2 * ks + (s >> 2)) + 4 * q + (s & 3);
                v[s] = f2bf((j < J && k < K) ? a.W[l][(long)j * K + k] : 0.f);
            }
            w.x = v[0] | ((unsigned)v[1] << 16); w.y = v[2] | ((unsigned)v[3] << 16);
            w.z = v[4] | ((unsigned)v[5] << 16); w.w = v[6] | ((unsigned)v[7] << 16);
        } else { // fp32: k = 16 ks + 4q + r in every layer (layer 0: the feature row; later: the previous accumulators)
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * ks + 4 * q + r;
                v[r] = (j < J && k < K) ? a.W[l][(long)j * K + k] : 0.f;
            }
            w.x = __float_as_uint(v[0]); w.y = __float_as_uint(v[1]); w.z = __float_as_uint(v[2]); w.w = __float_as_uint(v[3]);
        }
        *(uint4*)(dst + (size_t)F * 1024 + lane * 16) = w;
    }
    float* bias = (float*)(dst + a.stream_bytes);
    for (long e = tid; e < a.bias_off[a.n_layers]; e += nthreads) {
        int l = 0;
        while (l + 1 < a.n_layers && e >= a.bias_off[l + 1]) ++l;
        const int u = (int)e - a.bias_off[l];
        bias[e] = u < a.dims[l + 1] ? a.b[l][u] : 0.f;
    }
}

// ref_x (device, centred) -> plan copy + the constants the kernels need after it
template <typename S> // S = float: the module's buffer as the reference builds it; double: a `.double()` model's buffer
__global__ void pack_ref_kernel(float* __restrict__ dst, double* __restrict__ dst64, const S* __restrict__ ref,
                                int n_align) {
    if (blockIdx.x != 0) return;
    __shared__ double red[4][256];
    double s[4] = {0., 0., 0., 0.};
    for (int i = threadIdx.x; i < n_align; i += blockDim.x) {
        const S rx = ref[3 * i], ry = ref[3 * i + 1], rz = ref[3 * i + 2];
        dst[3 * i] = (float)rx; dst[3 * i + 1] = (float)ry; dst[3 * i + 2] = (float)rz;
        dst64[3 * i] = rx; dst64[3 * i + 1] = ry; dst64[3 * i + 2] = rz;
        s[0] += rx; s[1] += ry; s[2] += rz;
        s[3] += (double)rx * rx + (double)ry * ry + (double)rz * rz;
    }
    for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int w = blockDim.x >> 1; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* c = dst + 3 * n_align;
        double* c64 = dst64 + 3 * n_align;
        for (int k = 0; k < 4; ++k) { c[k] = (float)red[k][0]; c64[k] = red[k][0]; }
        c[4] = 1.0f / (float)n_align;
        c[5] = (float)n_align;
        c64[4] = 1.0 / (double)n_align;
        c64[5] = (double)n_align;
    }
}

inline int ceil_to(int v, int m) { return (v + m - 1) / m * m; }

// Switches that change WHAT is computed, skip part of it, or exist only for experiments (MOLANN_DEBUG_*,
// MOLANN_ELIDE_INVARIANT_ALIGNMENT, MOLANN_JIT_EXTRA_FLAGS) are read by the diagnostics build alone
// (`make diag` -> libmolann_hip_diag.so, -DMOLANN_DIAGNOSTICS; tools/ loads it with MOLANN_DIAG_LIB=1).  The
// product library never looks at them: molann_build_kind() says which one is loaded.
inline const char* diag_env(const char* name) {
#ifdef MOLANN_DIAGNOSTICS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// switches read once per process (never on the launch path)
struct DebugEnv {
    int ablate, lds_pad, wave_bpc, wave_pre;
    DebugEnv() {
        const char* e;
        ablate = (e = diag_env("MOLANN_DEBUG_ABLATE")) ? atoi(e) : 0;
        lds_pad = (e = diag_env("MOLANN_DEBUG_LDS_PAD")) ? atoi(e) : 0;
        wave_bpc = (e = getenv("MOLANN_WAVE_BPC")) ? atoi(e) : 0;
        wave_pre = (e = getenv("MOLANN_WAVE_PRE")) ? atoi(e) : -1; // item rounds preloaded by frames_wave_kernel
    }
};
const DebugEnv& debug_env() {
    static const DebugEnv env;
    return env;
}

} // namespace

// =============================================================================================
// plan
// =============================================================================================
struct molann_plan {
    int device;
    int num_cus;
    int n_inp, n_align, n_features, n_items, d_feat, use_angle_value;
    int n_layers, act, mlp_prec;
    int dims[MOLANN_MAX_LAYERS + 1];
    int out_dim;
    int family;        // 0 lane-per-frame, 1 wave-per-frame
    bool fused_mlp;    // lane kernel runs the MLP itself
    // device memory (one allocation)
    unsigned char* blob;
    int* d_align_idx;
    float* d_ref;      // [3a] + 6 constants
    double* d_ref64;   // the same in fp64 (covariance accumulation)
    ItemDev* d_items;
    float* d_wlane;    // fused layout
    void* d_wmfma;     // mfma layout
    float* d_work;     // two feature chunks [2][work_frames][d_feat] (ping-pong between gather and MLP kernels)
    long work_frames;
    hipStream_t side;  // the MLP kernel of chunk i runs here while the caller's stream gathers chunk i+1
    hipEvent_t ev_feat[2], ev_mlp[2];
    // The workspace, the side stream and these events belong to the plan: forwards of one plan issued from different
    // streams (or threads) are ordered one after the other - the host side by launch_mu, the device side by making a
    // new caller's stream wait for ev_done, recorded behind the previous call's join.
    hipEvent_t ev_done;
    hipStream_t last_stream;
    bool have_done;
    std::mutex* launch_mu;
    std::mutex* jit_mu;        // the lazily built kernels (features twin, backward pair, backward workspace)
    int kp[MOLANN_MAX_LAYERS], jp[MOLANN_MAX_LAYERS];
    long moff[MOLANN_MAX_LAYERS];
    int mlp_ld[2], mlp_lds_per_wave;
    // lane kernel geometry: [0] feature mode (tile + staging columns), [1] align-out mode (tile only)
    struct LaneGeom { int lds_per_wave, fbuf_off, wpb, ok; } geom[2];
    // register-resident mode: <= 16 touched atoms ("slots", align atoms first) and tables in slot indices
    // plan-specialised lane kernel (hipRTC), feature mode; nullptr -> generic kernel
    hipModule_t jit_mod;
    hipFunction_t jit_fn;
    int jit_nl;          // Linear layers fused into it (0: features only)
    LaneGeom jit_geom;   // its own LDS geometry: the compact tile (touched 16-byte windows only) + staging columns
    int jit_waves;       // waves per SIMD it was compiled for
    int jit_ncons, jit_nload, jit_nslot, jit_bpc, jit_lds_block; // loader / consumer block geometry of the specialised forward kernel
    bool jit_only;       // no ahead-of-time kernel serves this plan's fused forward (large frame / 33..64 features)
    char jit_note[96];
    struct JitSpecBox* spec;   // what the specialised kernels are generated from (kept for the lazy backward build)
    hipModule_t bwd_mod;
    hipFunction_t bwd_fn;      // backward of the preprocessing (molann_lane_bwd.inc), compiled at the first backward
    int bwd_state;             // 0 not tried, 1 ready, -1 unavailable
    hipModule_t rbwd_mod;
    hipFunction_t rbwd_fn;     // the whole backward in one pass over x (molann_bwd_ring.inc), compiled at the first backward
    int rbwd_state, rbwd_ncons, rbwd_nload, rbwd_nslot, rbwd_lds;
    hipModule_t vjp_mod;
    hipFunction_t vjp_fn;      // the one-pass backward that also stores the forward's outputs (molann_value_and_vjp_f32), same geometry
    int vjp_state;
    hipModule_t mbwd_mod;
    hipFunction_t mbwd_fn;     // backward of the fused family's MLP (molann_mlp_bwd.inc), compiled at the first backward
    int mbwd_state, mbwd_wpb;
    // features-only twin of a forward kernel that has the MLP fused in (what molann_features_f32 and the backward's
    // recompute launch on such a plan), compiled at the first use
    struct JitSpecBox* align_spec;  // AlignmentLayer.forward as alignment + one position item per atom through the specialised kernel
    hipModule_t align_mod;
    hipFunction_t align_fn;    // ... compiled at the first molann_align_f32
    int align_state;
    hipModule_t train_mod;
    hipFunction_t train_fn;    // the fused forward kernel that also writes the features (molann_forward_train_f32), same geometry
    int train_state;
    hipModule_t feat_mod;
    hipFunction_t feat_fn;
    int feat_state, feat_ncons, feat_nload, feat_nslot, feat_bpc, feat_lds_block;
    float* d_gpart;            // molann_mlp_bwd's parameter sums, one row per block: [num_cus][n_grad_params]
    hipEvent_t ev_bwork;       // the backward workspaces (d_gpart, d_bwork) are shared by all streams: see launch_mu
    hipStream_t bwork_stream;
    bool have_bwork;
    float* d_bwork;            // molann_backward_f32 on a plan with an MLP: features and their gradient, [2][bwork_frames][d_feat]
    long bwork_frames;
    int n_grad_params;         // floats of the parameter-gradient buffer (dW_l[J][K], db_l[J] per layer)
    // plan-specialised wide bf16 MLP (molann_mlp_jit.inc); nullptr -> mlp_mfma_kernel<bf16>
    hipModule_t chain_mod;
    hipFunction_t chain_fn;
    unsigned char* d_wchain;   // weight fragments in consumption order, then the padded fp32 biases
    long chain_stream_bytes;
    int chain_fb;              // 16-frame blocks per wave
    int chain_waves;           // waves per block: 4 (weights streamed through LDS slabs) or 8 (weight stream resident in LDS)
    char chain_note[96];
    char mlp_info[96];         // name + geometry of the last MLP kernel launch
    // large frames through frames_ring_kernel: per-frame window list, LDS positions of the alignment atoms and of the
    // items' atoms inside the staged image; ring_nd = LDS-DMA instructions per frame (0: frames_wave_kernel serves the plan)
    int align_first;           // align_idx[0]
    int* d_ring_win;
    int* d_ring_align_pos;
    ItemDev* d_ring_items;
    int ring_nd, ring_nwin;
    // large frames, backward without atomics (frames_wave_bwd_gather_kernel): touched atoms, CSR of their (item, atom) pairs
    int* d_bw_atoms; int* d_bw_ptr; int* d_bw_list; int* d_bw_align;
    int bw_touched;
    int n_slots;
    bool regs_mode;
    int* d_slots;
    ItemDev* d_items_slot;
    bool mlp_packed;
    char last_info[256];
};

namespace {

#define HIP_TRY(expr)                      \
    do {                                   \
        hipError_t _e = (expr);            \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

// waves never share LDS, so the block size is free: take the one that packs most waves into the CU's
// 160 KiB (blocks <= 64 KiB: the LDS-DMA destination offset is 16 bits), larger block on ties
void lane_geometry(molann_plan::LaneGeom& g, int tile_bytes, int cols_needed) {
    const int tile_pad = ceil_to(tile_bytes, 16);
    const int fbuf_bytes = ceil_to(cols_needed * FB_STRIDE * 4, 16);
    g.fbuf_off = tile_pad;
    g.lds_per_wave = tile_pad + fbuf_bytes;
    const long L = g.lds_per_wave;
    int best = 0, best_waves = 0;
    for (int wpb = 4; wpb >= 1; --wpb) {
        if (wpb * L > 65536) continue;
        long waves = wpb * (163840 / (wpb * L));
        if (waves > 32) waves = 32;
        if (waves > best_waves) { best_waves = (int)waves; best = wpb; }
    }
    g.wpb = best;
    g.ok = (best >= 1 && best_waves >= 4) ? 1 : 0; // fewer than 4 waves per CU: use the wave-per-frame kernel
}

int validate_desc(const molann_plan_desc* d) {
    if (!d) return MOLANN_E_NULL;
    if (d->abi_version != MOLANN_ABI_VERSION) return MOLANN_E_DESC;
    if (d->n_inp <= 0 || d->n_align < 0 || d->n_features < 0 || d->n_layers < 0) return MOLANN_E_DESC;
    if (d->n_layers > MOLANN_MAX_LAYERS) return MOLANN_E_UNSUPPORTED;
    if (d->n_align == 0 && d->n_features == 0 && d->n_layers == 0) return MOLANN_E_DESC;
    if (d->n_align > 0) {
        if (!d->align_idx || !d->ref_x) return MOLANN_E_NULL;
        for (int i = 0; i < d->n_align; ++i)
            if (d->align_idx[i] < 0 || d->align_idx[i] >= d->n_inp) return MOLANN_E_INDEX;
    }
    if (d->n_features > 0) {
        if (!d->feat_type || !d->feat_ptr || !d->feat_idx) return MOLANN_E_NULL;
        if (d->feat_ptr[0] != 0) return MOLANN_E_DESC;
        for (int f = 0; f < d->n_features; ++f) {
            const int cnt = d->feat_ptr[f + 1] - d->feat_ptr[f];
            const int t = d->feat_type[f];
            if (cnt < 0) return MOLANN_E_DESC;
            if (t == MOLANN_FEAT_ANGLE) { if (cnt != 3) return MOLANN_E_FEATURE; }   // feature.py:88
            else if (t == MOLANN_FEAT_BOND) { if (cnt != 2) return MOLANN_E_FEATURE; }   // feature.py:91
            else if (t == MOLANN_FEAT_DIHEDRAL) { if (cnt != 4) return MOLANN_E_FEATURE; } // feature.py:94
            else if (t == MOLANN_FEAT_POSITION) { if (cnt < 1) return MOLANN_E_FEATURE; }
            else return MOLANN_E_FEATURE;                                                  // feature.py:82
            for (int i = d->feat_ptr[f]; i < d->feat_ptr[f + 1]; ++i)
                if (d->feat_idx[i] < 0 || d->feat_idx[i] >= d->n_inp) return MOLANN_E_INDEX;
        }
    }
    if (d->n_layers > 0) {
        if (!d->layer_dims) return MOLANN_E_NULL;
        for (int i = 0; i <= d->n_layers; ++i)
            if (d->layer_dims[i] <= 0) return MOLANN_E_DESC;
        if (d->activation < 0 || d->activation > MOLANN_ACT_GELU) return MOLANN_E_UNSUPPORTED;
        if (d->mlp_precision != MOLANN_MLP_F32 && d->mlp_precision != MOLANN_MLP_BF16) return MOLANN_E_UNSUPPORTED;
    }
    return MOLANN_OK;
}

int grid_for(const molann_plan* p, long work_items, int items_per_block, int blocks_per_cu) {
    long need = (work_items + items_per_block - 1) / items_per_block;
    long cap = (long)p->num_cus * blocks_per_cu;
    long g = need < cap ? need : cap;
    return (int)(g < 1 ? 1 : g);
}

void fill_pre_args(const molann_plan* p, PreArgs& a, long n_frames, int mode, int out_cols, bool with_mlp,
                   const void* x, const void* out) {
    memset(&a, 0, sizeof(a));
    a.n_frames = n_frames;
    a.n_inp = p->n_inp;
    a.frame_dw = 3 * p->n_inp;
    a.mode = mode;
    a.n_align = p->n_align;
    a.n_items = p->n_items;
    a.out_cols = out_cols > 0 ? out_cols : 1;
    a.step_f = 64 / a.out_cols;
    a.step_c = 64 % a.out_cols;
    const molann_plan::LaneGeom& g = p->geom[mode == 1 ? 1 : 0];
    a.lds_per_wave = g.lds_per_wave;
    a.fbuf_off = g.fbuf_off;
    a.n_slots = p->n_slots;
    a.x_wide = (((uintptr_t)x) & 15) == 0;
    a.out_wide = (((uintptr_t)out) & 15) == 0;
    a.n_layers = with_mlp ? p->n_layers : 0;
    a.act = p->act;
    for (int i = 0; i <= p->n_layers; ++i) a.dims[i] = p->dims[i];
    a.out_vec4 = (a.out_wide && (a.out_cols & 3) == 0) ? 1 : 0;
    a.ablate = debug_env().ablate;
}


// ---------------------------------------------------------------------------------------------
// plan-time specialisation of the lane kernel (molann_lane_jit.inc) through hipRTC
// ---------------------------------------------------------------------------------------------
struct RtcApi {
    decltype(&hiprtcCreateProgram) create;
    decltype(&hiprtcCompileProgram) compile;
    decltype(&hiprtcGetCodeSize) code_size;
    decltype(&hiprtcGetCode) code;
    decltype(&hiprtcGetProgramLogSize) log_size;
    decltype(&hiprtcGetProgramLog) log;
    decltype(&hiprtcDestroyProgram) destroy;
    decltype(&hiprtcVersion) version;   // optional: part of the code-object cache's key
    bool ok;
};

const RtcApi* rtc_api() {
    static RtcApi api = [] {
        RtcApi a;
        memset(&a, 0, sizeof(a));
        void* h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libhiprtc.so.7", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.create = (decltype(a.create))dlsym(h, "hiprtcCreateProgram");
        a.compile = (decltype(a.compile))dlsym(h, "hiprtcCompileProgram");
        a.code_size = (decltype(a.code_size))dlsym(h, "hiprtcGetCodeSize");
        a.code = (decltype(a.code))dlsym(h, "hiprtcGetCode");
        a.log_size = (decltype(a.log_size))dlsym(h, "hiprtcGetProgramLogSize");
        a.log = (decltype(a.log))dlsym(h, "hiprtcGetProgramLog");
        a.destroy = (decltype(a.destroy))dlsym(h, "hiprtcDestroyProgram");
        a.version = (decltype(a.version))dlsym(h, "hiprtcVersion");
        a.ok = a.create && a.compile && a.code_size && a.code && a.log_size && a.log && a.destroy;
        return a;
    }();
    return &api;
}

std::string join_chunks(const char* const* chunks) {
    std::string s;
    for (int i = 0; chunks[i]; ++i) s += chunks[i];
    return s;
}

struct JitSpec { // what the specialised kernel is compiled for
    int n_inp, n_align, n_layers, act, d_feat, out_cols, wpb, lds_per_wave, fbuf_off;
    int waves_per_eu = 2;             // occupancy the backward kernel is compiled for (amdgpu_waves_per_eu)
    int nbuf = 1;
    bool save_feat = false;           // forward kernel: also writes the features (training; molann_forward_train_f32)
    bool with_values = false;         // one-pass backward kernel: also writes the forward's outputs (molann_value_and_vjp_f32)
    bool frag_lds = false;            // one-pass backward: weight fragments in an LDS image at img_off instead of registers
    int img_off = 0;
    // forward kernel: loader / consumer block around a ring of tile slots (molann_lane_jit.inc)
    int nload = 1;                    // loaders per block
    int ncons = 0, nslot = 0, depth = 0, ring_off = 0, tile_stride = 0, fb_off = 0, fb_bytes = 0, lds_block = 0, bpc = 0;
    std::vector<int> win;             // compact staging: first dword of each 16-byte window copied per frame
    std::vector<int> slots;           // slot -> atom
    std::vector<ItemDev> items;       // atoms as slot indices
    std::vector<int> dims;
};

constexpr int JIT_MAX_ITEMS = 128, JIT_MAX_SLOTS = 32;
} // namespace
struct JitSpecBox { JitSpec j; std::vector<int> kp, jp; std::vector<long> woff; };
namespace {

// Compact staging (molann_lane_jit.inc): per frame only the 16-byte windows that hold a touched atom are copied
// to LDS.  Greedy cover of the touched dwords (LDS-DMA gathers from any dword-aligned address); an odd number of
// windows keeps the per-lane ds_read_b128 of the register fill conflict-free (frame stride = NW x 16 B).
std::vector<int> compact_windows(const std::vector<int>& slot_atoms, int n_inp) {
    const int frame_dw = 3 * n_inp;
    std::vector<char> used(frame_dw, 0);
    for (int a : slot_atoms)
        for (int c = 0; c < 3; ++c) used[3 * a + c] = 1;
    std::vector<int> win;
    for (int d = 0; d < frame_dw; ++d) {
        if (!used[d]) continue;
        if (!win.empty() && d < win.back() + 4) continue; // covered by the last window
        win.push_back(std::min(d, frame_dw - 4));      // never past the end of the frame (the last frame of x)
    }
    if (win.empty()) win.push_back(0);
    if (win.size() % 2 == 0) win.push_back(win.back());
    return win;
}

// LDS geometry of the specialised forward kernel: per block a header of hand-off words, a ring of NSLOT compact
// tiles and one feature staging buffer per consumer wave (the fused MLP reads only the real D_FEAT rows; padded k's
// are zeros in registers).  A block is NCONS consumers + 1 loader; blocks <= 64 KiB (the LDS-DMA destination offset
// is 16 bits).  Preference: 8 consumers per CU (two per SIMD: the arithmetic saturates the vector ALU there), then
// the deepest ring, then the fewest blocks (= loader waves).
void jit_geometry(JitSpec& j, molann_plan::LaneGeom& g, int staging_rows, int fallback_cols, int max_cons = 14) {
    const int tile = ceil_to(64 * 16 * (int)j.win.size(), 16);
    const int fb = ceil_to(std::max(1, staging_rows) * FB_STRIDE * 4, 16);
    // One block per CU: 14 consumer waves + 2 loaders = four waves per SIMD (the kernel is built for <= 128 VGPRs).
    // Two loaders because a wave has at most 63 vector-memory operations in flight (vmcnt) = 63 KB of tiles, and
    // the loaded HBM latency (~4 us, tools/stamps.py) needs ~120 KB in flight per CU for 6 TB/s; the consumers beyond
    // two per SIMD add no vector-ALU rate but keep the SIMD issuing while others wait (LDS, MFMA results, a tile).
    j.bpc = 1;
    j.ncons = max_cons;
    j.nload = 2;
    const int header = 256;
    long nslot = 0;
    auto fit = [&]() {
        nslot = std::min<long>(16, (163840 / j.bpc - header - (long)j.ncons * fb) / tile);
        return nslot >= 2 * j.nload || (j.ncons == 1 && nslot >= 1);
    };
    while (!fit() && j.ncons > 1) { --j.ncons; if (j.ncons < 4) j.nload = 1; }   // large tiles: fewer consumers, down to one
    bool ok = fit();
    if (const char* e = diag_env("MOLANN_DEBUG_LC")) { // experiments: "blocks per CU,consumers,slots,loaders"
        int b = 0, c = 0, n = 0, ld = 1;
        if (sscanf(e, "%d,%d,%d,%d", &b, &c, &n, &ld) >= 3 && b >= 1 && c >= 1 && c <= 15 && n >= 1 && n <= 16 && ld >= 1 && ld <= 4 &&
            header + (long)n * tile + (long)c * fb <= 163840 / b) {
            j.bpc = b; j.ncons = c; nslot = n; j.nload = ld; ok = true;
        }
    }
    g.ok = ok ? 1 : 0;
    if (!g.ok) { lane_geometry(g, tile, fallback_cols); g.ok = 0; return; }
    j.nslot = (int)nslot;
    const int nw = (int)j.win.size();
    j.depth = std::min(6, 63 / std::max(1, nw));   // tiles a loader keeps in flight behind its newest published one: vmcnt <= 63
    j.depth = std::min(j.depth, std::max(0, j.nslot / j.nload - 1));
    if (const char* e = diag_env("MOLANN_DEBUG_LC_DEPTH")) j.depth = std::max(0, std::min(j.depth, atoi(e)));
    j.ring_off = header;
    j.tile_stride = tile;
    j.fb_off = header + j.nslot * tile;
    j.fb_bytes = fb;
    j.lds_block = j.fb_off + j.ncons * fb;
    j.nbuf = 1;
    // (the fields of the older one-buffer-per-wave geometry stay filled: the backward kernel's preamble names them)
    g.wpb = std::min(4, j.ncons); g.lds_per_wave = tile + std::max(fb, 16); g.fbuf_off = tile;
    j.wpb = g.wpb; j.lds_per_wave = g.lds_per_wave; j.fbuf_off = g.fbuf_off;
}

std::string jit_preamble(const JitSpec& j);

std::string jit_source(const JitSpec& j) {
    std::string s = jit_preamble(j);
    s += "#line 1 \"molann_ring.inc\"\n";
    s += join_chunks(k_src_molann_ring_inc);
    s += "#line 1 \"molann_lane_jit.inc\"\n";
    s += join_chunks(k_src_molann_lane_jit_inc);
    return s;
}

// wide bf16 MLP: layer widths, activation and frames per wave as constants
// the whole weight stream resident in LDS (molann_mlp_jit.inc: RESIDENT): no streaming, eight waves per block
inline bool chain_resident(const ChainGeom& g) { return g.total_frags() * 1024 <= 150 * 1024; }
inline int chain_waves(const ChainGeom& g) { return chain_resident(g) ? 8 : 4; }

std::string jit_source_chain(const ChainGeom& g, int act, int fb) {
    std::string s = "// preamble generated from the plan\n";
    char t[160];
    snprintf(t, sizeof(t), "constexpr int NL = %d;\n", g.nl); s += t;
    s += "constexpr int DIMS[] = {";
    for (int i = 0; i <= g.nl; ++i) { snprintf(t, sizeof(t), "%s%d", i ? ", " : "", g.dims[i]); s += t; }
    s += "};\n";
    snprintf(t, sizeof(t), "constexpr int ACT = %d;\nconstexpr int FB = %d;\nconstexpr bool BF16 = %s;\n", act, fb, g.bf16 ? "true" : "false"); s += t;
    snprintf(t, sizeof(t), "constexpr bool RESIDENT = %s;\nconstexpr int WAVES = %d;\n", chain_resident(g) ? "true" : "false", chain_waves(g)); s += t;
    s += "#line 1 \"molann_mlp_jit.inc\"\n";
    s += join_chunks(k_src_molann_mlp_jit_inc);
    return s;
}

// backward of the preprocessing: the forward preamble of the plan without its MLP
std::string jit_source_bwd(const JitSpecBox& b, int lds_per_wave) {
    JitSpec j = b.j;
    j.lds_per_wave = lds_per_wave;
    j.n_layers = 0;
    j.out_cols = j.d_feat;
    j.dims.clear();
    std::string s = jit_preamble(j);
    s += "#line 1 \"molann_lane_bwd.inc\"\n";
    s += join_chunks(k_src_molann_lane_bwd_inc);
    return s;
}

// rows of the [unit][frame] scratch of molann_mlp_bwd.inc (the kernel text computes the same number: n_rows())
int mlp_bwd_rows(const std::vector<int>& dims, int act) {
    const int nl = (int)dims.size() - 1;
    auto pad4 = [](int v) { return (v + 3) & ~3; };
    auto b16 = [](int v) { return (v + 15) >> 4; };
    auto act_row = [&](int l) { int r = 0; for (int i = 0; i < l; ++i) r += pad4(dims[i]); return r; };
    auto z_row = [&](int l) { int r = act_row(nl); for (int i = 0; i < l; ++i) r += pad4(dims[i + 1]); return r; };
    const int z_end = act == 5 ? z_row(nl - 1) : act_row(nl);
    auto d_row = [&](int l) { return l == nl - 1 ? z_end : act_row(l + 1); };
    int m = z_end + pad4(dims[nl]);
    for (int l = 0; l < nl; ++l) m = std::max(m, std::max(act_row(l) + 16 * b16(dims[l]), d_row(l) + 16 * b16(dims[l + 1])));
    return m;
}

// backward of the MLP: layer widths, activation, where the weights live (fp32 MFMA copy: Wp[Jp][Kp], bias[Jp]) and
// the layout of the parameter-gradient buffer (torch layout: dW[J][K] then db[J], layer after layer)
std::string jit_source_mlp_bwd(const JitSpecBox& b, int wpb) {
    const JitSpec& j = b.j;
    std::string s = "// preamble generated from the plan\n";
    char t[128];
    auto K = [&](const char* name, int v) { snprintf(t, sizeof(t), "constexpr int %s = %d;\n", name, v); s += t; };
    auto arr = [&](const char* name, const std::vector<long>& v) {
        s += std::string("constexpr int ") + name + "[] = {";
        for (size_t i = 0; i < v.size(); ++i) { snprintf(t, sizeof(t), "%s%ld", i ? ", " : "", v[i]); s += t; }
        s += "};\n";
    };
    K("NL", j.n_layers); K("ACT", j.act); K("WPB_M", wpb);
    std::vector<long> dims(j.dims.begin(), j.dims.end()), kp(b.kp.begin(), b.kp.end()), jp(b.jp.begin(), b.jp.end()), woff = b.woff, goff;
    long g = 0;
    for (int l = 0; l < j.n_layers; ++l) { goff.push_back(g); g += (long)j.dims[l + 1] * j.dims[l] + j.dims[l + 1]; }
    arr("DIMS", dims); arr("KP", kp); arr("JP", jp); arr("WOFF", woff); arr("GOFF", goff);
    K("N_PARAMS", (int)g);
    s += "constexpr bool FRAG_LDS = false;\n";   // this kernel's waves have the registers for their weight fragments
    s += "constexpr bool WITH_VALUES = false;\n";
    s += "#line 1 \"molann_mlp_tile.inc\"\n";
    s += join_chunks(k_src_molann_mlp_tile_inc);
    s += "#line 1 \"molann_mlp_bwd.inc\"\n";
    s += join_chunks(k_src_molann_mlp_bwd_inc);
    return s;
}

// One-pass backward (molann_bwd_ring.inc): loaders + consumers around the ring of compact tiles; every consumer owns a
// buffer that is MLP scratch, gradient tile and parameter sums in turn.  Two waves per SIMD (256 VGPRs each).
// fragments of the weights the one-pass backward keeps per lane (molann_bwd_ring.inc: frag_ids())
int bwd_ring_frags(const JitSpec& j) {
    int n = 0;
    for (int l = 0; l < j.n_layers; ++l)
        for (int b = 0; b < 2; ++b) {
            for (int s = 0; s < 8; ++s) {
                if (l + 1 < j.n_layers && 16 * b < j.dims[l + 1] && 4 * s < j.dims[l]) ++n;
                if (16 * b < j.dims[l] && 4 * s < j.dims[l + 1]) ++n;
            }
            if (l + 1 < j.n_layers && 16 * b < j.dims[l + 1]) n += 4;
        }
    return n;
}

bool bwd_ring_geometry(JitSpec& j, int n_params) {
    const int tile = ceil_to(64 * 16 * (int)j.win.size(), 16);
    const int rows = j.n_layers > 0 ? mlp_bwd_rows(j.dims, j.act) : 0;
    const int cbuf = ceil_to(std::max(std::max(64 * 12 * j.n_inp, rows * 68 * 4), std::max(16, n_params * 4)), 16);
    const int header = 256;
    j.bpc = 1;
    // weight fragments: in registers up to the C3 class (a few dozen), beyond that in an LDS image shared by the block
    const int frags = bwd_ring_frags(j);
    j.frag_lds = frags > 40;
    const int img = j.frag_lds ? frags * 256 : 0;
    // With an MLP the consumers are bound by their SIMDs' issue cycles (matrix and vector instructions of a SIMD do not overlap:
    // C3 takes 114 us per 1 M frames without the gradient tile whether 4 or 7 consumers work on it), the stream needs one
    // loader, and the seventh consumer shortens the per-wave tail of the gradient tile (152 -> 145 us).
    int c0 = j.n_layers > 0 ? 7 : 6, ld0 = j.n_layers > 0 ? 1 : 2;
    if (const char* e = diag_env("MOLANN_DEBUG_BWD_LC")) { // experiments: "consumers,loaders"
        int c = 0, ld = 0;
        if (sscanf(e, "%d,%d", &c, &ld) == 2 && c >= 1 && c <= 14 && ld >= 1 && ld <= 4) { c0 = c; ld0 = ld; }
    }
    for (int ncons = c0; ncons >= 2; --ncons) {
        const int nload = ncons >= 4 ? ld0 : 1;
        const long nslot = std::min<long>(16, (163840 - header - img - (long)ncons * cbuf) / tile);
        if (nslot < 2 * nload) continue;
        j.ncons = ncons; j.nload = nload; j.nslot = (int)nslot;
        j.depth = std::min(6, 63 / std::max(1, (int)j.win.size()));
        j.depth = std::min(j.depth, std::max(0, j.nslot / j.nload - 1));
        j.ring_off = header;
        j.tile_stride = tile;
        j.fb_off = header + j.nslot * tile;
        j.fb_bytes = cbuf;
        j.img_off = j.fb_off + j.ncons * cbuf;
        j.lds_block = j.img_off + img;
        return true;
    }
    return false;
}

std::string jit_source_bwd_ring(const JitSpecBox& b) {
    const JitSpec& j = b.j;
    std::string s = jit_preamble(j);
    char t[128];
    auto arr = [&](const char* name, const std::vector<long>& v) {
        s += std::string("constexpr int ") + name + "[] = {";
        for (size_t i = 0; i < v.size(); ++i) { snprintf(t, sizeof(t), "%s%ld", i ? ", " : "", v[i]); s += t; }
        s += "};\n";
    };
    std::vector<long> kp(b.kp.begin(), b.kp.end()), jp(b.jp.begin(), b.jp.end()), woff = b.woff, goff;
    long g = 0;
    for (int l = 0; l < j.n_layers; ++l) { goff.push_back(g); g += (long)j.dims[l + 1] * j.dims[l] + j.dims[l + 1]; }
    if (kp.empty()) { kp.push_back(1); jp.push_back(1); woff.push_back(0); goff.push_back(0); }
    arr("KP", kp); arr("JP", jp); arr("WOFF", woff); arr("GOFF", goff);
    snprintf(t, sizeof(t), "constexpr int N_PARAMS = %ld;\n", g);
    s += t;
    s += "#line 1 \"molann_ring.inc\"\n";
    s += join_chunks(k_src_molann_ring_inc);
    s += "#line 1 \"molann_mlp_tile.inc\"\n";
    s += join_chunks(k_src_molann_mlp_tile_inc);
    s += "#line 1 \"molann_bwd_ring.inc\"\n";
    s += join_chunks(k_src_molann_bwd_ring_inc);
    return s;
}

std::string jit_preamble(const JitSpec& j) {
    std::string s = "// preamble generated from the plan\n";
    char b[256];
    auto K = [&](const char* name, int v) { snprintf(b, sizeof(b), "constexpr int %s = %d;\n", name, v); s += b; };
    K("N_INP", j.n_inp); K("N_ALIGN", j.n_align); K("N_SLOTS", (int)j.slots.size()); K("N_ITEMS", (int)j.items.size());
    K("NL", j.n_layers); K("ACT", j.act); K("D_FEAT", j.d_feat); K("OUT_COLS", j.out_cols); K("WPB", j.wpb);
    K("LDS_PER_WAVE", j.lds_per_wave); K("FBUF_OFF", j.fbuf_off);
    K("WAVES_PER_EU", j.waves_per_eu); K("NBUF", j.nbuf);
    K("NCONS", j.ncons); K("NLOAD", j.nload); K("LDS_BLOCK", j.lds_block); K("NSLOT", j.nslot); K("DEPTH", j.depth); K("RING_OFF", j.ring_off); K("TILE_STRIDE", j.tile_stride);
    K("FB_OFF", j.fb_off); K("FB_BYTES", j.fb_bytes);
    s += j.save_feat ? "constexpr bool SAVE_FEAT = true;\n" : "constexpr bool SAVE_FEAT = false;\n";
    s += j.with_values ? "constexpr bool WITH_VALUES = true;\n" : "constexpr bool WITH_VALUES = false;\n";
    s += j.frag_lds ? "constexpr bool FRAG_LDS = true;\n" : "constexpr bool FRAG_LDS = false;\n";
    K("IMG_OFF", j.img_off);
    {   // waves per SIMD the forward kernel must fit (its register budget): every wave of the (NCONS + 1)-wave blocks
        // a CU is to hold - the loader waves carry the consumers' allocation
        const int waves = (j.ncons + j.nload) * std::max(1, j.bpc);
        K("WAVES_FWD", std::max(1, std::min(8, (waves + 3) / 4)));
    }
    {   // Opt-in, never used for a reported number: a plan whose items are all invariant under rigid motion produces the
        // same output with or without its alignment; MOLANN_ELIDE_INVARIANT_ALIGNMENT=1 (read at plan creation) drops
        // the (then dead) Kabsch from the specialised kernel.  Default: the alignment is computed, as the reference does.
        const char* e = diag_env("MOLANN_ELIDE_INVARIANT_ALIGNMENT");
        s += (e && e[0] == '1') ? "constexpr bool ELIDE_ALIGN = true;\n" : "constexpr bool ELIDE_ALIGN = false;\n";
    }
    {   // cache policy of the x stream's LDS-DMA: 2 = nt, the default (every frame is read once; round 3: C2 39.5 -> 36.5 us,
        // C3 66.4 -> 64.7 us per 1 M frames, tools/r03_ab.sh); experiments: 0 = default policy, 19 = sc0 sc1 nt
        const char* e = diag_env("MOLANN_DEBUG_DMA_AUX");
        K("DMA_AUX", e ? atoi(e) : 2);
    }
    {
        std::vector<int> win = j.win.empty() ? std::vector<int>(1, 0) : j.win;
        K("NWIN", (int)win.size());
        s += "constexpr int WIN_START[] = {";
        for (size_t i = 0; i < win.size(); ++i) { snprintf(b, sizeof(b), "%s%d", i ? ", " : "", win[i]); s += b; }
        s += "};\n";
    }
    s += (debug_env().ablate & 32) ? "constexpr bool STAMPS = true;\n" : "constexpr bool STAMPS = false;\n";
    // diagnostic (MOLANN_DEBUG_ABLATE bit 64): every wave stages its first tile only and recomputes it for all its
    // tiles - the kernel's compute time without the HBM stream
    s += (debug_env().ablate & 64) ? "constexpr bool NO_RESTAGE = true;\n" : "constexpr bool NO_RESTAGE = false;\n";
    // diagnostic (bit 128): the whole next tile's LDS-DMA right after the register fill instead of in slices
    s += (debug_env().ablate & 128) ? "constexpr bool DMA_EARLY = true;\n" : "constexpr bool DMA_EARLY = false;\n";
    // diagnostic (bit 256): every DMA re-reads the wave's first tile (L2-resident): the DMA mechanics without HBM
    s += (debug_env().ablate & 256) ? "constexpr bool SAME_TILE = true;\n" : "constexpr bool SAME_TILE = false;\n";
    // diagnostic (bit 512): the fused MLP's 16-byte output stores are (data-dependently) never executed
    s += (debug_env().ablate & 512) ? "constexpr bool NO_STORES = true;\n" : "constexpr bool NO_STORES = false;\n";
    // diagnostic (bit 2048): consumers hand every tile back as soon as it is in registers and compute nothing
    s += (debug_env().ablate & 2048) ? "constexpr bool NO_COMPUTE = true;\n" : "constexpr bool NO_COMPUTE = false;\n";
    { const char* e = diag_env("MOLANN_DEBUG_TILE_GROUP"); K("TILE_GROUP", e && atoi(e) > 0 ? atoi(e) : 1); } // adjacent tiles per block and turn (experiment)
    { const char* e = diag_env("MOLANN_DEBUG_ST_POLICY"); K("ST_POLICY", e ? atoi(e) : 1); } // cache policy of the output stores (1 = nt)
    { const char* e = diag_env("MOLANN_DEBUG_SLEEP"); K("SLEEP_N", e ? atoi(e) : 0); }
    { const char* e = diag_env("MOLANN_DEBUG_SPIN"); K("SPIN_N", e ? atoi(e) : 0); }   // with NO_COMPUTE: 64 x N v_fma per tile // with NO_COMPUTE: idle ~8k cycles x N per tile
    auto A = [&](const char* name, const std::vector<int>& v) {
        s += std::string("constexpr int ") + name + "[] = {";
        for (size_t i = 0; i < v.size(); ++i) { snprintf(b, sizeof(b), "%s%d", i ? ", " : "", v[i]); s += b; }
        s += "};\n";
    };
    std::vector<int> used(j.slots.size(), 0), type, col;
    A("SLOT_ATOM", j.slots);
    s += "constexpr int ITEM_IDX[][4] = {";
    for (size_t i = 0; i < j.items.size(); ++i) {
        const ItemDev& it = j.items[i];
        const int na = item_atoms(it.type);
        for (int k = 0; k < na; ++k) used[it.idx[k]] = 1;
        type.push_back(it.type);
        col.push_back(it.col);
        snprintf(b, sizeof(b), "%s{%d, %d, %d, %d}", i ? ", " : "", it.idx[0], it.idx[1], it.idx[2], it.idx[3]);
        s += b;
    }
    s += "};\n";
    A("ITEM_TYPE", type); A("ITEM_COL", col); A("SLOT_USED", used);
    std::vector<int> dims = j.dims;
    if (dims.empty()) dims.push_back(j.d_feat);
    while (dims.size() < 2) dims.push_back(1); // the kernel text names DIMS[1] even when NL == 0 discards its use
    A("DIMS", dims);
    return s;
}

// compile to a gfx950 code object; returns 0 or a hiprtcResult, log filled on failure
int jit_compile(const std::string& src, std::vector<char>& code, std::string& log, const char* more_flags = nullptr) {
    const RtcApi* rtc = rtc_api();
    if (!rtc->ok) { log = "libhiprtc.so not found"; return -1; }
    const std::string math = join_chunks(k_src_molann_math_h);
    const char* hdr_src[] = {math.c_str()};
    const char* hdr_name[] = {"molann_math.h"};
    hiprtcProgram prog;
    hiprtcResult r = rtc->create(&prog, src.c_str(), "molann_lane_jit.hip", 1, hdr_src, hdr_name);
    if (r != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram failed"; return (int)r; }
    std::vector<std::string> flags = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    if (more_flags && more_flags[0]) flags.push_back(more_flags);
    if (const char* extra = diag_env("MOLANN_JIT_EXTRA_FLAGS")) { // experiments: space-separated compiler flags
        std::string e(extra);
        size_t pos = 0;
        while (pos < e.size()) {
            const size_t sp = e.find(' ', pos);
            const std::string tok = e.substr(pos, sp == std::string::npos ? std::string::npos : sp - pos);
            if (!tok.empty()) flags.push_back(tok);
            if (sp == std::string::npos) break;
            pos = sp + 1;
        }
    }
    std::vector<const char*> opts;
    for (auto& f : flags) opts.push_back(f.c_str());
    // MOLANN_JIT_CACHE_DIR=<dir>: code objects kept on disk under a hash of everything they are made from (kernel text,
    // molann_math.h, flags, the hipRTC version), so a process that builds a plan another one has built skips the compile
    // (1.5-4 s per kernel).  Opt-in; a file that is missing, unreadable or of the wrong size is simply rebuilt.
    std::string cache_file;
    if (const char* dir = getenv("MOLANN_JIT_CACHE_DIR")) {
        if (dir[0]) {
            unsigned long long h = 1469598103934665603ull;   // FNV-1a, 64 bit
            auto mix = [&](const std::string& t) { for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; } h ^= 0xff; h *= 1099511628211ull; };
            mix(src); mix(math);
            for (auto& f : flags) mix(f);
            int maj = 0, min = 0;
            if (rtc->version) (void)rtc->version(&maj, &min);
            char name[96];
            snprintf(name, sizeof(name), "/molann_%016llx_%zu_rtc%d.%d.hsaco", h, src.size(), maj, min);
            cache_file = std::string(dir) + name;
            if (FILE* f = fopen(cache_file.c_str(), "rb")) {
                std::vector<char> blob;
                char buf[1 << 16];
                size_t got;
                while ((got = fread(buf, 1, sizeof(buf), f)) > 0) blob.insert(blob.end(), buf, buf + got);
                fclose(f);
                if (blob.size() > 64 && memcmp(blob.data(), "\177ELF", 4) == 0) {
                    code.swap(blob);
                    rtc->destroy(&prog);
                    return 0;
                }
            }
        }
    }
    r = rtc->compile(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    rtc->log_size(prog, &ls);
    if (ls > 1) { log.resize(ls); rtc->log(prog, &log[0]); }
    if (r == HIPRTC_SUCCESS) {
        size_t cs = 0;
        rtc->code_size(prog, &cs);
        code.resize(cs);
        rtc->code(prog, code.data());
        if (!cache_file.empty()) { // written under a temporary name and renamed: a reader never sees half a file
            const std::string tmp = cache_file + ".tmp" + std::to_string((long)getpid());
            if (FILE* f = fopen(tmp.c_str(), "wb")) {
                const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
                fclose(f);
                if (!ok || rename(tmp.c_str(), cache_file.c_str()) != 0) (void)remove(tmp.c_str());
            }
        }
    }
    rtc->destroy(&prog);
    return (int)r;
}

// preprocessing (align / features / fused forward) for n_frames starting at x -> out
int launch_pre(molann_plan* p, const float* x, long n_frames, float* out, int mode, bool with_mlp,
               hipStream_t stream, float* feat_out = nullptr) {
    const int out_cols = mode == 1 ? 0 : (with_mlp ? p->out_dim : p->d_feat);
    PreArgs a;
    fill_pre_args(p, a, n_frames, mode, out_cols, with_mlp, x, out);
    // the plan-specialised lane kernel first: it serves every plan it was built for, including few-atom plans on
    // frames too large for the ahead-of-time lane kernel's dense tile
    const bool jit_allowed = mode == 0 && (a.ablate & ~(32 | 64 | 128 | 256 | 512 | 1024 | 2048)) == 0;
    auto launch_jit = [&](hipFunction_t fn, int nl, int ncons, int nload, int nslot, int bpc, int lds_block) {
        const long n_tiles = (n_frames + 63) / 64;
        const int jgrid = grid_for(p, n_tiles, 1, bpc);   // every block needs at least one tile
        const size_t jlds = (size_t)debug_env().lds_pad;   // the kernel declares its block's LDS statically
        unsigned long long* stamps = nullptr;
        if (a.ablate & 32) (void)hipGetSymbolAddress((void**)&stamps, HIP_SYMBOL(g_stamps));
        struct { const float* x; float* out; const double* ref64; const float* wfrag; long n; int out_vec4, pad_;
                 unsigned long long* stamps; const float* ref32; float* feat; } ka = {x, out, p->d_ref64, p->d_wlane, n_frames, a.out_vec4, 0,
                                                                                     stamps, p->d_ref, feat_out};
        size_t ksz = sizeof(ka);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksz, HIP_LAUNCH_PARAM_END};
        const int jblock = 64 * (ncons + nload);
        const hipError_t le = hipModuleLaunchKernel(fn, jgrid, 1, 1, jblock, 1, 1, (unsigned)jlds, stream, nullptr, cfg);
        snprintf(p->last_info, sizeof(p->last_info), "molann_lane_jit<NL=%d> (plan-specialised; %d consumer waves + %d loader, ring of %d tiles) grid=%d block=%d lds=%d",
                 nl, ncons, nload, nslot, jgrid, jblock, lds_block);
        return (int)le;
    };
    if (feat_out) { // training forward: the fused kernel's twin that keeps the features, built now
        if (!(jit_allowed && with_mlp && p->jit_fn && p->jit_nl == p->n_layers && p->jit_nl > 0 && p->spec)) return MOLANN_E_UNSUPPORTED;
        if (p->train_state == 0) {
            std::lock_guard<std::mutex> lock(*p->jit_mu);
            if (p->train_state == 0) {
                JitSpec j = p->spec->j;
                j.save_feat = true;
                std::vector<char> code;
                std::string log;
                int st = -1;
                if (jit_compile(jit_source(j), code, log, "-fno-slp-vectorize") == 0 && hipModuleLoadData(&p->train_mod, code.data()) == hipSuccess &&
                    hipModuleGetFunction(&p->train_fn, p->train_mod, "molann_lane_jit") == hipSuccess)
                    st = 1;
                else if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann training-forward jit failed\n%s\n", log.c_str());
                p->train_state = st;
            }
        }
        if (p->train_state != 1) return MOLANN_E_UNSUPPORTED;
        return launch_jit(p->train_fn, p->jit_nl, p->jit_ncons, p->jit_nload, p->jit_nslot, p->jit_bpc, p->jit_lds_block);
    }
    if (jit_allowed && p->jit_fn && p->jit_nl == (with_mlp ? p->n_layers : 0))
        return launch_jit(p->jit_fn, p->jit_nl, p->jit_ncons, p->jit_nload, p->jit_nslot, p->jit_bpc, p->jit_lds_block);
    if (jit_allowed && !with_mlp && p->jit_fn && p->jit_nl > 0 && p->spec && p->feat_state >= 0) {
        // features of a plan whose forward kernel has the MLP fused in: the same kernel text without the MLP, built now
        if (p->feat_state == 0) {
            std::lock_guard<std::mutex> lock(*p->jit_mu);
            if (p->feat_state == 0) {
                JitSpec j = p->spec->j;
                j.n_layers = 0; j.out_cols = j.d_feat; j.dims.clear();
                molann_plan::LaneGeom g;
                memset(&g, 0, sizeof(g));
                jit_geometry(j, g, std::max(1, j.d_feat), std::max(1, j.d_feat));
                std::vector<char> code;
                std::string log;
                int st = -1;
                if (g.ok && jit_compile(jit_source(j), code, log, "-fno-slp-vectorize") == 0 &&
                    hipModuleLoadData(&p->feat_mod, code.data()) == hipSuccess &&
                    hipModuleGetFunction(&p->feat_fn, p->feat_mod, "molann_lane_jit") == hipSuccess) {
                    p->feat_ncons = j.ncons; p->feat_nload = j.nload; p->feat_nslot = j.nslot; p->feat_bpc = j.bpc; p->feat_lds_block = j.lds_block;
                    st = 1;
                } else if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann features jit failed\n%s\n", log.c_str());
                p->feat_state = st;
            }
        }
        if (p->feat_state == 1)
            return launch_jit(p->feat_fn, 0, p->feat_ncons, p->feat_nload, p->feat_nslot, p->feat_bpc, p->feat_lds_block);
    }
    const molann_plan::LaneGeom& g = p->geom[mode == 1 ? 1 : 0];
    if (mode == 0 && p->jit_only && (with_mlp || !g.ok)) return MOLANN_E_UNSUPPORTED; // served by the specialised kernel only, and a diagnostic switch excluded it
    if (g.ok) {
        const int wpb = g.wpb;
        const long n_tiles = (n_frames + 63) / 64;
        int bpc = (int)(163840 / ((long)wpb * g.lds_per_wave));
        if (bpc < 1) bpc = 1;
        if (bpc * wpb > 16) bpc = std::max(1, 16 / wpb);
        const int grid = grid_for(p, n_tiles, wpb, bpc);
        size_t lds = (size_t)wpb * g.lds_per_wave;
        lds += (size_t)debug_env().lds_pad; // diagnostic: lower the occupancy
        const dim3 block(64 * wpb);
        const int w = with_mlp ? p->n_layers : 0;
        const bool regs = p->regs_mode;
#define LAUNCH_LANE(W, M)                                                                                         \
    hipLaunchKernelGGL((frames_lane_kernel<W, M>), dim3(grid), block, lds, stream, x, out, p->d_align_idx, p->d_ref, \
                       p->d_ref64, (M == 2 ? p->d_items_slot : p->d_items), p->d_slots, p->d_wlane, a)
#define LAUNCH_LANE_W(M)            \
    if (w == 0) LAUNCH_LANE(0, M);  \
    else if (w == 1) LAUNCH_LANE(1, M); \
    else if (w == 2) LAUNCH_LANE(2, M); \
    else if (w == 3) LAUNCH_LANE(3, M); \
    else LAUNCH_LANE(4, M)
        if (mode == 1) LAUNCH_LANE(0, 1);
        else if (regs) { LAUNCH_LANE_W(2); }
        else { LAUNCH_LANE_W(0); }
#undef LAUNCH_LANE_W
#undef LAUNCH_LANE
        snprintf(p->last_info, sizeof(p->last_info), "frames_lane_kernel<%d,%s> grid=%d block=%d lds=%zu", w,
                 mode == 1 ? "align_out" : (regs ? "features_regs" : "features_lds"), grid, 64 * wpb, lds);
    } else if (mode == 0 && p->ring_nd > 0 && getenv("MOLANN_NO_RING") == nullptr) {
        // large frames, features: every frame staged once into an LDS ring by loader waves (frames_ring_kernel)
        const int nd = p->ring_nd;
        const int img = nd * 1024;
        RingArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.n_frames = n_frames; ra.frame_bytes = 12 * p->n_inp; ra.n_align = p->n_align; ra.n_items = p->n_items;
        ra.out_cols = a.out_cols; ra.n_win = p->ring_nwin;
        ra.n_slot = std::min(16, (163840 - RING_HEADER) / img);
        ra.n_load = ra.n_slot >= 6 ? 2 : 1;
        // a consumer keeps its slot while it computes from the image: slots = consumers + what the loaders keep in flight
        ra.n_cons = std::max(1, std::min(14, ra.n_slot - 2 * ra.n_load));
        ra.depth = std::max(0, std::min(std::min(6, 63 / nd), (ra.n_slot - ra.n_cons) / ra.n_load - 1));
        if (debug_env().wave_bpc > 0) ra.n_cons = std::max(1, std::min(ra.n_cons, debug_env().wave_bpc)); // MOLANN_WAVE_BPC: experiment with fewer consumers
        { const char* e = getenv("MOLANN_RING_NT"); ra.nt = e ? (e[0] == '1') : 1; }
        const int block = 64 * (ra.n_cons + ra.n_load);
        const size_t lds = (size_t)RING_HEADER + (size_t)ra.n_slot * img;
        const int grid = (int)std::min<long>(n_frames, p->num_cus);
#define LAUNCH_RING(N)                                                                                                    \
    case N: {                                                                                                             \
        static bool attr_##N = false;                                                                                     \
        if (!attr_##N) { (void)hipFuncSetAttribute((const void*)frames_ring_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); attr_##N = true; } \
        hipLaunchKernelGGL(frames_ring_kernel<N>, dim3(grid), dim3(block), lds, stream, x, out, p->d_ring_win, p->d_ring_align_pos, \
                           p->d_ref, p->d_ref64, p->d_ring_items, ra);                                                     \
    } break;
        switch (nd) {
            LAUNCH_RING(2) LAUNCH_RING(4) LAUNCH_RING(6) LAUNCH_RING(8) LAUNCH_RING(10) LAUNCH_RING(12) LAUNCH_RING(14) LAUNCH_RING(16)
            LAUNCH_RING(18) LAUNCH_RING(20) LAUNCH_RING(22) LAUNCH_RING(24) LAUNCH_RING(26) LAUNCH_RING(28) LAUNCH_RING(30) LAUNCH_RING(32)
        default: return MOLANN_E_UNSUPPORTED;
        }
#undef LAUNCH_RING
        snprintf(p->last_info, sizeof(p->last_info), "frames_ring_kernel<ND=%d> (%d consumer + %d loader waves, ring of %d frames, %d windows) grid=%d block=%d lds=%zu",
                 nd, ra.n_cons, ra.n_load, ra.n_slot, p->ring_nwin, grid, block, lds);
    } else if (mode == 1 && getenv("MOLANN_NO_RING") == nullptr && p->n_inp <= 24 * 512) {
        // AlignmentLayer.forward on large frames: the frame held in the registers of a block of W waves (molann_align_regs.inc)
        AlignRegsArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.n_frames = n_frames; ra.n_inp = p->n_inp; ra.n_align = p->n_align;
        ra.first_align = p->align_first;
        { const char* e = diag_env("MOLANN_DEBUG_ALIGN_FLAGS"); ra.flags = e ? atoi(e) : 0; }
        int W = 1;
        while ((p->n_inp + 64 * W - 1) / (64 * W) > 24) W *= 2;
        const int units = ((p->n_inp + 64 * W - 1) / (64 * W) + 3) / 4;       // atoms per thread, in fours
        const void* fn = nullptr;
#define ALIGN_REGS_CASE(WW, U) if (W == WW && units == U) fn = (const void*)frames_align_regs_kernel<WW, U>;
        ALIGN_REGS_CASE(1, 1) ALIGN_REGS_CASE(1, 2) ALIGN_REGS_CASE(1, 3) ALIGN_REGS_CASE(1, 4) ALIGN_REGS_CASE(1, 5) ALIGN_REGS_CASE(1, 6)
        ALIGN_REGS_CASE(2, 4) ALIGN_REGS_CASE(2, 5) ALIGN_REGS_CASE(2, 6) ALIGN_REGS_CASE(4, 4) ALIGN_REGS_CASE(4, 5) ALIGN_REGS_CASE(4, 6)
        ALIGN_REGS_CASE(8, 4) ALIGN_REGS_CASE(8, 5) ALIGN_REGS_CASE(8, 6)
#undef ALIGN_REGS_CASE
        if (!fn) return MOLANN_E_UNSUPPORTED;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64 * (W + 1), 0) != hipSuccess || occ < 1) occ = 1;
        // frames in flight per CU: two blocks of the 5000-atom class (measured, A4: 7.9 / 6.4 / 7.4 ms per 262 144 frames at 1 / 2 / 3
        // blocks per CU), more of the smaller ones: about 120 KB of frames per CU
        occ = std::min(occ, std::max(2, (int)(122880 / (12l * p->n_inp))));
        if (debug_env().wave_bpc > 0) occ = std::min(occ, debug_env().wave_bpc);
        const int grid = (int)std::min<long>(n_frames, (long)p->num_cus * occ);
        void* kargs[] = {(void*)&x, (void*)&out, (void*)&p->d_align_idx, (void*)&p->d_ref, (void*)&p->d_ref64, (void*)&ra};
        const hipError_t le = hipLaunchKernel(fn, dim3(grid), dim3(64 * (W + 1)), kargs, 0, stream);
        snprintf(p->last_info, sizeof(p->last_info), "frames_align_regs_kernel<W=%d,U=%d> (%d data waves + 1 solver per frame, %d atoms per thread, %d blocks per CU) grid=%d block=%d",
                 W, units, W, 4 * units, occ, grid, 64 * (W + 1));
        if (le != hipSuccess) return (int)le;
    } else {
        const int wpb = 4;
        // blocks per CU: all wave slots.  (The gather is HBM-latency bound and wants every wave it can get; the MLP
        // kernel on the plan's side stream takes the CUs it needs as gather blocks retire.  Measured, 262144
        // frames: C5 5.8e7 / 6.8e7 / 7.3e7 frames/s at 2 / 4 / 8 blocks per CU.  MOLANN_WAVE_BPC to experiment.)
        int bpc = 8;
        if (debug_env().wave_bpc > 0) bpc = debug_env().wave_bpc;
        const int grid = grid_for(p, n_frames, wpb, bpc);
        // item rounds (64 items each) whose atoms are loaded with the alignment atoms.  Measured (PMC, C5, 256 items):
        // 4 rounds: 519 TCC misses/frame, 279 us per 21 845 frames; two-phase: 603 misses/frame, 291 us.  With one
        // round or less (C4) the extra registers cost more occupancy than the shared lines save: two-phase.
        int pre = (mode == 1 || p->n_items <= 64) ? 0 : (p->n_items <= 128 ? 2 : 4);
        if (debug_env().wave_pre >= 0) pre = debug_env().wave_pre;
#define LAUNCH_WAVE(P)                                                                                                    \
    hipLaunchKernelGGL(frames_wave_kernel<P>, dim3(grid), dim3(64 * wpb), 0, stream, x, out, p->d_align_idx, p->d_ref, \
                       p->d_ref64, p->d_items, a)
        if (pre >= 4) LAUNCH_WAVE(4);
        else if (pre >= 2) LAUNCH_WAVE(2);
        else if (pre == 1) LAUNCH_WAVE(1);
        else LAUNCH_WAVE(0);
#undef LAUNCH_WAVE
        snprintf(p->last_info, sizeof(p->last_info), "frames_wave_kernel<pre=%d> grid=%d block=%d mode=%d", pre >= 4 ? 4 : (pre >= 2 ? 2 : pre),
                 grid, 64 * wpb, mode);
    }
    return (int)hipGetLastError();
}

int launch_mlp(molann_plan* p, const float* feat, long n_frames, int in_stride, float* out, hipStream_t stream) {
    if (p->chain_fn) { // plan-specialised chain kernel: one 4-wave block per CU, 64*FB frames per block and tile
        const int cw = p->chain_waves > 0 ? p->chain_waves : 4;
        const long tile = 16l * cw * p->chain_fb, n_tiles = (n_frames + tile - 1) / tile;
        const int grid = (int)std::min<long>(n_tiles, p->num_cus);
        struct { const float* feat; float* out; const unsigned char* w; const float* b; long n; int in_stride; } ka =
            {feat, out, p->d_wchain, (const float*)(p->d_wchain + p->chain_stream_bytes), n_frames, in_stride};
        size_t ksz = sizeof(ka);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksz, HIP_LAUNCH_PARAM_END};
        snprintf(p->mlp_info, sizeof(p->mlp_info), "molann_mlp_chain<%s,FB=%d%s> (plan-specialised) grid=%d block=%d",
                 p->mlp_prec == MOLANN_MLP_BF16 ? "bf16" : "f32", p->chain_fb, cw == 8 ? ",resident" : "", grid, 64 * cw);
        return (int)hipModuleLaunchKernel(p->chain_fn, grid, 1, 1, 64 * cw, 1, 1, 0, stream, nullptr, cfg);
    }
    MlpArgs a;
    memset(&a, 0, sizeof(a));
    a.n_frames = n_frames;
    a.n_layers = p->n_layers;
    a.act = p->act;
    for (int i = 0; i <= p->n_layers; ++i) a.dims[i] = p->dims[i];
    for (int i = 0; i < p->n_layers; ++i) { a.kp[i] = p->kp[i]; a.jp[i] = p->jp[i]; a.woff[i] = p->moff[i]; }
    a.ld[0] = p->mlp_ld[0];
    a.ld[1] = p->mlp_ld[1];
    a.lds_per_wave = p->mlp_lds_per_wave;
    a.in_stride = in_stride;
    // block size that packs most waves into the CU's LDS (waves are independent), larger block on ties
    int wpb = 1, best_waves = 0;
    for (int w = 4; w >= 1; --w) {
        if ((long)w * p->mlp_lds_per_wave > 65536 && w > 1) continue;
        long waves = w * (163840 / ((long)w * p->mlp_lds_per_wave));
        if (waves > 16) waves = 16;
        if (waves > best_waves) { best_waves = (int)waves; wpb = w; }
    }
    int bpc = (int)(163840 / ((long)wpb * p->mlp_lds_per_wave));
    if (bpc < 1) bpc = 1;
    if (bpc * wpb > 16) bpc = std::max(1, 16 / wpb);
    const long n_blocks = (n_frames + 15) / 16;
    const int grid = grid_for(p, n_blocks, wpb, bpc);
    const size_t lds = (size_t)wpb * p->mlp_lds_per_wave;
    snprintf(p->mlp_info, sizeof(p->mlp_info), "mlp_mfma_kernel<%s> grid=%d block=%d lds=%zu", p->mlp_prec == MOLANN_MLP_BF16 ? "bf16" : "f32",
             grid, 64 * wpb, lds);
    if (p->mlp_prec == MOLANN_MLP_BF16)
        hipLaunchKernelGGL((mlp_mfma_kernel<true>), dim3(grid), dim3(64 * wpb), lds, stream, feat, out, p->d_wmfma, a);
    else
        hipLaunchKernelGGL((mlp_mfma_kernel<false>), dim3(grid), dim3(64 * wpb), lds, stream, feat, out, p->d_wmfma, a);
    return (int)hipGetLastError();
}

} // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int molann_abi_version(void) { return MOLANN_ABI_VERSION; }

const char* molann_build_kind(void) {
#ifdef MOLANN_DIAGNOSTICS
    return "diagnostics";
#else
    return "release";
#endif
}

const char* molann_error_string(int code) {
    switch (code) {
    case MOLANN_OK: return "ok";
    case MOLANN_E_NULL: return "required pointer is NULL";
    case MOLANN_E_DESC: return "inconsistent plan description";
    case MOLANN_E_INDEX: return "atom index outside [0, n_inp)";
    case MOLANN_E_FEATURE: return "unknown feature type or wrong atom count for its type";
    case MOLANN_E_STAGE: return "plan lacks the stage this call needs";
    case MOLANN_E_ALIGNMENT: return "pointer is not 4-byte aligned";
    case MOLANN_E_UNSUPPORTED: return "shape not covered by the gfx950 kernels";
    case MOLANN_E_NOT_PACKED: return "MLP weights were never packed (call molann_plan_update_mlp)";
    case MOLANN_E_DEVICE: return "no gfx950 device";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int molann_plan_create(const molann_plan_desc* d, molann_plan** out_plan) {
    if (!out_plan) return MOLANN_E_NULL;
    *out_plan = nullptr;
    const int v = validate_desc(d);
    if (v != MOLANN_OK) return v;

    // ---- expand the feature list into items (column order = list order, ann.py:473) ------------
    std::vector<ItemDev> items;
    int col = 0;
    for (int f = 0; f < d->n_features; ++f) {
        const int* idx = d->feat_idx + d->feat_ptr[f];
        const int cnt = d->feat_ptr[f + 1] - d->feat_ptr[f];
        const int t = d->feat_type[f];
        if (t == MOLANN_FEAT_POSITION) {
            for (int i = 0; i < cnt; ++i) {
                ItemDev it = {IT_POSITION, col, {idx[i], idx[i], idx[i], idx[i]}, {0, 0}};
                items.push_back(it);
                col += 3;
            }
        } else {
            ItemDev it;
            it.type = t == MOLANN_FEAT_ANGLE ? (d->use_angle_value ? IT_ANGLE_VAL : IT_ANGLE_COS)
                      : t == MOLANN_FEAT_BOND ? IT_BOND
                                              : (d->use_angle_value ? IT_DIHEDRAL_VAL : IT_DIHEDRAL_CS);
            it.col = col;
            for (int i = 0; i < 4; ++i) it.idx[i] = idx[i < cnt ? i : 0];
            it.pad[0] = it.pad[1] = 0;
            items.push_back(it);
            col += item_width(it.type);
        }
    }
    const int d_feat = col;
    if (d->n_layers > 0 && d->n_features > 0 && d->layer_dims[0] != d_feat) return MOLANN_E_DESC;

    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));

    molann_plan* p = new (std::nothrow) molann_plan();
    if (!p) return (int)hipErrorOutOfMemory;
    memset(p, 0, sizeof(*p));
    p->launch_mu = new std::mutex();
    p->jit_mu = new std::mutex();
    p->device = dev;
    p->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p->n_inp = d->n_inp;
    p->n_align = d->n_align;
    p->align_first = d->n_align > 0 ? d->align_idx[0] : 0;
    p->n_features = d->n_features;
    p->n_items = (int)items.size();
    p->d_feat = d_feat;
    p->use_angle_value = d->use_angle_value;
    p->n_layers = d->n_layers;
    p->act = d->activation;
    p->mlp_prec = d->mlp_precision;
    if (d->n_layers > 0)
        for (int i = 0; i <= d->n_layers; ++i) p->dims[i] = d->layer_dims[i];
    p->out_dim = d->n_layers > 0 ? d->layer_dims[d->n_layers] : d_feat;

    // ---- kernel family and MLP placement -------------------------------------------------------
    int max_w = 0;
    for (int i = 1; i <= d->n_layers; ++i) max_w = std::max(max_w, d->layer_dims[i]);
    const bool cheap_act = d->activation != MOLANN_ACT_ELU && d->activation != MOLANN_ACT_SOFTPLUS &&
                           d->activation != MOLANN_ACT_GELU;
    const bool small_mlp = d->n_layers > 0 && d->n_layers <= LANE_MLP_MAX_LAYERS && max_w <= LANE_MLP_MAX_WIDTH &&
                           d_feat <= LANE_MLP_MAX_WIDTH && d->mlp_precision == MOLANN_MLP_F32 && cheap_act;
    // feature dims 33..64 in front of such an MLP: fused too, by the plan-specialised kernel only (16 k-steps in layer 0)
    const bool wide_in_mlp = !small_mlp && d->n_layers > 0 && d->n_layers <= LANE_MLP_MAX_LAYERS && max_w <= LANE_MLP_MAX_WIDTH &&
                             d_feat <= 2 * LANE_MLP_MAX_WIDTH && d->mlp_precision == MOLANN_MLP_F32 && cheap_act;
    const int cols_needed = std::max(1, small_mlp ? ceil_to(d_feat, 4) : d_feat);
    // touched atoms -> slots in first-use order: align atoms, then the feature table's atoms
    std::vector<int> slot_of(d->n_inp, -1), slots;
    auto slot = [&](int atom) {
        if (slot_of[atom] < 0) { slot_of[atom] = (int)slots.size(); slots.push_back(atom); }
        return slot_of[atom];
    };
    bool align_is_prefix = true; // align atom i must be slot i (no repeated align atoms)
    for (int i = 0; i < d->n_align; ++i) align_is_prefix = align_is_prefix && (slot(d->align_idx[i]) == i);
    std::vector<ItemDev> items_slot(items);
    for (auto& it : items_slot)
        for (int i = 0; i < 4; ++i) it.idx[i] = slot(it.idx[i]);
    p->n_slots = (int)slots.size();
    // (plan creation is setup time: MOLANN_NO_REGS / MOLANN_NO_JIT select the other generic modes here)
    p->regs_mode = p->n_items > 0 && p->n_items <= 64 && p->n_slots <= 16 && align_is_prefix &&
                   getenv("MOLANN_NO_REGS") == nullptr;
    memset(p->geom, 0, sizeof(p->geom));
    const bool lane_tables_fit = d->n_align <= 64 && (long)d->n_inp * 768 <= 65536;
    if (p->n_items > 0 && lane_tables_fit && cols_needed <= LANE_MAX_COLS)
        lane_geometry(p->geom[0], 64 * d->n_inp * 12, cols_needed);
    if (d->n_align > 0 && lane_tables_fit) lane_geometry(p->geom[1], 64 * d->n_inp * 12, 1);
    // The plan-specialised lane kernel stages only the touched 16-byte windows of a frame, so its tile does not grow
    // with n_inp: a plan that touches few atoms (<= 32) of a LARGE frame is a lane-per-frame plan too, as long as
    // hipRTC is there to build it (the ahead-of-time lane kernel needs the dense tile and cannot serve it).
    const char* nojit_env = getenv("MOLANN_NO_JIT");
    bool jit_possible = rtc_api()->ok && !(nojit_env && nojit_env[0] == '1') && p->n_items > 0 &&
                        p->n_items <= JIT_MAX_ITEMS && p->n_slots <= JIT_MAX_SLOTS && align_is_prefix &&
                        cols_needed <= LANE_MAX_COLS && 3 * d->n_inp >= 4;
    if (jit_possible) { // and its LDS geometry (compact tile + staging rows) must leave room for >= 4 waves per CU
        JitSpec probe;
        probe.win = compact_windows(slots, d->n_inp);
        molann_plan::LaneGeom pg;
        memset(&pg, 0, sizeof(pg));
        jit_geometry(probe, pg, ((small_mlp || wide_in_mlp) && d->n_features > 0) ? d_feat : cols_needed, cols_needed);
        jit_possible = pg.ok != 0;
    }
    const bool lane_by_jit_only = jit_possible && !p->geom[0].ok;
    const bool fused_by_jit_only = jit_possible && wide_in_mlp && d->n_features > 0;
    // the family names the kernel that serves the plan's main product (features if it has any)
    p->family = (p->n_items > 0 ? (p->geom[0].ok || lane_by_jit_only) : p->geom[1].ok) ? 0 : 1;
    p->fused_mlp = ((p->family == 0) && small_mlp && d->n_features > 0) || fused_by_jit_only;
    p->jit_only = lane_by_jit_only || fused_by_jit_only;

    // ---- device blob ----------------------------------------------------------------------------
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_align = carve(sizeof(int) * std::max(1, d->n_align));
    const size_t o_ref = carve(sizeof(float) * (3 * (size_t)d->n_align + 8));
    const size_t o_ref64 = carve(sizeof(double) * (3 * (size_t)d->n_align + 8));
    const size_t o_items = carve(sizeof(ItemDev) * std::max<size_t>(1, items.size()));
    const size_t o_items_slot = carve(sizeof(ItemDev) * std::max<size_t>(1, items.size()));
    const size_t o_slots = carve(sizeof(int) * std::max<size_t>(1, slots.size()));
    // frames_ring_kernel tables (plans the lane kernels do not serve): windows, image positions
    std::vector<int> ring_win, ring_align_pos;
    std::vector<ItemDev> ring_items;
    p->ring_nd = 0;
    if (p->family == 1 && p->n_items > 0 && align_is_prefix) {
        const std::vector<int> win = compact_windows(slots, d->n_inp);
        static const int buckets[] = {2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32};
        int nd = 0;
        for (int b : buckets)
            if (nd == 0 && (long)b * 64 >= (long)win.size()) nd = b;
        if (nd > 0) {
            // dword position of the atom's x inside the image; its three dwords are contiguous there: either one
            // window holds all of them (always so for the window clamped to the frame's end, which may overlap its
            // predecessor), or the atom runs over the end of window k and window k + 1 starts right behind it
            auto pos_of = [&](int atom) {
                const int d0 = 3 * atom;
                for (size_t k = 0; k < win.size(); ++k)
                    if (d0 >= win[k] && d0 + 2 < win[k] + 4) return (int)(4 * k) + d0 - win[k];
                for (size_t k = 0; k < win.size(); ++k)
                    if (d0 >= win[k] && d0 < win[k] + 4) return (int)(4 * k) + d0 - win[k];
                return 0;
            };
            for (int w : win) ring_win.push_back(4 * w);
            for (int i = 0; i < d->n_align; ++i) ring_align_pos.push_back(pos_of(d->align_idx[i]));
            ring_items = items;
            for (auto& it : ring_items)
                for (int i = 0; i < 4; ++i) it.idx[i] = pos_of(it.idx[i]);
            p->ring_nd = nd;
            p->ring_nwin = (int)win.size();
        }
    }
    const size_t o_ring_win = carve(sizeof(int) * std::max<size_t>(1, ring_win.size()));
    const size_t o_ring_apos = carve(sizeof(int) * std::max<size_t>(1, ring_align_pos.size()));
    const size_t o_ring_items = carve(sizeof(ItemDev) * std::max<size_t>(1, ring_items.size()));
    // backward of large frames without atomics: who contributes to which touched atom
    std::vector<int> bw_atoms, bw_ptr, bw_list, bw_align;
    if (!p->geom[0].ok && p->n_items > 0 && d->n_layers >= 0) {
        std::vector<std::vector<int>> contrib(d->n_inp);
        std::vector<int> al_of(d->n_inp, -1);
        std::vector<char> touched(d->n_inp, 0);
        for (size_t it = 0; it < items.size(); ++it)
            for (int j = 0; j < item_atoms(items[it].type); ++j) { contrib[items[it].idx[j]].push_back((int)(4 * it + j)); touched[items[it].idx[j]] = 1; }
        for (int i = 0; i < d->n_align; ++i) { if (al_of[d->align_idx[i]] < 0) al_of[d->align_idx[i]] = i; touched[d->align_idx[i]] = 1; }
        bool repeated_align = false;
        { std::vector<char> seen(d->n_inp, 0); for (int i = 0; i < d->n_align; ++i) { repeated_align = repeated_align || seen[d->align_idx[i]]; seen[d->align_idx[i]] = 1; } }
        if (!repeated_align) {   // (an alignment set that names an atom twice keeps the atomics: the atom has two reference rows)
            bw_ptr.push_back(0);
            for (int a0 = 0; a0 < d->n_inp; ++a0)
                if (touched[a0]) {
                    bw_atoms.push_back(a0);
                    bw_align.push_back(al_of[a0]);
                    bw_list.insert(bw_list.end(), contrib[a0].begin(), contrib[a0].end());
                    bw_ptr.push_back((int)bw_list.size());
                }
        }
    }
    p->bw_touched = (int)bw_atoms.size();
    const size_t o_bw_atoms = carve(sizeof(int) * std::max<size_t>(1, bw_atoms.size()));
    const size_t o_bw_ptr = carve(sizeof(int) * std::max<size_t>(1, bw_ptr.size()));
    const size_t o_bw_list = carve(sizeof(int) * std::max<size_t>(1, bw_list.size()));
    const size_t o_bw_align = carve(sizeof(int) * std::max<size_t>(1, bw_align.size()));

    size_t lane_floats = 0;
    if (p->fused_mlp) lane_floats = (size_t)d->n_layers * (1024 + 512) + 1024;
    const size_t o_wlane = carve(sizeof(float) * std::max<size_t>(1, lane_floats));
    size_t mfma_bytes = 0;
    const bool bf16 = d->mlp_precision == MOLANN_MLP_BF16;
    const int kgran = bf16 ? 32 : 16;
    int max_kp = 16;
    if (d->n_layers > 0) {
        const size_t es = bf16 ? 2 : 4;
        for (int l = 0; l < d->n_layers; ++l) {
            p->kp[l] = ceil_to(p->dims[l], kgran);
            p->jp[l] = ceil_to(p->dims[l + 1], 16);
            max_kp = std::max(max_kp, std::max(p->kp[l], p->jp[l]));
        }
        // the activations written by layer l (Jp columns) are read as layer l+1's Kp columns
        for (int l = 0; l + 1 < d->n_layers; ++l) max_kp = std::max(max_kp, p->kp[l + 1]);
        for (int l = 0; l < d->n_layers; ++l) {
            p->moff[l] = (long)(mfma_bytes / es);
            mfma_bytes += ((size_t)p->jp[l] * p->kp[l]) * es + (size_t)p->jp[l] * 4;
            mfma_bytes = (mfma_bytes + 15) & ~(size_t)15;
        }
        // two activation buffers: [0] holds the inputs of even layers, [1] of odd layers (layer l writes what
        // layer l+1 reads).  Row strides: 16-byte multiples, off the power of two.
        int need[2] = {16, 16};
        for (int l = 0; l < d->n_layers; ++l) {
            need[l & 1] = std::max(need[l & 1], p->kp[l]);
            if (l + 1 < d->n_layers) need[(l + 1) & 1] = std::max(need[(l + 1) & 1], std::max(p->jp[l], p->kp[l + 1]));
        }
        (void)max_kp;
        for (int i = 0; i < 2; ++i) p->mlp_ld[i] = need[i] + (bf16 ? 8 : 4);
        p->mlp_lds_per_wave = 16 * (p->mlp_ld[0] + p->mlp_ld[1]) * (int)es;
        if (p->mlp_lds_per_wave > 163840) { delete p; return MOLANN_E_UNSUPPORTED; }
        if (p->mlp_lds_per_wave > 65536) { // a single wave's two activation buffers exceed the default 64 KiB cap
            hipError_t ea = bf16 ? hipFuncSetAttribute((const void*)mlp_mfma_kernel<true>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
                                 : hipFuncSetAttribute((const void*)mlp_mfma_kernel<false>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            if (ea != hipSuccess) { delete p; return (int)ea; }
        }
    }
    const size_t o_wmfma = carve(std::max<size_t>(16, mfma_bytes));
    // wide bf16 MLP next to a gather kernel: the chain kernel's weight stream (molann_mlp_jit.inc), when its
    // two LDS slabs fit and at least one 16-frame block per wave fits the register file
    ChainGeom cg;
    memset(&cg, 0, sizeof(cg));
    size_t chain_bytes = 0;
    int chain_fb = 0;
    if (d->n_layers > 0 && !p->fused_mlp) {
        cg.nl = d->n_layers;
        cg.bf16 = bf16 ? 1 : 0;
        for (int i = 0; i <= d->n_layers; ++i) cg.dims[i] = p->dims[i];
        for (int fb = 4; fb >= 1 && chain_fb == 0; --fb) // upper bound; plan creation steps down while the build spills
            if (fb * cg.regs_per_fb() <= (chain_resident(cg) ? 256 : 512)) chain_fb = fb;     // resident: two waves per SIMD
        if (!chain_resident(cg) && 2 * cg.slab_max() * 1024 > 163840 - 1024) chain_fb = 0;
        if (chain_fb > 0) {
            p->chain_stream_bytes = cg.total_frags() * 1024;
            chain_bytes = (size_t)p->chain_stream_bytes + (size_t)cg.bias_off(cg.nl) * 4;
        }
    }
    const size_t o_wchain = carve(std::max<size_t>(16, chain_bytes));
    p->work_frames = 0;
    size_t work_bytes = 0;
    if (d->n_layers > 0 && d->n_features > 0 && !p->fused_mlp) {
        // feature chunk handed from the preprocessing kernel to the MLP kernel: sized to stay
        // resident in the 256 MiB Infinity Cache
        long wf = (64l << 20) / ((long)d_feat * 4);
        wf = std::max<long>(1024, std::min<long>(wf, 1l << 21)); // (narrow feature rows: few, large chunks - each costs ~6 host API calls, and a
                                                                  //  feature launch of 512 k frames takes 37 us where one of 1 M takes 46)
        wf &= ~63l;
        wf = std::max<long>(512, (wf / 2) & ~63l); // per half
        // Large frames: the feature rows are a few percent of the frame bytes, so letting them spill past the Infinity
        // Cache costs little, while a chunk that small leaves the MLP kernel (one block per CU, 64 FB frames per block
        // and step) a fraction of the chip: frames_ring_kernel holds every CU, the two kernels run one after the other,
        // and C5's MLP took 3.7 us per 1000 frames in 24 576-frame chunks against 1.2 on its own.  Up to 256 MiB per half.
        if (p->family == 1 && 12l * d->n_inp >= 32l * d_feat)
            wf = std::max<long>(wf, std::min<long>(1l << 18, ((256l << 20) / ((long)d_feat * 4)) & ~63l));
        p->work_frames = wf;
        work_bytes = 2 * (size_t)wf * d_feat * 4;
    }
    const size_t o_work = carve(std::max<size_t>(16, work_bytes));

    hipError_t e = hipMalloc((void**)&p->blob, off);
    if (e != hipSuccess) { delete p; return (int)e; }
    p->d_align_idx = (int*)(p->blob + o_align);
    p->d_ref = (float*)(p->blob + o_ref);
    p->d_ref64 = (double*)(p->blob + o_ref64);
    p->d_items = (ItemDev*)(p->blob + o_items);
    p->d_items_slot = (ItemDev*)(p->blob + o_items_slot);
    p->d_slots = (int*)(p->blob + o_slots);
    p->d_ring_win = (int*)(p->blob + o_ring_win);
    p->d_ring_align_pos = (int*)(p->blob + o_ring_apos);
    p->d_ring_items = (ItemDev*)(p->blob + o_ring_items);
    p->d_bw_atoms = (int*)(p->blob + o_bw_atoms); p->d_bw_ptr = (int*)(p->blob + o_bw_ptr);
    p->d_bw_list = (int*)(p->blob + o_bw_list); p->d_bw_align = (int*)(p->blob + o_bw_align);
    p->d_wlane = (float*)(p->blob + o_wlane);
    p->d_wmfma = (void*)(p->blob + o_wmfma);
    p->d_work = (float*)(p->blob + o_work);
    p->d_wchain = p->blob + o_wchain;

    // ---- upload (synchronous: plan creation is setup time) --------------------------------------
    if (d->n_align > 0) {
        std::vector<float> refc(3 * (size_t)d->n_align + 8, 0.f);
        std::vector<double> refd(3 * (size_t)d->n_align + 8, 0.);
        double s[4] = {0, 0, 0, 0};
        for (int i = 0; i < d->n_align; ++i) {
            for (int c = 0; c < 3; ++c) {
                const float r = d->ref_x[3 * i + c];
                refc[3 * i + c] = r;
                refd[3 * i + c] = r;
                s[c] += r;
                s[3] += (double)r * r;
            }
        }
        float* c = refc.data() + 3 * (size_t)d->n_align;
        double* c64 = refd.data() + 3 * (size_t)d->n_align;
        for (int k = 0; k < 4; ++k) { c[k] = (float)s[k]; c64[k] = s[k]; }
        c[4] = 1.0f / (float)d->n_align;
        c[5] = (float)d->n_align;
        c64[4] = 1.0 / (double)d->n_align;
        c64[5] = (double)d->n_align;
        e = hipMemcpy(p->d_align_idx, d->align_idx, sizeof(int) * d->n_align, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_ref, refc.data(), sizeof(float) * refc.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_ref64, refd.data(), sizeof(double) * refd.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !items.empty())
        e = hipMemcpy(p->d_items, items.data(), sizeof(ItemDev) * items.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && !items.empty())
        e = hipMemcpy(p->d_items_slot, items_slot.data(), sizeof(ItemDev) * items.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && !slots.empty())
        e = hipMemcpy(p->d_slots, slots.data(), sizeof(int) * slots.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->ring_nd > 0) {
        e = hipMemcpy(p->d_ring_win, ring_win.data(), sizeof(int) * ring_win.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && !ring_align_pos.empty())
            e = hipMemcpy(p->d_ring_align_pos, ring_align_pos.data(), sizeof(int) * ring_align_pos.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_ring_items, ring_items.data(), sizeof(ItemDev) * ring_items.size(), hipMemcpyHostToDevice);
    }

    if (e == hipSuccess && p->bw_touched > 0) {
        e = hipMemcpy(p->d_bw_atoms, bw_atoms.data(), sizeof(int) * bw_atoms.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_bw_ptr, bw_ptr.data(), sizeof(int) * bw_ptr.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && !bw_list.empty()) e = hipMemcpy(p->d_bw_list, bw_list.data(), sizeof(int) * bw_list.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_bw_align, bw_align.data(), sizeof(int) * bw_align.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) { (void)hipFree(p->blob); delete p; return (int)e; }
    snprintf(p->last_info, sizeof(p->last_info), "(no launch yet)");
    if (p->work_frames > 0) {
        e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
        for (int h = 0; h < 2 && e == hipSuccess; ++h) {
            e = hipEventCreateWithFlags(&p->ev_feat[h], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_mlp[h], hipEventDisableTiming);
            if (e == hipSuccess && h == 0) e = hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming);
        }
        if (e != hipSuccess) { (void)hipFree(p->blob); delete p; return (int)e; }
    }
    // ---- plan-specialised lane kernel --------------------------------------------------------------
    snprintf(p->jit_note, sizeof(p->jit_note), "jit: not applicable");
    const char* nojit = nojit_env;
    if (jit_possible) {
        JitSpec j;
        j.n_inp = d->n_inp; j.n_align = d->n_align; j.act = d->activation; j.d_feat = d_feat;
        j.n_layers = p->fused_mlp ? d->n_layers : 0;
        j.out_cols = p->fused_mlp ? p->out_dim : d_feat;
        j.slots = slots; j.items = items_slot;
        // compact tile: only the 16-byte windows of a frame that hold a touched atom go to LDS, so more waves fit
        j.win = compact_windows(slots, d->n_inp);
        jit_geometry(j, p->jit_geom, p->fused_mlp ? d_feat : cols_needed, cols_needed);
        if (p->fused_mlp) j.dims.assign(p->dims, p->dims + d->n_layers + 1);
        p->spec = new (std::nothrow) JitSpecBox();
        if (p->spec) {
            p->spec->j = j;
            for (int l = 0; l < j.n_layers; ++l) { p->spec->kp.push_back(p->kp[l]); p->spec->jp.push_back(p->jp[l]); p->spec->woff.push_back(p->moff[l]); }
            p->n_grad_params = 0;
            for (int l = 0; l < j.n_layers; ++l) p->n_grad_params += p->dims[l + 1] * p->dims[l] + p->dims[l + 1];
        }
        int rc = -1;
        const bool frame_ok = p->jit_geom.ok && 3 * d->n_inp >= 4;
        if (frame_ok) {
            std::vector<char> code;
            std::string log;
            // no SLP vectorisation: hipcc otherwise packs a fifth of this straight-line fp32 code into v_pk_* pairs, which
            // buys ~1.2x on those operations at two waves per SIMD and pays for it with ~130 register moves per tile and 44
            // more registers (C3: 166 -> 122 VGPRs, 74 -> 70 us; tools/ab_flags.sh)
            rc = jit_compile(jit_source(j), code, log, "-fno-slp-vectorize");
            hipModule_t mod = nullptr;
            hipFunction_t fn = nullptr;
            bool loaded = rc == 0 && hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, "molann_lane_jit") == hipSuccess;
            int scratch = 0;
            if (loaded && j.ncons > 10 && hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn) == hipSuccess && scratch > 0) {
                // Three or four layers of 32 units: the weight fragments do not fit the 128 registers of four waves per SIMD and the
                // build spills.  Ten consumers (three waves per SIMD, 168 registers) serve the stream as well as fourteen.
                JitSpec j3 = j;
                molann_plan::LaneGeom g3;
                memset(&g3, 0, sizeof(g3));
                jit_geometry(j3, g3, p->fused_mlp ? d_feat : cols_needed, cols_needed, 10);
                std::vector<char> code3;
                hipModule_t mod3 = nullptr;
                hipFunction_t fn3 = nullptr;
                int scratch3 = 0;
                if (g3.ok && jit_compile(jit_source(j3), code3, log, "-fno-slp-vectorize") == 0 && hipModuleLoadData(&mod3, code3.data()) == hipSuccess &&
                    hipModuleGetFunction(&fn3, mod3, "molann_lane_jit") == hipSuccess &&
                    hipFuncGetAttribute(&scratch3, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn3) == hipSuccess && scratch3 < scratch) {
                    (void)hipModuleUnload(mod);
                    mod = mod3; fn = fn3; code.swap(code3);
                    j = j3; p->jit_geom = g3;
                    if (p->spec) p->spec->j = j;
                } else if (mod3) {
                    (void)hipModuleUnload(mod3);
                }
            }
            if (!loaded) {
                if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann jit failed rc=%d\n%s\n", rc, log.c_str());
                if (mod) (void)hipModuleUnload(mod);
            } else {
                p->jit_mod = mod; p->jit_fn = fn; p->jit_nl = j.n_layers; p->jit_waves = 2;
                p->jit_ncons = j.ncons; p->jit_nload = j.nload; p->jit_nslot = j.nslot; p->jit_bpc = j.bpc; p->jit_lds_block = j.lds_block;
                snprintf(p->jit_note, sizeof(p->jit_note), "jit: specialised kernel, %d+1 waves per block, %zu bytes", j.ncons, code.size());
            }
        }
        if (!p->jit_fn) snprintf(p->jit_note, sizeof(p->jit_note), "jit: unavailable (rc=%d), generic kernel", rc);
    }
    if (p->jit_only && !p->jit_fn) { // hipRTC is present but the build failed: no other lane kernel for this plan
        if (p->fused_mlp) { // its MLP was planned into that kernel: nothing to fall back to
            molann_plan_destroy(p);
            return MOLANN_E_UNSUPPORTED;
        }
        p->jit_only = false;
        if (!p->geom[0].ok) p->family = 1; // features from the wave-per-frame kernel
    }
    // ---- AlignmentLayer.forward through the specialised kernel: the aligned frame is the feature row of one position item per
    // atom, so molann_align_f32 is the loader / consumer kernel too (built at its first call).  Small frames only.
    if (d->n_align > 0 && d->n_inp <= JIT_MAX_SLOTS && 3 * d->n_inp <= LANE_MAX_COLS && 3 * d->n_inp >= 4 && rtc_api()->ok &&
        !(nojit && nojit[0] == '1')) {
        JitSpec j;
        std::vector<int> seen(d->n_inp, 0);
        bool distinct = true;
        for (int i = 0; i < d->n_align; ++i) { distinct = distinct && !seen[d->align_idx[i]]; seen[d->align_idx[i]] = 1; j.slots.push_back(d->align_idx[i]); }
        for (int a = 0; a < d->n_inp; ++a) if (!seen[a]) j.slots.push_back(a);
        if (distinct) {
            for (int u = 0; u < d->n_inp; ++u) { ItemDev it = {IT_POSITION, 3 * j.slots[u], {u, u, u, u}, {0, 0}}; j.items.push_back(it); }
            j.n_inp = d->n_inp; j.n_align = d->n_align; j.act = 0; j.d_feat = 3 * d->n_inp; j.n_layers = 0; j.out_cols = 3 * d->n_inp;
            j.win = compact_windows(j.slots, d->n_inp);
            molann_plan::LaneGeom g;
            memset(&g, 0, sizeof(g));
            jit_geometry(j, g, j.d_feat, j.d_feat);
            if (g.ok && (p->align_spec = new (std::nothrow) JitSpecBox())) p->align_spec->j = j;
        }
    }
    // ---- plan-specialised wide bf16 MLP ----------------------------------------------------------------
    snprintf(p->chain_note, sizeof(p->chain_note), "chain: not applicable");
    if (chain_fb > 0 && !(nojit && nojit[0] == '1')) {
        // most frames per wave (A-fragment reuse) that the register file holds without scratch
        int rc = -1;
        for (int fb = chain_fb; fb >= 1 && !p->chain_fn; --fb) {
            std::vector<char> code;
            std::string log;
            rc = jit_compile(jit_source_chain(cg, p->act, fb), code, log);
            hipModule_t mod = nullptr;
            hipFunction_t fn = nullptr;
            if (rc != 0 || hipModuleLoadData(&mod, code.data()) != hipSuccess ||
                hipModuleGetFunction(&fn, mod, "molann_mlp_chain") != hipSuccess) {
                if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann chain jit failed rc=%d\n%s\n", rc, log.c_str());
                if (mod) (void)hipModuleUnload(mod);
                break;
            }
            int scratch = 0;
            (void)hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn);
            if (scratch > 0 && fb > 1) { (void)hipModuleUnload(mod); continue; }
            p->chain_mod = mod; p->chain_fn = fn; p->chain_fb = fb; p->chain_waves = chain_waves(cg);
            snprintf(p->chain_note, sizeof(p->chain_note), "chain: specialised kernel, FB=%d, %zu bytes", fb, code.size());
        }
        if (!p->chain_fn) snprintf(p->chain_note, sizeof(p->chain_note), "chain: unavailable (rc=%d), mlp_mfma_kernel", rc);
    }
    *out_plan = p;
    return MOLANN_OK;
}

int molann_plan_destroy(molann_plan* p) {
    if (!p) return MOLANN_OK;
    {   // kernels of this plan may still be running or queued (`y = model(x); del model`): its code objects and device
        // memory go only when the device has drained.  Destroying a plan is setup-time work, like creating one.
        int cur = -1;
        const bool sw = hipGetDevice(&cur) == hipSuccess && cur != p->device && hipSetDevice(p->device) == hipSuccess;
        (void)hipDeviceSynchronize();
        if (sw) (void)hipSetDevice(cur);
    }
    if (p->jit_mod) (void)hipModuleUnload(p->jit_mod);
    if (p->bwd_mod) (void)hipModuleUnload(p->bwd_mod);
    if (p->mbwd_mod) (void)hipModuleUnload(p->mbwd_mod);
    if (p->rbwd_mod) (void)hipModuleUnload(p->rbwd_mod);
    if (p->vjp_mod) (void)hipModuleUnload(p->vjp_mod);
    if (p->feat_mod) (void)hipModuleUnload(p->feat_mod);
    if (p->train_mod) (void)hipModuleUnload(p->train_mod);
    if (p->align_mod) (void)hipModuleUnload(p->align_mod);
    delete p->align_spec;
    if (p->d_bwork) (void)hipFree(p->d_bwork);
    if (p->d_gpart) (void)hipFree(p->d_gpart);
    if (p->ev_bwork) (void)hipEventDestroy(p->ev_bwork);
    if (p->chain_mod) (void)hipModuleUnload(p->chain_mod);
    delete p->spec;
    if (p->side) {
        (void)hipStreamSynchronize(p->side);
        for (int h = 0; h < 2; ++h) { (void)hipEventDestroy(p->ev_feat[h]); (void)hipEventDestroy(p->ev_mlp[h]); }
        (void)hipEventDestroy(p->ev_done);
        (void)hipStreamDestroy(p->side);
    }
    delete p->launch_mu;
    delete p->jit_mu;
    hipError_t e = hipFree(p->blob);
    delete p;
    return (int)e;
}

int molann_plan_feature_dim(const molann_plan* p) { return p ? p->d_feat : MOLANN_E_NULL; }
int molann_plan_out_dim(const molann_plan* p) { return p ? p->out_dim : MOLANN_E_NULL; }
int molann_plan_kernel_family(const molann_plan* p) { return p ? p->family : MOLANN_E_NULL; }

int molann_plan_last_launch_info(const molann_plan* p, char* buf, int cap) {
    if (!p || !buf || cap <= 0) return MOLANN_E_NULL;
    snprintf(buf, (size_t)cap, "%s", p->last_info);
    return (int)strlen(buf);
}

int molann_plan_update_ref(molann_plan* p, const float* ref_x, molann_stream_t stream) {
    if (!p || !ref_x) return MOLANN_E_NULL;
    if (p->n_align <= 0) return MOLANN_E_STAGE;
    hipLaunchKernelGGL(pack_ref_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, p->d_ref, p->d_ref64, ref_x, p->n_align);
    return (int)hipGetLastError();
}

int molann_plan_update_ref_f64(molann_plan* p, const double* ref_x, molann_stream_t stream) {
    if (!p || !ref_x) return MOLANN_E_NULL;
    if (p->n_align <= 0) return MOLANN_E_STAGE;
    hipLaunchKernelGGL(pack_ref_kernel<double>, dim3(1), dim3(256), 0, (hipStream_t)stream, p->d_ref, p->d_ref64, ref_x, p->n_align);
    return (int)hipGetLastError();
}

int molann_plan_update_mlp(molann_plan* p, const float* const* W, const float* const* b, molann_stream_t stream) {
    if (!p || !W || !b) return MOLANN_E_NULL;
    if (p->n_layers <= 0) return MOLANN_E_STAGE;
    PackArgs a;
    memset(&a, 0, sizeof(a));
    a.n_layers = p->n_layers;
    for (int i = 0; i <= p->n_layers; ++i) a.dims[i] = p->dims[i];
    for (int l = 0; l < p->n_layers; ++l) {
        if (!W[l] || !b[l]) return MOLANN_E_NULL;
        a.W[l] = W[l]; a.b[l] = b[l];
        a.kp[l] = p->kp[l]; a.jp[l] = p->jp[l]; a.moff[l] = p->moff[l];
    }
    a.fused = p->fused_mlp ? 1 : 0;
    a.bf16 = p->mlp_prec == MOLANN_MLP_BF16;
    if (p->fused_mlp)
        hipLaunchKernelGGL(pack_lane_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, p->d_wlane, a);
    // the MFMA copy serves molann_mlp_packed_f32 and the unfused forward
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(64, p->n_layers), dim3(256), 0, (hipStream_t)stream, p->d_wmfma, a);
    if (p->chain_fn) {
        ChainGeom g;
        memset(&g, 0, sizeof(g));
        g.nl = p->n_layers;
        g.bf16 = p->mlp_prec == MOLANN_MLP_BF16 ? 1 : 0;
        for (int i = 0; i <= p->n_layers; ++i) g.dims[i] = p->dims[i];
        ChainPackArgs c;
        memset(&c, 0, sizeof(c));
        c.n_layers = g.nl; c.npair = g.npair(); c.bf16 = g.bf16; c.stream_bytes = p->chain_stream_bytes;
        for (int i = 0; i <= g.nl; ++i) { c.dims[i] = g.dims[i]; c.bias_off[i] = g.bias_off(i); }
        for (int l = 0; l < g.nl; ++l) { c.W[l] = W[l]; c.b[l] = b[l]; }
        int start = 0;
        for (int q = 0; q < c.npair; ++q) {
            c.pair_start[q] = start; c.slab_frags[q] = g.slab_frags(q); c.ks_in[q] = g.ks(2 * q);
            start += g.nchunk(q) * g.slab_frags(q);
        }
        c.pair_start[c.npair] = start;
        hipLaunchKernelGGL(pack_chain_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, p->d_wchain, c);
    }
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) p->mlp_packed = true;
    return (int)e;
}

static int check_io(const void* x, const void* out, int64_t n) {
    if (n < 0) return MOLANN_E_DESC;
    if (n == 0) return MOLANN_OK;
    if (!x || !out) return MOLANN_E_NULL;
    if ((((uintptr_t)x) & 3) || (((uintptr_t)out) & 3)) return MOLANN_E_ALIGNMENT;
    return MOLANN_OK;
}

int molann_align_f32(const molann_plan* cp, const float* x, int64_t n, float* out_xyz, molann_stream_t stream) {
    if (!cp) return MOLANN_E_NULL;
    molann_plan* p = const_cast<molann_plan*>(cp);
    if (p->n_align <= 0) return MOLANN_E_STAGE;
    const int c = check_io(x, out_xyz, n);
    if (c != MOLANN_OK || n == 0) return c;
    if (p->align_spec && p->align_state >= 0 && (debug_env().ablate & ~(32 | 512)) == 0) {
        if (p->align_state == 0) {
            std::lock_guard<std::mutex> lock(*p->jit_mu);
            if (p->align_state == 0) {
                std::vector<char> code;
                std::string log;
                int st = -1;
                if (jit_compile(jit_source(p->align_spec->j), code, log, "-fno-slp-vectorize") == 0 && hipModuleLoadData(&p->align_mod, code.data()) == hipSuccess &&
                    hipModuleGetFunction(&p->align_fn, p->align_mod, "molann_lane_jit") == hipSuccess)
                    st = 1;
                else if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann alignment jit failed\n%s\n", log.c_str());
                p->align_state = st;
            }
        }
        if (p->align_state == 1) {
            const JitSpec& j = p->align_spec->j;
            const long n_tiles = (n + 63) / 64;
            const int grid = grid_for(p, n_tiles, 1, j.bpc);
            struct { const float* x; float* out; const double* ref64; const float* wfrag; long n; int out_vec4, pad_;
                     unsigned long long* stamps; const float* ref32; float* feat; } ka = {x, out_xyz, p->d_ref64, nullptr, (long)n, 0, 0, nullptr, p->d_ref, nullptr};
            size_t ksz = sizeof(ka);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksz, HIP_LAUNCH_PARAM_END};
            const int block = 64 * (j.ncons + j.nload);
            const hipError_t le = hipModuleLaunchKernel(p->align_fn, grid, 1, 1, block, 1, 1, 0, (hipStream_t)stream, nullptr, cfg);
            snprintf(p->last_info, sizeof(p->last_info), "molann_lane_jit<align_out> (plan-specialised; %d consumer waves + %d loader, ring of %d tiles) grid=%d block=%d lds=%d",
                     j.ncons, j.nload, j.nslot, grid, block, j.lds_block);
            return (int)le;
        }
    }
    return launch_pre(p, x, n, out_xyz, 1, false, (hipStream_t)stream);
}

int molann_features_f32(const molann_plan* cp, const float* x, int64_t n, float* out, molann_stream_t stream) {
    if (!cp) return MOLANN_E_NULL;
    molann_plan* p = const_cast<molann_plan*>(cp);
    if (p->n_items <= 0) return MOLANN_E_STAGE;
    const int c = check_io(x, out, n);
    if (c != MOLANN_OK || n == 0) return c;
    return launch_pre(p, x, n, out, 0, false, (hipStream_t)stream);
}

int molann_mlp_packed_f32(const molann_plan* cp, const float* f, int64_t n, float* out, molann_stream_t stream) {
    if (!cp) return MOLANN_E_NULL;
    molann_plan* p = const_cast<molann_plan*>(cp);
    if (p->n_layers <= 0) return MOLANN_E_STAGE;
    if (!p->mlp_packed) return MOLANN_E_NOT_PACKED;
    const int c = check_io(f, out, n);
    if (c != MOLANN_OK || n == 0) return c;
    const int e = launch_mlp(p, f, n, p->dims[0], out, (hipStream_t)stream);
    snprintf(p->last_info, sizeof(p->last_info), "%s", p->mlp_info);
    return e;
}

int molann_forward_packed_f32(const molann_plan* cp, const float* x, int64_t n, float* out, molann_stream_t stream) {
    if (!cp) return MOLANN_E_NULL;
    molann_plan* p = const_cast<molann_plan*>(cp);
    if (p->n_layers <= 0 || p->n_items <= 0) return MOLANN_E_STAGE;
    if (!p->mlp_packed) return MOLANN_E_NOT_PACKED;
    const int c = check_io(x, out, n);
    if (c != MOLANN_OK || n == 0) return c;
    if (p->fused_mlp) return launch_pre(p, x, n, out, 0, true, (hipStream_t)stream);
    // unfused: features of a chunk -> plan workspace (cache resident) -> MFMA MLP.  Two workspace halves:
    // the MLP of chunk i runs on the plan's side stream while this stream already gathers chunk i+1
    // (HBM-bound gather next to an MFMA/L2-bound kernel); events fork and join, so capture still works.
    char info[256];
    info[0] = 0;
    hipStream_t main = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(*p->launch_mu);
    if (p->have_done && p->last_stream != main) HIP_TRY(hipStreamWaitEvent(main, p->ev_done, 0)); // another stream used the workspace last
    int i = 0, rc = MOLANN_OK;
    bool mlp_recorded[2] = {false, false};
    if (n <= p->work_frames) {
        // one chunk: nothing to overlap, so both kernels go to the caller's stream back to back - handing the chunk to the side
        // stream and back costs two cross-stream waits (~15 us each on this stack: [6, 64, 64, 8] at 1 M frames 226 -> 196 us)
        rc = launch_pre(p, x, (long)n, p->d_work, 0, false, main);
        if (rc == MOLANN_OK) {
            snprintf(info, sizeof(info), "%s", p->last_info);
            rc = launch_mlp(p, p->d_work, (long)n, p->d_feat, out, main);
        }
        if (hipEventRecord(p->ev_done, main) == hipSuccess) { p->have_done = true; p->last_stream = main; }
        if (rc != MOLANN_OK) return rc;
        snprintf(p->last_info, sizeof(p->last_info), "%.130s || %.90s chunk=%ld", info, p->mlp_info, p->work_frames);
        return MOLANN_OK;
    }
    for (int64_t s = 0; s < n && rc == MOLANN_OK; s += p->work_frames, ++i) {
        const int h = i & 1;
        const long m = (long)std::min<int64_t>(p->work_frames, n - s);
        float* work = p->d_work + (size_t)h * p->work_frames * p->d_feat;
        if (i >= 2 && (rc = (int)hipStreamWaitEvent(main, p->ev_mlp[h], 0)) != 0) break; // this half is free again
        if ((rc = launch_pre(p, x + s * (long)p->n_inp * 3, m, work, 0, false, main)) != 0) break;
        if (s == 0) snprintf(info, sizeof(info), "%s", p->last_info);
        if ((rc = (int)hipEventRecord(p->ev_feat[h], main)) != 0) break;
        if ((rc = (int)hipStreamWaitEvent(p->side, p->ev_feat[h], 0)) != 0) break;
        rc = launch_mlp(p, work, m, p->d_feat, out + s * (long)p->out_dim, p->side);
        const int er = (int)hipEventRecord(p->ev_mlp[h], p->side); // also behind a failed launch: the join below needs it
        if (er == 0) mlp_recorded[h] = true;
        if (rc == 0) rc = er;
    }
    // join - on the error paths too: whatever reached the side stream is ordered before the caller's next work
    for (int h = 0; h < 2; ++h)
        if (mlp_recorded[h]) {
            const int er = (int)hipStreamWaitEvent(main, p->ev_mlp[h], 0);
            if (rc == 0) rc = er;
        }
    if (hipEventRecord(p->ev_done, main) == hipSuccess) { p->have_done = true; p->last_stream = main; }
    if (rc != MOLANN_OK) return rc;
    snprintf(p->last_info, sizeof(p->last_info), "%.130s || %.90s chunk=%ld", info, p->mlp_info, p->work_frames);
    return MOLANN_OK;
}

int molann_forward_f32(molann_plan* p, const float* x, int64_t n, const float* const* W, const float* const* b,
                       float* out, molann_stream_t stream) {
    const int e = molann_plan_update_mlp(p, W, b, stream);
    if (e != MOLANN_OK) return e;
    return molann_forward_packed_f32(p, x, n, out, stream);
}

// The fused forward that also keeps the features, for a backward that does not recompute them
int molann_forward_train_f32(molann_plan* p, const float* x, int64_t n, float* out, float* features, molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (p->n_layers <= 0 || p->n_items <= 0) return MOLANN_E_STAGE;
    if (!p->fused_mlp) return MOLANN_E_UNSUPPORTED;
    if (!p->mlp_packed) return MOLANN_E_NOT_PACKED;
    const int c = check_io(x, out, n);
    if (c != MOLANN_OK || n == 0) return c;
    if (!features) return MOLANN_E_NULL;
    if (((uintptr_t)features) & 3) return MOLANN_E_ALIGNMENT;
    return launch_pre(p, x, (long)n, out, 0, true, (hipStream_t)stream, features);
}


// ---- float64 entry points (the reference's modules follow x.dtype) -------------------------------------------------
static int check_io_f64(const void* x, const void* out, int64_t n) {
    if (n < 0) return MOLANN_E_DESC;
    if (n == 0) return MOLANN_OK;
    if (!x || !out) return MOLANN_E_NULL;
    if ((((uintptr_t)x) & 7) || (((uintptr_t)out) & 7)) return MOLANN_E_ALIGNMENT;
    return MOLANN_OK;
}

static int launch_f64(const molann_plan* p, const double* x, int64_t n, double* out, int mode, hipStream_t stream) {
    F64Args a;
    a.n_frames = n; a.n_inp = p->n_inp; a.n_align = p->n_align; a.n_items = p->n_items; a.out_cols = p->d_feat; a.mode = mode;
    const int grid = grid_for(p, n, 4, 8);
    hipLaunchKernelGGL(frames_f64_kernel, dim3(grid), dim3(256), 0, stream, x, out, p->d_align_idx, p->d_ref64, p->d_items, a);
    return (int)hipGetLastError();
}

int molann_align_f64(const molann_plan* p, const double* x, int64_t n, double* out_xyz, molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (p->n_align <= 0) return MOLANN_E_STAGE;
    const int c = check_io_f64(x, out_xyz, n);
    if (c != MOLANN_OK || n == 0) return c;
    snprintf(const_cast<molann_plan*>(p)->last_info, sizeof(p->last_info), "frames_f64_kernel (aligned coordinates)");
    return launch_f64(p, x, n, out_xyz, 1, (hipStream_t)stream);
}

int molann_features_f64(const molann_plan* p, const double* x, int64_t n, double* out, molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (p->n_items <= 0) return MOLANN_E_STAGE;
    const int c = check_io_f64(x, out, n);
    if (c != MOLANN_OK || n == 0) return c;
    snprintf(const_cast<molann_plan*>(p)->last_info, sizeof(p->last_info), "frames_f64_kernel (features)");
    return launch_f64(p, x, n, out, 0, (hipStream_t)stream);
}

// dL/dx of molann_features_f64 for the same x: grad_f[N, feature_dim] -> grad_x[N, n_inp, 3], everything in double
int molann_features_backward_f64(const molann_plan* p, const double* x, const double* grad_f, int64_t n, double* grad_x, molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (p->n_items <= 0) return MOLANN_E_STAGE;
    if (n < 0) return MOLANN_E_DESC;
    if (n == 0) return MOLANN_OK;
    if (!x || !grad_f || !grad_x) return MOLANN_E_NULL;
    if ((((uintptr_t)x) & 7) || (((uintptr_t)grad_f) & 7) || (((uintptr_t)grad_x) & 7)) return MOLANN_E_ALIGNMENT;
    F64Args a;
    a.n_frames = n; a.n_inp = p->n_inp; a.n_align = p->n_align; a.n_items = p->n_items; a.out_cols = p->d_feat; a.mode = 0;
    const int grid = grid_for(p, n, 4, 8);
    hipLaunchKernelGGL(frames_bwd_f64_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, grad_f, grad_x, p->d_align_idx, p->d_ref64, p->d_items, a);
    snprintf(const_cast<molann_plan*>(p)->last_info, sizeof(p->last_info), "frames_bwd_f64_kernel");
    return (int)hipGetLastError();
}

int molann_mlp_f64(const molann_plan* p, const double* f, int64_t n, const double* const* W, const double* const* b, double* out,
                   molann_stream_t stream) {
    if (!p || !W || !b) return MOLANN_E_NULL;
    if (p->n_layers <= 0) return MOLANN_E_STAGE;
    const int c = check_io_f64(f, out, n);
    if (c != MOLANN_OK || n == 0) return c;
    F64Mlp m;
    memset(&m, 0, sizeof(m));
    m.n_layers = p->n_layers; m.act = p->act;
    for (int i = 0; i <= p->n_layers; ++i) { m.dims[i] = p->dims[i]; m.max_w = std::max(m.max_w, p->dims[i]); }
    for (int l = 0; l < p->n_layers; ++l) {
        if (!W[l] || !b[l]) return MOLANN_E_NULL;
        m.W[l] = W[l]; m.b[l] = b[l];
    }
    const size_t lds = (size_t)4 * 2 * m.max_w * sizeof(double);
    if (lds > 65536) return MOLANN_E_UNSUPPORTED;
    const int grid = grid_for(p, n, 4, 8);
    hipLaunchKernelGGL(mlp_f64_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, f, out, (long)n, m);
    return (int)hipGetLastError();
}

int molann_forward_f64(const molann_plan* p, const double* x, int64_t n, const double* const* W, const double* const* b,
                       double* features_work, double* out, molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (p->n_layers <= 0 || p->n_items <= 0) return MOLANN_E_STAGE;
    if (n > 0 && !features_work) return MOLANN_E_NULL;
    int e = molann_features_f64(p, x, n, features_work, stream);
    if (e != MOLANN_OK || n == 0) return e;
    e = molann_mlp_f64(p, features_work, n, W, b, out, stream);
    snprintf(const_cast<molann_plan*>(p)->last_info, sizeof(p->last_info), "frames_f64_kernel (features) + mlp_f64_kernel");
    return e;
}

int molann_plan_grad_params_size(const molann_plan* p) { return p ? p->n_grad_params : MOLANN_E_NULL; }

int molann_plan_supports_backward(const molann_plan* p) {
    if (!p) return MOLANN_E_NULL;
    if (!p->geom[0].ok) return (p->n_items > 0 && p->n_layers == 0) ? 1 : 0; // large frames: features only (frames_wave_bwd_kernel)
    if (!p->spec || p->n_items <= 0 || p->bwd_state < 0 || !rtc_api()->ok) return 0;
    if (p->n_layers > 0 && (!p->fused_mlp || p->d_feat > LANE_MLP_MAX_WIDTH || p->mbwd_state < 0)) return 0;
    const int act = p->act;
    if (p->n_layers > 0 && !(act == 0 || act == 1 || act == 2 || act == 3 || act == 5 || act == 7)) return 0;
    return 1;
}

namespace {
// ---- backward: lazily built kernels and workspaces --------------------------------------------------------------------
// The workspaces (parameter partial sums, recomputed features) belong to the plan and are shared by all streams: the
// enqueue is serialised by launch_mu and a caller on another stream first waits for the event recorded behind the
// previous backward (the protocol of the unfused forward).
struct BwdGuard {
    molann_plan* p;
    hipStream_t s;
    std::unique_lock<std::mutex> lock;
    int rc;
    BwdGuard(molann_plan* plan, hipStream_t stream) : p(plan), s(stream), lock(*plan->launch_mu), rc(0) {
        if (p->have_bwork && p->bwork_stream != s) rc = (int)hipStreamWaitEvent(s, p->ev_bwork, 0);
    }
    ~BwdGuard() {
        if (hipEventRecord(p->ev_bwork, s) == hipSuccess) { p->have_bwork = true; p->bwork_stream = s; }
    }
};

int ensure_bwd_event(molann_plan* p) { // jit_mu held
    if (!p->ev_bwork) HIP_TRY(hipEventCreateWithFlags(&p->ev_bwork, hipEventDisableTiming));
    return MOLANN_OK;
}

int ensure_mlp_bwd(molann_plan* p) {
    if (p->mbwd_state == 0 || !p->d_gpart) {
        std::lock_guard<std::mutex> lock(*p->jit_mu);
        if (p->mbwd_state == 0) {
            const int rows = mlp_bwd_rows(p->spec->j.dims, p->act);
            const int wpb = (int)std::min<long>(8, (163840 - 64) / ((long)rows * 68 * 4));
            std::vector<char> code;
            std::string log;
            int rc = -1;
            if (wpb >= 1) rc = jit_compile(jit_source_mlp_bwd(*p->spec, wpb), code, log);
            if (rc == 0 && hipModuleLoadData(&p->mbwd_mod, code.data()) == hipSuccess &&
                hipModuleGetFunction(&p->mbwd_fn, p->mbwd_mod, "molann_mlp_bwd") == hipSuccess) {
                p->mbwd_wpb = wpb;
                p->mbwd_state = 1;
            } else {
                p->mbwd_state = -1;
                if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann mlp backward jit failed rc=%d\n%s\n", rc, log.c_str());
            }
        }
        if (p->mbwd_state == 1 && !p->d_gpart) {
            { const int er = ensure_bwd_event(p); if (er != MOLANN_OK) return er; }
            HIP_TRY(hipMalloc((void**)&p->d_gpart, (size_t)p->num_cus * std::max(1, p->n_grad_params) * 4));
        }
    }
    return p->mbwd_state == 1 ? MOLANN_OK : MOLANN_E_UNSUPPORTED;
}

int ensure_features_bwd(molann_plan* p, molann_plan::LaneGeom& g) {
    lane_geometry(g, 64 * p->n_inp * 12, 1); // the dense frame tile (reused for the gradient rows)
    if (!g.ok) return MOLANN_E_UNSUPPORTED;
    if (p->bwd_state == 0) {
        std::lock_guard<std::mutex> lock(*p->jit_mu);
        if (p->bwd_state == 0) {
            std::vector<char> code;
            std::string log;
            JitSpecBox b = *p->spec;
            b.j.wpb = g.wpb;
            const int rc = jit_compile(jit_source_bwd(b, g.lds_per_wave), code, log);
            if (rc == 0 && hipModuleLoadData(&p->bwd_mod, code.data()) == hipSuccess &&
                hipModuleGetFunction(&p->bwd_fn, p->bwd_mod, "molann_lane_bwd") == hipSuccess) {
                p->bwd_state = 1;
            } else {
                p->bwd_state = -1;
                if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann backward jit failed rc=%d\n%s\n", rc, log.c_str());
            }
        }
    }
    return p->bwd_state == 1 ? MOLANN_OK : MOLANN_E_UNSUPPORTED;
}

// the one-pass backward of the plan (molann_bwd_ring.inc): built at the first backward; a build that does not fit the LDS or
// needs scratch memory (register spills) leaves the two-kernel path in charge
int ensure_ring_bwd(molann_plan* p) {
    if (p->rbwd_state == 0 || (p->rbwd_state == 1 && p->n_grad_params > 0 && !p->d_gpart)) {
        std::lock_guard<std::mutex> lock(*p->jit_mu);
        if (p->rbwd_state == 0) {
            int st = -1;
            JitSpecBox b = *p->spec;
            const char* off = getenv("MOLANN_NO_RING_BWD");
            if (!(off && off[0] == '1') && p->geom[0].ok && bwd_ring_geometry(b.j, p->n_grad_params)) {
                // The kernel lives at the edge of its 256 registers.  Without SLP vectorisation first (C3: 2267 vector instructions and
                // no scratch, against 2564 + 16 spilled registers with it); the default for the plans that spill without it.
                const std::string src = jit_source_bwd_ring(b);
                for (int attempt = 0; attempt < 2 && st != 1; ++attempt) {
                    std::vector<char> code;
                    std::string log;
                    const int rc = jit_compile(src, code, log, attempt == 0 ? "-fno-slp-vectorize" : nullptr);
                    int scratch = -1;
                    hipModule_t mod = nullptr;
                    hipFunction_t fn = nullptr;
                    if (rc == 0 && hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, "molann_bwd_ring") == hipSuccess &&
                        hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn) == hipSuccess && scratch == 0) {
                        p->rbwd_mod = mod; p->rbwd_fn = fn;
                        p->rbwd_ncons = b.j.ncons; p->rbwd_nload = b.j.nload; p->rbwd_nslot = b.j.nslot; p->rbwd_lds = b.j.lds_block;
                        st = 1;
                    } else {
                        if (mod) (void)hipModuleUnload(mod);
                        if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann one-pass backward, build %d not used (rc=%d, scratch=%d)\n%s\n", attempt, rc, scratch, log.c_str());
                    }
                }
            }
            p->rbwd_state = st;
        }
        if (p->rbwd_state == 1 && p->n_grad_params > 0 && !p->d_gpart) {
            { const int er = ensure_bwd_event(p); if (er != MOLANN_OK) return er; }
            HIP_TRY(hipMalloc((void**)&p->d_gpart, (size_t)p->num_cus * p->n_grad_params * 4));
        }
    }
    return p->rbwd_state == 1 ? MOLANN_OK : MOLANN_E_UNSUPPORTED;
}

// molann_bwd_ring (+ reduce_rows_kernel); with parameter gradients the caller holds the workspace (BwdGuard)
int launch_ring_bwd(molann_plan* p, const float* x, const float* grad_out, long n, float* grad_x, float* grad_params, hipStream_t stream,
                    float* values = nullptr) {
    const long n_tiles = (n + 63) / 64;
    const int grid = (int)std::max<long>(1, std::min<long>(p->num_cus, n_tiles));
    const bool params = grad_params && p->n_grad_params > 0;
    struct { const float* x; const float* gout; const double* ref64; const float* ref32; const float* wnat; float* gx; float* gp; long n; float* y; } ka =
        {x, grad_out, p->d_ref64, p->d_ref, (const float*)p->d_wmfma, grad_x, params ? p->d_gpart : nullptr, n, values};
    size_t ksz = sizeof(ka);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksz, HIP_LAUNCH_PARAM_END};
    const int block = 64 * (p->rbwd_ncons + p->rbwd_nload);
    const hipError_t le = hipModuleLaunchKernel(values ? p->vjp_fn : p->rbwd_fn, grid, 1, 1, block, 1, 1, 0, stream, nullptr, cfg);
    if (le != hipSuccess) return (int)le;
    if (params) {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((p->n_grad_params + 63) / 64), dim3(1024), 0, stream, p->d_gpart, grid, p->n_grad_params,
                           grad_params);
        HIP_TRY(hipGetLastError());
    }
    snprintf(p->last_info, sizeof(p->last_info), "molann_bwd_ring%s (plan-specialised; %d consumer waves + %d loader, ring of %d tiles) grid=%d block=%d lds=%d%s",
             values ? "<values>" : "", p->rbwd_ncons, p->rbwd_nload, p->rbwd_nslot, grid, block, p->rbwd_lds, params ? " + reduce_rows_kernel" : "");
    return MOLANN_OK;
}

// molann_lane_bwd: grad_f -> grad_x (no workspace)
int launch_features_bwd(molann_plan* p, const molann_plan::LaneGeom& g, const float* x, const float* grad_f, long n, float* grad_x,
                        hipStream_t stream) {
    const long n_tiles = (n + 63) / 64;
    int bpc = (int)(163840 / ((long)g.wpb * g.lds_per_wave));
    if (bpc < 1) bpc = 1;
    if (bpc * g.wpb > 8) bpc = std::max(1, 8 / g.wpb);
    const int grid = grid_for(p, n_tiles, g.wpb, bpc);
    struct { const float* x; const float* gf; const double* ref64; float* gx; long n; int x_wide; } ka =
        {x, grad_f, p->d_ref64, grad_x, n, (((uintptr_t)x) & 15) == 0 ? 1 : 0};
    size_t ksz = sizeof(ka);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksz, HIP_LAUNCH_PARAM_END};
    const hipError_t le = hipModuleLaunchKernel(p->bwd_fn, grid, 1, 1, 64 * g.wpb, 1, 1, (unsigned)((size_t)g.wpb * g.lds_per_wave), stream,
                                                nullptr, cfg);
    snprintf(p->last_info, sizeof(p->last_info), "molann_lane_bwd (plan-specialised) grid=%d block=%d", grid, 64 * g.wpb);
    return (int)le;
}

// molann_mlp_bwd (+ reduce_rows_kernel when parameter gradients are wanted); the caller holds the workspace (BwdGuard)
int launch_mlp_bwd(molann_plan* p, const float* f, const float* grad_out, long n, float* grad_f, float* grad_params, hipStream_t stream) {
    const long n_tiles = (n + 63) / 64;
    const int grid = (int)std::max<long>(1, std::min<long>(p->num_cus, (n_tiles + p->mbwd_wpb - 1) / p->mbwd_wpb)); // one block per CU
    struct { const float* f; const float* gout; const float* wnat; float* gf; float* gp; long n; } ka =
        {f, grad_out, (const float*)p->d_wmfma, grad_f, grad_params ? p->d_gpart : nullptr, n};
    size_t ksz = sizeof(ka);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksz, HIP_LAUNCH_PARAM_END};
    const hipError_t le = hipModuleLaunchKernel(p->mbwd_fn, grid, 1, 1, 64 * p->mbwd_wpb, 1, 1, 0, stream, nullptr, cfg);
    if (le != hipSuccess) return (int)le;
    if (grad_params) {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((p->n_grad_params + 63) / 64), dim3(1024), 0, stream, p->d_gpart, grid, p->n_grad_params,
                           grad_params);
        HIP_TRY(hipGetLastError());
    }
    snprintf(p->last_info, sizeof(p->last_info), "molann_mlp_bwd (plan-specialised) grid=%d block=%d%s", grid, 64 * p->mbwd_wpb,
             grad_params ? " + reduce_rows_kernel" : "");
    return MOLANN_OK;
}
} // namespace

// dL/dx and dL/d(parameters) of molann_forward_packed_f32 / molann_features_f32 for the same x.
int molann_backward_f32(molann_plan* p, const float* x, const float* grad_out, int64_t n, float* grad_x, float* grad_params,
                        molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (n < 0) return MOLANN_E_DESC;
    if (n == 0) return MOLANN_OK;
    if (!x || !grad_out) return MOLANN_E_NULL;
    if (!p->geom[0].ok && p->n_items > 0 && p->n_layers == 0) { // large frames: one wave per frame, no parameters
        if (!grad_x) return MOLANN_OK;
        if ((((uintptr_t)x) & 3) || (((uintptr_t)grad_out) & 3) || (((uintptr_t)grad_x) & 3)) return MOLANN_E_ALIGNMENT;
        PreArgs a;
        fill_pre_args(p, a, n, 0, p->d_feat, false, x, grad_x);
        const int per_wave = ((p->n_items * 48 + 15) / 16) * 16;   // g_y of every item's four atoms
        if (p->bw_touched > 0 && per_wave <= 65536 && getenv("MOLANN_BWD_ATOMICS") == nullptr) {
            const int wpb = std::max(1, std::min(4, 65536 / per_wave));
            const int bpc = std::max(1, std::min(8, 160 * 1024 / (wpb * per_wave)));
            const int grid = grid_for(p, n, wpb, bpc);
            BwGatherArgs b = {p->bw_touched, per_wave};
            hipLaunchKernelGGL(frames_wave_bwd_gather_kernel, dim3(grid), dim3(64 * wpb), (size_t)wpb * per_wave, (hipStream_t)stream, x, grad_out,
                               grad_x, p->d_align_idx, p->d_ref, p->d_ref64, p->d_items, p->d_bw_atoms, p->d_bw_ptr, p->d_bw_list, p->d_bw_align,
                               a, b);
            snprintf(p->last_info, sizeof(p->last_info), "frames_wave_bwd_gather_kernel grid=%d block=%d lds=%d", grid, 64 * wpb, wpb * per_wave);
            return (int)hipGetLastError();
        }
        const int wpb = 4;
        const int grid = grid_for(p, n, wpb, 8);
        hipLaunchKernelGGL(frames_wave_bwd_kernel, dim3(grid), dim3(64 * wpb), 0, (hipStream_t)stream, x, grad_out, grad_x,
                           p->d_align_idx, p->d_ref, p->d_ref64, p->d_items, a);
        snprintf(p->last_info, sizeof(p->last_info), "frames_wave_bwd_kernel grid=%d block=%d", grid, 64 * wpb);
        return (int)hipGetLastError();
    }
    if (!p->spec || p->n_items <= 0) return MOLANN_E_UNSUPPORTED;
    if (p->n_layers == 0 || molann_plan_supports_backward(p)) { // one pass over x when the plan's kernel could be built
        if (p->n_layers > 0 && !p->mlp_packed) return MOLANN_E_NOT_PACKED;
        if ((((uintptr_t)x) & 3) || (((uintptr_t)grad_out) & 3) || (((uintptr_t)grad_x) & 3) || (((uintptr_t)grad_params) & 3)) return MOLANN_E_ALIGNMENT;
        if (!grad_x && !(grad_params && p->n_layers > 0)) return MOLANN_OK;
        const int er = ensure_ring_bwd(p);
        if (er == MOLANN_OK) {
            if (!(grad_params && p->n_grad_params > 0)) return launch_ring_bwd(p, x, grad_out, (long)n, grad_x, nullptr, (hipStream_t)stream);
            BwdGuard guard(p, (hipStream_t)stream);
            if (guard.rc != 0) return guard.rc;
            return launch_ring_bwd(p, x, grad_out, (long)n, grad_x, grad_params, (hipStream_t)stream);
        }
        if (er != MOLANN_E_UNSUPPORTED) return er;
    }
    if (p->n_layers == 0) return molann_features_backward_f32(p, x, grad_out, n, grad_x, stream);
    // plans with an MLP, nothing saved from the forward: features (recomputed) -> MLP backward -> preprocessing backward,
    // in chunks through the plan's backward workspace (allocated at the first call, like the kernels are compiled then)
    if (!molann_plan_supports_backward(p)) return MOLANN_E_UNSUPPORTED;
    if (!p->mlp_packed) return MOLANN_E_NOT_PACKED;
    if ((((uintptr_t)x) & 3) || (((uintptr_t)grad_out) & 3) || (((uintptr_t)grad_x) & 3) || (((uintptr_t)grad_params) & 3)) return MOLANN_E_ALIGNMENT;
    if (!grad_x && !grad_params) return MOLANN_OK;
    molann_plan::LaneGeom g;
    int rc = ensure_mlp_bwd(p);
    if (rc == MOLANN_OK && grad_x) rc = ensure_features_bwd(p, g);
    if (rc != MOLANN_OK) return rc;
    if (!p->d_bwork) {
        std::lock_guard<std::mutex> lock(*p->jit_mu);
        if (!p->d_bwork) {
            long bf = std::min<long>(1l << 20, (64l << 20) / ((long)p->d_feat * 4)) & ~63l;
            bf = std::max<long>(bf, 4096);
            float* w = nullptr;
            HIP_TRY(hipMalloc((void**)&w, 2 * (size_t)bf * p->d_feat * 4));
            p->bwork_frames = bf;
            p->d_bwork = w;
        }
    }
    hipStream_t main = (hipStream_t)stream;
    BwdGuard guard(p, main);
    if (guard.rc != 0) return guard.rc;
    float* wf = p->d_bwork;
    float* wg = p->d_bwork + (size_t)p->bwork_frames * p->d_feat;
    char info[3][96];
    info[0][0] = info[1][0] = info[2][0] = 0;
    for (int64_t s = 0; s < n && rc == MOLANN_OK; s += p->bwork_frames) {
        const long m = (long)std::min<int64_t>(p->bwork_frames, n - s);
        const float* xs = x + s * (long)p->n_inp * 3;
        if ((rc = launch_pre(p, xs, m, wf, 0, false, main)) != 0) break;
        if (s == 0) snprintf(info[0], sizeof(info[0]), "%.95s", p->last_info);
        if ((rc = launch_mlp_bwd(p, wf, grad_out + s * (long)p->out_dim, m, grad_x ? wg : nullptr, grad_params, main)) != 0) break;
        if (s == 0) snprintf(info[1], sizeof(info[1]), "%.95s", p->last_info);
        if (grad_x && (rc = launch_features_bwd(p, g, xs, wg, m, grad_x + s * (long)p->n_inp * 3, main)) != 0) break;
        if (s == 0 && grad_x) snprintf(info[2], sizeof(info[2]), "%.95s", p->last_info);
    }
    if (rc == MOLANN_OK) snprintf(p->last_info, sizeof(p->last_info), "%.80s || %.80s || %.80s", info[0], info[1], info[2]);
    return rc;
}

// The forward's outputs AND the vector-Jacobian product of a batch in ONE launch: the one-pass backward recomputes the forward per
// frame anyway, so a build of it that also stores the outputs (WITH_VALUES: one more product on the matrix cores, from the
// activations already in its scratch) returns both.  For callers that differentiate a small batch at every step with a cotangent
// they know up front - or want the Jacobian: a batch of d_out copies of a frame with the identity as cotangent (README.rst:49's
// use, a collective variable inside an MD engine).  Parameters are data here (no parameter gradients).  Plans the one-pass
// backward serves (molann_plan_backward_kind == 2); E_UNSUPPORTED otherwise.  The first call builds the kernel: outside a capture.
int molann_value_and_vjp_f32(molann_plan* p, const float* x, const float* grad_out, int64_t n, float* out, float* grad_x, molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (n < 0) return MOLANN_E_DESC;
    if (n == 0) return MOLANN_OK;
    if (!x || !grad_out || !out || !grad_x) return MOLANN_E_NULL;
    if ((((uintptr_t)x) & 3) || (((uintptr_t)grad_out) & 3) || (((uintptr_t)grad_x) & 3) || (((uintptr_t)out) & 3)) return MOLANN_E_ALIGNMENT;
    if (!p->spec || p->n_items <= 0 || !p->geom[0].ok) return MOLANN_E_UNSUPPORTED;
    if (p->n_layers > 0 && (!molann_plan_supports_backward(p) || !p->fused_mlp)) return MOLANN_E_UNSUPPORTED;
    if (p->n_layers > 0 && !p->mlp_packed) return MOLANN_E_NOT_PACKED;
    const int er = ensure_ring_bwd(p);       // the geometry is the one-pass backward's
    if (er != MOLANN_OK) return er;
    if (p->vjp_state == 0) {
        std::lock_guard<std::mutex> lock(*p->jit_mu);
        if (p->vjp_state == 0) {
            int st = -1;
            JitSpecBox b = *p->spec;
            if (bwd_ring_geometry(b.j, p->n_grad_params)) {
                b.j.with_values = true;
                const std::string src = jit_source_bwd_ring(b);
                for (int attempt = 0; attempt < 2 && st != 1; ++attempt) {   // (scratch is tolerated here: a latency path, not a throughput path)
                    std::vector<char> code;
                    std::string log;
                    hipModule_t mod = nullptr;
                    hipFunction_t fn = nullptr;
                    if (jit_compile(src, code, log, attempt == 0 ? "-fno-slp-vectorize" : nullptr) == 0 && hipModuleLoadData(&mod, code.data()) == hipSuccess &&
                        hipModuleGetFunction(&fn, mod, "molann_bwd_ring") == hipSuccess) {
                        p->vjp_mod = mod; p->vjp_fn = fn;
                        st = 1;
                    } else {
                        if (mod) (void)hipModuleUnload(mod);
                        if (getenv("MOLANN_JIT_VERBOSE")) fprintf(stderr, "molann value + vjp build %d failed\n%s\n", attempt, log.c_str());
                    }
                }
            }
            p->vjp_state = st;
        }
    }
    if (p->vjp_state != 1) return MOLANN_E_UNSUPPORTED;
    return launch_ring_bwd(p, x, grad_out, (long)n, grad_x, nullptr, (hipStream_t)stream, out);
}

// how molann_backward_f32 will serve this plan: 2 one pass over x (nothing worth saving from the forward), 1 two kernels
// (a caller that keeps the features of its forward saves their recompute), 0 not at all.  Builds the kernel it reports.
int molann_plan_backward_kind(molann_plan* p) {
    if (!p) return MOLANN_E_NULL;
    if (!molann_plan_supports_backward(p)) return 0;
    if (!p->geom[0].ok || !p->spec) return 1;
    return ensure_ring_bwd(p) == MOLANN_OK ? 2 : 1;
}

// dL/dx of molann_features_f32 for the same x: grad_f[N, feature_dim] -> grad_x[N, n_inp, 3]
int molann_features_backward_f32(molann_plan* p, const float* x, const float* grad_f, int64_t n, float* grad_x, molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (n < 0) return MOLANN_E_DESC;
    if (n == 0) return MOLANN_OK;
    if (!x || !grad_f) return MOLANN_E_NULL;
    if (!p->geom[0].ok) { // large frames: the wave-per-frame kernel (plans without an MLP)
        if (p->n_items <= 0 || p->n_layers != 0) return MOLANN_E_UNSUPPORTED;
        return molann_backward_f32(p, x, grad_f, n, grad_x, nullptr, stream);
    }
    if (!p->spec || p->n_items <= 0) return MOLANN_E_UNSUPPORTED;
    if (!grad_x) return MOLANN_OK;
    if ((((uintptr_t)x) & 3) || (((uintptr_t)grad_f) & 3) || (((uintptr_t)grad_x) & 3)) return MOLANN_E_ALIGNMENT;
    molann_plan::LaneGeom g;
    const int rc = ensure_features_bwd(p, g);
    if (rc != MOLANN_OK) return rc;
    return launch_features_bwd(p, g, x, grad_f, (long)n, grad_x, (hipStream_t)stream);
}

// dL/df and dL/d(parameters) of molann_mlp_packed_f32 for the same f (the fused family: every width <= 32)
int molann_mlp_backward_f32(molann_plan* p, const float* f, const float* grad_out, int64_t n, float* grad_f, float* grad_params,
                            molann_stream_t stream) {
    if (!p) return MOLANN_E_NULL;
    if (n < 0) return MOLANN_E_DESC;
    if (n == 0) return MOLANN_OK;
    if (!f || !grad_out) return MOLANN_E_NULL;
    if (p->n_layers <= 0) return MOLANN_E_STAGE;
    if (!p->spec || !molann_plan_supports_backward(p)) return MOLANN_E_UNSUPPORTED;
    if (!p->mlp_packed) return MOLANN_E_NOT_PACKED;
    if ((((uintptr_t)f) & 3) || (((uintptr_t)grad_out) & 3) || (((uintptr_t)grad_f) & 3) || (((uintptr_t)grad_params) & 3)) return MOLANN_E_ALIGNMENT;
    if (!grad_f && !grad_params) return MOLANN_OK;
    const int rc = ensure_mlp_bwd(p);
    if (rc != MOLANN_OK) return rc;
    if (!grad_params) return launch_mlp_bwd(p, f, grad_out, (long)n, grad_f, nullptr, (hipStream_t)stream); // no workspace involved
    BwdGuard guard(p, (hipStream_t)stream);
    if (guard.rc != 0) return guard.rc;
    return launch_mlp_bwd(p, f, grad_out, (long)n, grad_f, grad_params, (hipStream_t)stream);
}

// diagnostic / test hook: generate (and optionally compile, needs no GPU) the plan-specialised kernel
// source for a description.  Returns the source length, or a negative MOLANN_E_* / positive hiprtcResult.
int molann_debug_jit(const molann_plan_desc* d, int do_compile, char* buf, int cap) {
    const int v = validate_desc(d);
    if (v != MOLANN_OK) return v;
    if (do_compile & 4) { // the wide bf16 MLP kernel of the same plan (FB as plan creation would choose it)
        if (d->n_layers <= 0) return MOLANN_E_STAGE;
        ChainGeom cg;
        memset(&cg, 0, sizeof(cg));
        cg.nl = d->n_layers;
        cg.bf16 = d->mlp_precision == MOLANN_MLP_BF16 ? 1 : 0;
        for (int i = 0; i <= d->n_layers; ++i) cg.dims[i] = d->layer_dims[i];
        int fb = 0;
        for (int f = 4; f >= 1 && fb == 0; --f)
            if (f * cg.regs_per_fb() <= (chain_resident(cg) ? 256 : 400)) fb = f;
        if (fb == 0 || (!chain_resident(cg) && 2 * cg.slab_max() * 1024 > 163840 - 1024)) return MOLANN_E_UNSUPPORTED;
        const std::string csrc = jit_source_chain(cg, d->activation, fb);
        if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", csrc.c_str());
        if (do_compile & 1) {
            std::vector<char> code;
            std::string log;
            const int rc = jit_compile(csrc, code, log);
            if (rc != 0) {
                if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", log.c_str());
                return rc > 0 ? rc : MOLANN_E_UNSUPPORTED;
            }
        }
        return (int)csrc.size();
    }
    JitSpec j;
    std::vector<int> slot_of(d->n_inp, -1);
    auto slot = [&](int atom) {
        if (slot_of[atom] < 0) { slot_of[atom] = (int)j.slots.size(); j.slots.push_back(atom); }
        return slot_of[atom];
    };
    for (int i = 0; i < d->n_align; ++i)
        if (slot(d->align_idx[i]) != i) return MOLANN_E_UNSUPPORTED;
    int col = 0;
    for (int f = 0; f < d->n_features; ++f) {
        const int* idx = d->feat_idx + d->feat_ptr[f];
        const int cnt = d->feat_ptr[f + 1] - d->feat_ptr[f], t = d->feat_type[f];
        if (t == MOLANN_FEAT_POSITION) {
            for (int i = 0; i < cnt; ++i) { ItemDev it = {IT_POSITION, col, {slot(idx[i]), 0, 0, 0}, {0, 0}}; it.idx[1] = it.idx[2] = it.idx[3] = it.idx[0]; j.items.push_back(it); col += 3; }
        } else {
            ItemDev it;
            it.type = t == MOLANN_FEAT_ANGLE ? (d->use_angle_value ? IT_ANGLE_VAL : IT_ANGLE_COS)
                      : t == MOLANN_FEAT_BOND ? IT_BOND : (d->use_angle_value ? IT_DIHEDRAL_VAL : IT_DIHEDRAL_CS);
            it.col = col;
            for (int i = 0; i < 4; ++i) it.idx[i] = slot(idx[i < cnt ? i : 0]);
            it.pad[0] = it.pad[1] = 0;
            j.items.push_back(it);
            col += item_width(it.type);
        }
    }
    if (j.items.empty() || (int)j.items.size() > JIT_MAX_ITEMS || (int)j.slots.size() > JIT_MAX_SLOTS) return MOLANN_E_UNSUPPORTED;
    j.n_inp = d->n_inp; j.n_align = d->n_align; j.act = d->activation; j.d_feat = col;
    j.n_layers = d->n_layers; j.out_cols = d->n_layers > 0 ? d->layer_dims[d->n_layers] : col;
    if (d->n_layers > 0) j.dims.assign(d->layer_dims, d->layer_dims + d->n_layers + 1);
    molann_plan::LaneGeom g;
    j.win = compact_windows(j.slots, d->n_inp);
    jit_geometry(j, g, col, std::max(1, d->n_layers > 0 ? ceil_to(col, 4) : col));
    if (!g.ok || 3 * d->n_inp < 4) return MOLANN_E_UNSUPPORTED;
    j.waves_per_eu = 2;
    j.save_feat = (do_compile & 32) != 0 && j.n_layers > 0;   // the feature-keeping twin of the fused forward
    std::string src = jit_source(j);
    if (do_compile & 2) { // the backward kernel of the same plan
        JitSpecBox b;
        b.j = j;
        long off = 0;
        for (int l = 0; l < j.n_layers; ++l) {
            const int kp = ceil_to(j.dims[l], 16), jp = ceil_to(j.dims[l + 1], 16);
            b.kp.push_back(kp); b.jp.push_back(jp); b.woff.push_back(off);
            off += (long)jp * kp + jp;
            off = (off + 3) & ~3l;
        }
        if (do_compile & 16) { // ... in one pass (molann_bwd_ring.inc)
            if (j.n_layers > 0)
                for (int v : j.dims) if (v > 32) return MOLANN_E_UNSUPPORTED;
            long np = 0;
            for (int l = 0; l < j.n_layers; ++l) np += (long)j.dims[l + 1] * j.dims[l] + j.dims[l + 1];
            if (!bwd_ring_geometry(b.j, (int)np)) return MOLANN_E_UNSUPPORTED;
            b.j.with_values = (do_compile & 64) != 0;          // ... the build that also stores the forward's outputs (molann_value_and_vjp_f32)
            src = jit_source_bwd_ring(b);
        } else if (do_compile & 8) { // ... its MLP half (molann_mlp_bwd.inc)
            if (j.n_layers <= 0) return MOLANN_E_STAGE;
            for (int v : j.dims) if (v > 32) return MOLANN_E_UNSUPPORTED;
            const int rows = mlp_bwd_rows(j.dims, j.act);
            const int wpb = (int)std::min<long>(8, (163840 - 64) / ((long)rows * 68 * 4));
            if (wpb < 1) return MOLANN_E_UNSUPPORTED;
            src = jit_source_mlp_bwd(b, wpb);
        } else {             // ... its preprocessing half (molann_lane_bwd.inc)
            molann_plan::LaneGeom gb;
            lane_geometry(gb, 64 * d->n_inp * 12, 1);
            b.j.wpb = gb.wpb;
            src = jit_source_bwd(b, gb.lds_per_wave);
        }
    }
    if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", src.c_str());
    if (do_compile & 1) {
        std::vector<char> code;
        std::string log;
        const int rc = jit_compile(src, code, log, (do_compile & 2) ? nullptr : "-fno-slp-vectorize");
        if (rc != 0) {
            if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", log.c_str());
            return rc > 0 ? rc : MOLANN_E_UNSUPPORTED;
        }
    }
    return (int)src.size();
}

// diagnostic: read and clear the phase-stamp sums (16 x u64; [0..5] consumer phases, [6] clock ratio, [7] = number of
// consumer waves that reported, [8..10] loader: waiting for a free slot / issuing DMA / waiting for a tile to land,
// [11] loader waves, [12] tiles issued)
int molann_debug_read_stamps(unsigned long long* out16) {
    if (!out16) return MOLANN_E_NULL;
    unsigned long long zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero)));
    return MOLANN_OK;
}

// ---- self-test hooks: the same __host__ __device__ source, compiled for the host -------------
int molann_selftest_kabsch_rotation(const double* H9, double e0, float* R9) {
    if (!H9 || !R9) return MOLANN_E_NULL;
    double h[9];
    float r[9];
    for (int i = 0; i < 9; ++i) h[i] = H9[i];
    kabsch_rotation(h, e0, r);
    for (int i = 0; i < 9; ++i) R9[i] = r[i];
    return MOLANN_OK;
}

int molann_selftest_kabsch_rotation_f32(const float* H9, float e0, float* R9) {
    if (!H9 || !R9) return MOLANN_E_NULL;
    float h[9], r[9];
    for (int i = 0; i < 9; ++i) h[i] = H9[i];
    kabsch_rotation_f32(h, e0, r);
    for (int i = 0; i < 9; ++i) R9[i] = r[i];
    return MOLANN_OK;
}

int molann_selftest_feature(int type, int use_angle_value, const float* a, float* out3) {
    if (!a || !out3) return MOLANN_E_NULL;
    int it;
    if (type == MOLANN_FEAT_ANGLE) it = use_angle_value ? IT_ANGLE_VAL : IT_ANGLE_COS;
    else if (type == MOLANN_FEAT_BOND) it = IT_BOND;
    else if (type == MOLANN_FEAT_DIHEDRAL) it = use_angle_value ? IT_DIHEDRAL_VAL : IT_DIHEDRAL_CS;
    else if (type == MOLANN_FEAT_POSITION) it = IT_POSITION;
    else return MOLANN_E_FEATURE;
    float v[3] = {0.f, 0.f, 0.f};
    const int w = eval_item(it, v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), v);
    for (int i = 0; i < w; ++i) out3[i] = v[i];
    return w;
}

float molann_selftest_activation(int act, float v) { return apply_activation(act, v); }

int molann_selftest_feature_backward(int type, int use_angle_value, const float* a, const float* g3, float* ga12) {
    if (!a || !g3 || !ga12) return MOLANN_E_NULL;
    int it;
    if (type == MOLANN_FEAT_ANGLE) it = use_angle_value ? IT_ANGLE_VAL : IT_ANGLE_COS;
    else if (type == MOLANN_FEAT_BOND) it = IT_BOND;
    else if (type == MOLANN_FEAT_DIHEDRAL) it = use_angle_value ? IT_DIHEDRAL_VAL : IT_DIHEDRAL_CS;
    else if (type == MOLANN_FEAT_POSITION) it = IT_POSITION;
    else return MOLANN_E_FEATURE;
    V3 g[4] = {v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
    const float gg[3] = {g3[0], g3[1], g3[2]};
    eval_item_backward(it, v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), gg, g[0],
                       g[1], g[2], g[3]);
    for (int i = 0; i < 4; ++i) { ga12[3 * i] = g[i].x; ga12[3 * i + 1] = g[i].y; ga12[3 * i + 2] = g[i].z; }
    return MOLANN_OK;
}

int molann_selftest_kabsch_backward(const double* H9, const float* R9, const float* GR9, float* GH9) {
    if (!H9 || !R9 || !GR9 || !GH9) return MOLANN_E_NULL;
    double h[9];
    float r[9], gr[9], gh[9];
    for (int i = 0; i < 9; ++i) { h[i] = H9[i]; r[i] = R9[i]; gr[i] = GR9[i]; }
    kabsch_rotation_backward(h, r, gr, gh);
    for (int i = 0; i < 9; ++i) GH9[i] = gh[i];
    return MOLANN_OK;
}

float molann_selftest_act_derivative(int act, float z) { return act_derivative(act, z, apply_activation(act, z)); }

} // extern "C"
