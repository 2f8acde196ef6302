/*
 * molann_hip.h - C ABI of the MI355X (gfx950) implementation of molann's per-frame forward path
 *
 *     x[N, n_inp, 3]  ->  AlignmentLayer (Kabsch)  ->  FeatureLayer  ->  MLP  ->  y[N, d_out]
 *
 * The reference (zwpku/molann v1.1.7) is pure Python: it has no FFI for this path.  The interface a
 * replacement sits behind is the forward of its torch.nn.Modules (molann/ann.py); each entry point
 * below names the reference method it replaces.  A host binding (ctypes, see INTEGRATION.md, and
 * molann_amd/_capi.py) builds one immutable plan per module and then calls the launch functions with
 * raw device pointers.
 *
 * Conventions
 *   - return value: 0 = ok; negative = MOLANN_E_* (bad argument / unsupported); positive = hipError_t.
 *   - every `x`, `out`, `W[i]`, `b[i]`, `ref_x` passed to a LAUNCH function is a DEVICE pointer owned
 *     by the caller; pointers inside molann_plan_desc are HOST pointers, read during plan_create only.
 *   - launch functions only enqueue work on `stream`: no host synchronisation, no allocation
 *     (graph-capturable).  They are thread-safe for distinct plans; one plan may be used from several
 *     streams as long as molann_plan_update_* calls are ordered before the launches that need them.
 *     A plan whose MLP is not fused into the frame kernel (wide MLPs / large frames) owns a feature
 *     workspace, a side stream and events: its forward calls are serialised by the library itself (a
 *     host mutex around the enqueue; a caller on another stream first waits for the event recorded
 *     behind the previous call), so they are safe from any stream or thread but do not overlap.
 *     The same protocol covers the backward's plan-owned workspaces (per-block parameter sums; the recomputed
 *     features of the three-launch path).  The FIRST call of an entry point whose kernel is built lazily (the
 *     backward kernels, molann_forward_train_f32, molann_features_f32 on a plan with a fused MLP, molann_align_f32
 *     on small frames) compiles it with hipRTC and may allocate its workspace: make that call outside a graph
 *     capture (molann_plan_backward_kind builds the backward ahead of time).
 *   - x is [n_frames, n_inp, 3] fp32, contiguous, frame-major / atom-major / xyz-minor (the layout of
 *     the tensor the reference's forward receives, ann.py:170).  Any 4-byte aligned pointer works;
 *     16-byte aligned pointers take the wide-load path.
 *   - n_frames == 0 is legal and does nothing (the reference returns an empty tensor).
 *   - code object: gfx950 only.
 */
#ifndef MOLANN_HIP_H
#define MOLANN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOLANN_ABI_VERSION 1

/* feature type ids: molann/feature.py:87-97 */
#define MOLANN_FEAT_ANGLE 0
#define MOLANN_FEAT_BOND 1
#define MOLANN_FEAT_DIHEDRAL 2
#define MOLANN_FEAT_POSITION 3

/* activation between the Linear layers of create_sequential_nn (ann.py:37,64) */
#define MOLANN_ACT_TANH 0
#define MOLANN_ACT_RELU 1
#define MOLANN_ACT_SIGMOID 2
#define MOLANN_ACT_IDENTITY 3
#define MOLANN_ACT_ELU 4
#define MOLANN_ACT_SILU 5
#define MOLANN_ACT_SOFTPLUS 6
#define MOLANN_ACT_LEAKY_RELU 7 /* slope 0.01 */
#define MOLANN_ACT_GELU 8       /* erf form (torch.nn.GELU default) */

/* MLP arithmetic */
#define MOLANN_MLP_F32 0  /* fp32 weights, fp32 FMA / fp32-input MFMA (exact fp32) */
#define MOLANN_MLP_BF16 1 /* bf16 weights + activations, fp32 accumulate on bf16 MFMA */

#define MOLANN_MAX_LAYERS 16

/* error codes (negative) */
#define MOLANN_OK 0
#define MOLANN_E_NULL (-1)        /* a required pointer is NULL */
#define MOLANN_E_DESC (-2)        /* inconsistent plan description (sizes, counts) */
#define MOLANN_E_INDEX (-3)       /* an atom index is outside [0, n_inp) */
#define MOLANN_E_FEATURE (-4)     /* unknown feature type or wrong atom count for its type */
#define MOLANN_E_STAGE (-5)       /* the plan lacks the stage this call needs (no align / features / MLP) */
#define MOLANN_E_ALIGNMENT (-6)   /* pointer not 4-byte aligned */
#define MOLANN_E_UNSUPPORTED (-7) /* shape outside what the kernels cover */
#define MOLANN_E_NOT_PACKED (-8)  /* forward_packed before any plan_update_mlp */
#define MOLANN_E_DEVICE (-9)      /* no gfx950 device */

typedef struct molann_plan molann_plan;         /* opaque; owns a small device blob */
typedef struct ihipStream_t* molann_stream_t;   /* == hipStream_t */

/*
 * Everything the modules fix at construction time.
 *   AlignmentLayer.__init__  ann.py:123-146  -> n_align, align_idx (= _local_align_atom_indices), ref_x
 *   FeatureMap.__init__      ann.py:244-263  -> feat_type / feat_idx (= _local_atom_indices), use_angle_value
 *   FeatureLayer.__init__    ann.py:418-427  -> the list order = output column order (ann.py:473)
 *   create_sequential_nn     ann.py:37-67    -> layer_dims, activation
 * A stage that is absent has its count set to 0.
 */
typedef struct molann_plan_desc {
    int32_t abi_version;       /* MOLANN_ABI_VERSION */
    int32_t n_inp;             /* atoms per frame, ann.py:133 */

    int32_t n_align;           /* 0: no AlignmentLayer (PreprocessingANN uses Identity, ann.py:542) */
    const int32_t* align_idx;  /* [n_align] positions inside the n_inp axis, ann.py:144 */
    const float* ref_x;        /* [n_align*3] reference coordinates ALREADY centred, ann.py:140-141 */

    int32_t n_features;        /* 0: no FeatureLayer */
    const int32_t* feat_type;  /* [n_features] MOLANN_FEAT_* */
    const int32_t* feat_ptr;   /* [n_features+1] offsets into feat_idx */
    const int32_t* feat_idx;   /* positions inside the n_inp axis, ann.py:261, in the order given */
    int32_t use_angle_value;   /* ann.py:253 */

    int32_t n_layers;          /* number of Linear layers; 0: no MLP */
    const int32_t* layer_dims; /* [n_layers+1]; layer_dims[0] must equal the feature dimension */
    int32_t activation;        /* MOLANN_ACT_* */
    int32_t mlp_precision;     /* MOLANN_MLP_* */
} molann_plan_desc;

/* -- plan ------------------------------------------------------------------------------------- */

/* Validates the description (the reference raises ValueError / AssertionError for the same
 * conditions at module construction, ann.py:146,263,423 and feature.py:88-94), copies it to the
 * current device and returns the plan.  Allocates; call once per module. */
int molann_plan_create(const molann_plan_desc* desc, molann_plan** out_plan);
int molann_plan_destroy(molann_plan* plan);

/* FeatureLayer.output_dimension() ann.py:446-452 (0 when the plan has no features). */
int molann_plan_feature_dim(const molann_plan* plan);
/* Width of what molann_forward_* writes: last layer_dims entry, or the feature dim without an MLP. */
int molann_plan_out_dim(const molann_plan* plan);
/* Which kernel family serves the frames of this plan: 0 = lane-per-frame (LDS-staged small frames),
 * 1 = wave-per-frame (gathered large frames). */
int molann_plan_kernel_family(const molann_plan* plan);

/* The module's `ref_x` buffer is live state (register_buffer, ann.py:137: load_state_dict / .to()
 * can replace it).  Re-reads it from DEVICE memory [n_align*3], already centred. */
int molann_plan_update_ref(molann_plan* plan, const float* ref_x, molann_stream_t stream);

/* The Linear parameters are live (trainable).  Re-reads W[i] ([dims[i+1], dims[i]] row-major,
 * torch.nn.Linear layout) and b[i] ([dims[i+1]]) from DEVICE memory into the plan's packed copy.
 * W and b themselves are HOST arrays of n_layers device pointers. */
int molann_plan_update_mlp(molann_plan* plan, const float* const* W, const float* const* b,
                           molann_stream_t stream);

/* -- launches --------------------------------------------------------------------------------- */

/* AlignmentLayer.forward ann.py:157-199: out_xyz[N, n_inp, 3] = (x - c(x)) . R(x). */
int molann_align_f32(const molann_plan* plan, const float* x, int64_t n_frames, float* out_xyz,
                     molann_stream_t stream);

/* PreprocessingANN.forward ann.py:553-565 (= FeatureLayer.forward ann.py:454-474 when the plan has
 * no alignment): out[N, feature_dim]. */
int molann_features_f32(const molann_plan* plan, const float* x, int64_t n_frames, float* out,
                        molann_stream_t stream);

/* MolANN.forward ann.py:620-624 with the packed copy of the weights: out[N, out_dim]. */
int molann_forward_packed_f32(const molann_plan* plan, const float* x, int64_t n_frames, float* out,
                              molann_stream_t stream);

/* MolANN.forward reading the live parameters: molann_plan_update_mlp + molann_forward_packed_f32. */
int molann_forward_f32(molann_plan* plan, const float* x, int64_t n_frames, const float* const* W,
                       const float* const* b, float* out, molann_stream_t stream);

/* ann_layers alone on precomputed features f[N, layer_dims[0]] (create_sequential_nn's Sequential,
 * ann.py:60-65): out[N, out_dim].  Uses the packed weights. */
int molann_mlp_packed_f32(const molann_plan* plan, const float* f, int64_t n_frames, float* out,
                          molann_stream_t stream);

/* -- float64 ---------------------------------------------------------------------------------- */
/* The reference's modules follow x.dtype: after `model.double()` the same forward runs in float64 (ann.py:187-197,
 * 323-354).  These entry points are that mode: x, out, W[i], b[i], ref_x are DEVICE pointers to
 * doubles (8-byte aligned), the plan is the same one (index tables do not depend on the dtype).  Everything is
 * computed in double; nothing is packed: the Linear parameters are read from the caller's tensors at every call.
 * Written for agreement with the reference's float64 run to rounding, not for speed (one wave per frame). */

/* The `ref_x` buffer of a `.double()` AlignmentLayer: DEVICE [n_align*3] doubles, already centred. */
int molann_plan_update_ref_f64(molann_plan* plan, const double* ref_x, molann_stream_t stream);
/* AlignmentLayer.forward ann.py:157-199 in float64. */
int molann_align_f64(const molann_plan* plan, const double* x, int64_t n_frames, double* out_xyz, molann_stream_t stream);
/* PreprocessingANN.forward / FeatureLayer.forward ann.py:454-474, 553-565 in float64: out[N, feature_dim]. */
int molann_features_f64(const molann_plan* plan, const double* x, int64_t n_frames, double* out, molann_stream_t stream);
/* dL/dx of molann_features_f64 for the same x (the reference differentiates its float64 forward with autograd too):
 * grad_f[N, feature_dim] -> grad_x[N, n_inp, 3], doubles, any frame size.  The MLP of a float64 model is differentiated by
 * the caller (torch autograd over its own Linear modules). */
int molann_features_backward_f64(const molann_plan* plan, const double* x, const double* grad_f, int64_t n_frames, double* grad_x,
                                 molann_stream_t stream);

/* ann_layers ann.py:60-65 in float64 on features f[N, layer_dims[0]]: W, b HOST arrays of n_layers device pointers. */
int molann_mlp_f64(const molann_plan* plan, const double* f, int64_t n_frames, const double* const* W, const double* const* b,
                   double* out, molann_stream_t stream);
/* MolANN.forward ann.py:620-624 in float64; features_work is a caller-owned DEVICE buffer of n_frames * feature_dim
 * doubles (launch functions do not allocate). */
int molann_forward_f64(const molann_plan* plan, const double* x, int64_t n_frames, const double* const* W, const double* const* b,
                       double* features_work, double* out, molann_stream_t stream);

/* -- backward (the reference relies on torch autograd, incl. through its SVD) -------- */

/* Floats of the parameter-gradient buffer: for every Linear layer dW[J][K] (torch layout) then db[J]. */
int molann_plan_grad_params_size(const molann_plan* plan);

/* 1 if molann_backward_f32 can serve this plan (see below), else 0. */
int molann_plan_supports_backward(const molann_plan* plan);

/* How molann_backward_f32 serves this plan: 2 = one pass over x (recompute + MLP on the matrix cores + reverse mode in one
 * kernel: nothing worth saving from the forward), 1 = two or three launches (a caller that keeps the features of its
 * forward, molann_forward_train_f32, saves their recompute), 0 = not at all.  Builds the kernel it reports (hipRTC). */
int molann_plan_backward_kind(molann_plan* plan);

/* Gradients of molann_forward_packed_f32 (plans with an MLP) / molann_features_f32 (plans without) for the
 * same x: grad_out[N, out_dim] -> grad_x[N, n_inp, 3] (written; zeros for atoms the plan does not touch;
 * may be NULL) and grad_params (ACCUMULATED into with float atomics, so zero it first; may be NULL).
 * Nothing is saved from the forward.  One launch (molann_plan_backward_kind 2: loaders stream x through a ring in LDS,
 * consumers recompute the forward per frame, run the MLP's backward on the matrix cores and the analytic reverse mode
 * of the preprocessing) plus the reduction of the per-block parameter sums; where that kernel cannot be built, plans
 * with an MLP run three launches per chunk of frames - the features (recomputed into a workspace the plan allocates at
 * its first backward), molann_mlp_backward_f32 and molann_features_backward_f32 - and a caller that kept the features
 * of its forward (molann_forward_train_f32) calls those two itself and saves the recompute.
 * Available for plans served by the lane-per-frame kernel with the MLP fused (or no MLP) and tanh / ReLU /
 * sigmoid / identity / SiLU / LeakyReLU (kernels compiled with hipRTC at the first call); for feature
 * plans without an MLP on large frames (one wave per frame; frames up to 1024 atoms with an alignment: eight / four / two
 * frames per wave and round); and for large-frame plans whose MLP is within the fused MLP's limits (every width and the
 * feature dim <= 32, <= 4 layers, the activations above): the three launches, the features recomputed by the plan's
 * forward kernel; otherwise MOLANN_E_UNSUPPORTED. */
int molann_backward_f32(molann_plan* plan, const float* x, const float* grad_out, int64_t n_frames, float* grad_x,
                        float* grad_params, molann_stream_t stream);

/* out[N, out_dim] = molann_forward_packed_f32(x) (molann_features_f32(x) for a plan without an MLP; MolANN.forward ann.py:620-624)
 * AND grad_x[N, n_inp, 3] = the vector-Jacobian product for the cotangent grad_out[N, out_dim], in ONE launch: the one-pass
 * backward recomputes the forward per frame anyway; this build of it also stores the outputs.  For a host that differentiates a
 * small batch at every step (a collective variable and its forces inside an MD engine, README.rst:49): one launch instead of a
 * forward and a backward.  The Jacobian of one frame: a batch of out_dim copies of it with the identity as grad_out.  Parameters
 * are data (no parameter gradients).  Plans with molann_plan_backward_kind == 2, else MOLANN_E_UNSUPPORTED.  First call: hipRTC. */
int molann_value_and_vjp_f32(molann_plan* plan, const float* x, const float* grad_out, int64_t n_frames, float* out,
                             float* grad_x, molann_stream_t stream);

/* molann_forward_packed_f32 that also writes features[N, feature_dim] (what molann_features_f32 would give), for a
 * backward through molann_mlp_backward_f32 + molann_features_backward_f32 without the recompute.  Plans whose MLP
 * is fused into the lane kernel, and large-frame plans with a head within the fused MLP's limits (the features are written
 * where the caller keeps them and the head reads them there) - the ones molann_plan_supports_backward accepts with an MLP;
 * same `out` bit for bit. */
int molann_forward_train_f32(molann_plan* plan, const float* x, int64_t n_frames, float* out, float* features,
                             molann_stream_t stream);

/* dL/dx of molann_features_f32 (PreprocessingANN.forward ann.py:553-565) for the same x:
 * grad_f[N, feature_dim] -> grad_x[N, n_inp, 3].  On a plan with an MLP this is the preprocessing half of its
 * backward. */
int molann_features_backward_f32(molann_plan* plan, const float* x, const float* grad_f, int64_t n_frames,
                                 float* grad_x, molann_stream_t stream);

/* Backward of molann_mlp_packed_f32 (create_sequential_nn's Sequential, ann.py:60-65) for the same f[N, layer_dims[0]]:
 * grad_out[N, out_dim] -> grad_f[N, layer_dims[0]] (written; may be NULL) and grad_params (accumulated; may be NULL;
 * layout of molann_plan_grad_params_size).  fp32 matrix cores; plans for which molann_plan_supports_backward is 1. */
int molann_mlp_backward_f32(molann_plan* plan, const float* f, const float* grad_out, int64_t n_frames, float* grad_f,
                            float* grad_params, molann_stream_t stream);

/* -- misc ------------------------------------------------------------------------------------- */
int molann_abi_version(void);
/* "release" (the product library: reads no switch that changes or skips part of the computation) or
 * "diagnostics" (libmolann_hip_diag.so, `make diag`: additionally honours MOLANN_DEBUG_*,
 * MOLANN_ELIDE_INVARIANT_ALIGNMENT, MOLANN_JIT_EXTRA_FLAGS for tools/; never used for a reported number). */
const char* molann_build_kind(void);
const char* molann_error_string(int code);
/* Name + launch geometry of the kernels the last launch on this plan used (for bench / profiles).
 * Writes a NUL-terminated string of at most `cap` bytes; returns its length. */
int molann_plan_last_launch_info(const molann_plan* plan, char* buf, int cap);

/* Diagnostic / test hook: the source of the plan-specialised lane kernel for a description (copied to
 * buf, NUL-terminated, at most cap bytes) and, if do_compile != 0, a hipRTC compile of it for gfx950 (no
 * GPU needed).  do_compile bit 0: compile; bit 1: the backward kernel instead of the forward one.
 * Returns the source length; on a compile failure a positive hiprtcResult and the log in buf. */
int molann_debug_jit(const molann_plan_desc* desc, int do_compile, char* buf, int cap);

/* Diagnostic (diagnostics build only): per-phase shader-clock sums recorded when MOLANN_DEBUG_ABLATE has bit 32
 * set (see tools/stamps.py); reads and clears 16 counters.  Synchronises the device: never on a product path. */
int molann_debug_read_stamps(unsigned long long* out16);

/* Self-test hooks: the __host__ __device__ math the kernels are built from, compiled for the HOST, so
 * the CPU test-suite can check it against the oracle without a GPU.  Not a product path. */
int molann_selftest_kabsch_rotation(const double* H9, double e0, float* R9);
/* the fp32 instantiation of the same solver (plans whose items are all bond / angle / dihedral) */
int molann_selftest_kabsch_rotation_f32(const float* H9, float e0, float* R9);
int molann_selftest_feature(int type, int use_angle_value, const float* atoms_xyz, float* out3);
float molann_selftest_activation(int act, float v);
int molann_selftest_feature_backward(int type, int use_angle_value, const float* atoms_xyz, const float* g3, float* ga12);
int molann_selftest_kabsch_backward(const double* H9, const float* R9, const float* GR9, float* GH9);
float molann_selftest_act_derivative(int act, float z);

#ifdef __cplusplus
}
#endif
#endif /* MOLANN_HIP_H */
