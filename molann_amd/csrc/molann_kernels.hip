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
#include <sys/stat.h>
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

// One translation unit, split by concern.  Device code:
#include "molann_dev_common.inc"
#include "molann_dev_lane.inc"
#include "molann_dev_wave.inc"
#include "molann_dev_f64.inc"
#include "molann_dev_bwd.inc"
#include "molann_dev_mlp.inc"
// Host code (what `make san` instruments: -fno-gpu-sanitize leaves the device code alone):
#include "molann_host_plan.inc"
#include "molann_host_jit.inc"
#include "molann_host_launch.inc"
#include "molann_capi.inc"
