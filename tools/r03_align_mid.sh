#!/bin/bash
# Run ON THE GPU BOX: frames_align_regs_kernel over mid-size frames, with the solver's parts switched off (diagnostics library)
cd "$(dirname "$0")/.."
export MOLANN_DIAG_LIB=1 ONLY_REGS=1
for cfg in "X=0" "MOLANN_DEBUG_ALIGN_FLAGS=1" "MOLANN_DEBUG_ALIGN_FLAGS=3" "MOLANN_WAVE_BPC=4" "MOLANN_WAVE_BPC=2"; do
  echo "[$cfg]"
  env $cfg timeout -k 10 300 python tools/time_align_sizes.py 500 700 900 1000 1100 1300 1536 1600 2>&1 | grep -v amdgpu.ids
done
