#!/bin/bash
# Run ON THE GPU BOX: A4 (AlignmentLayer.forward, 5000 atoms) variants of frames_align_regs_kernel (diagnostics library)
cd "$(dirname "$0")/.."
export MOLANN_DIAG_LIB=1
for rep in 1 2; do
for cfg in "MOLANN_WAVE_BPC=2" "MOLANN_DEBUG_ALIGN_OCC=4 MOLANN_WAVE_BPC=2" "MOLANN_WAVE_BPC=2 MOLANN_DEBUG_ALIGN_FLAGS=8" "MOLANN_DEBUG_ALIGN_OCC=4 MOLANN_WAVE_BPC=2 MOLANN_DEBUG_ALIGN_FLAGS=8" "MOLANN_DEBUG_ALIGN_OCC=4 MOLANN_WAVE_BPC=3 MOLANN_DEBUG_ALIGN_FLAGS=8" "MOLANN_DEBUG_ALIGN_OCC=4 MOLANN_WAVE_BPC=2 MOLANN_DEBUG_ALIGN_FLAGS=11" "MOLANN_DEBUG_ALIGN_OCC=4 MOLANN_WAVE_BPC=2 MOLANN_DEBUG_ALIGN_FLAGS=3"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 200 python bench.py --workload A4 --diagnostic --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3f ms  %.0f GB/s' % (d['ms_per_step'], d['roofline']['achieved']), d['config']['kernels'][100:130])"
done; done
