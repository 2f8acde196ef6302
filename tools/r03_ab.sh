#!/bin/bash
# Run ON THE GPU BOX (round 3): C3 tile-group / DMA-policy experiments (diagnostics library) and the ring kernel's nt gathers.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== C3, 1M frames per launch"
REPS=2 tools/ab_flags.sh "X=0" "MOLANN_DEBUG_TILE_GROUP=2" "MOLANN_DEBUG_TILE_GROUP=4" "MOLANN_DEBUG_TILE_GROUP=8" "MOLANN_DEBUG_DMA_AUX=2" "MOLANN_DEBUG_DMA_AUX=19" "MOLANN_DEBUG_TILE_GROUP=4 MOLANN_DEBUG_DMA_AUX=2"
echo "== C3, 8M frames per launch"
REPS=2 FRAMES=8388608 BUFFERS=2 STEPS=30 tools/ab_flags.sh "X=0" "MOLANN_DEBUG_TILE_GROUP=2" "MOLANN_DEBUG_TILE_GROUP=4" "MOLANN_DEBUG_TILE_GROUP=8" "MOLANN_DEBUG_DMA_AUX=2" "MOLANN_DEBUG_TILE_GROUP=4 MOLANN_DEBUG_DMA_AUX=2"
echo "== C2, 1M frames per launch"
WL=C2 REPS=2 tools/ab_flags.sh "X=0" "MOLANN_DEBUG_TILE_GROUP=4" "MOLANN_DEBUG_DMA_AUX=2"
} > gpurun_out/r03_ab_c3.txt 2>&1
{
for wl in C4 C5; do
  for nt in 0 1 0 1; do
    echo -n "$wl MOLANN_RING_NT=$nt: "
    MOLANN_RING_NT=$nt timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3f ms/step' % d['ms_per_step'], d['config']['kernels'][:60])"
  done
done
} > gpurun_out/r03_ab_ring_nt.txt 2>&1
