#!/bin/bash
# Run ON THE GPU BOX (round 3): C3 output rows stored as bursts of G tiles by one wave (STORE_GROUP = TILE_GROUP = G), diagnostics library
cd "$(dirname "$0")/.."
{
echo "== parity of the grouped build (golden molann_C3, G = 2 and 4)"
for g in 2 4; do MOLANN_DIAG_LIB=1 MOLANN_DEBUG_STORE_GROUP=$g MOLANN_DEBUG_TILE_GROUP=$g timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "golden_case and molann_C3 or full_size" 2>&1 | tail -1; done
echo "== C3, 1M frames per launch"
REPS=2 tools/ab_flags.sh "X=0" "MOLANN_DEBUG_TILE_GROUP=2 MOLANN_DEBUG_STORE_GROUP=2" "MOLANN_DEBUG_TILE_GROUP=4 MOLANN_DEBUG_STORE_GROUP=4" "MOLANN_DEBUG_TILE_GROUP=2"
echo "== C3, 8M frames per launch"
REPS=2 FRAMES=8388608 BUFFERS=2 STEPS=30 tools/ab_flags.sh "X=0" "MOLANN_DEBUG_TILE_GROUP=2 MOLANN_DEBUG_STORE_GROUP=2" "MOLANN_DEBUG_TILE_GROUP=4 MOLANN_DEBUG_STORE_GROUP=4"
} > gpurun_out/r03_ab_c3_store_group.txt 2>&1
cat gpurun_out/r03_ab_c3_store_group.txt
