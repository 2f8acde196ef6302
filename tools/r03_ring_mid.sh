#!/bin/bash
# Run ON THE GPU BOX: frames_ring_kernel<1,8> on P1 - slots, consumers, loaders, depth
cd "$(dirname "$0")/.."
export MOLANN_DIAG_LIB=1
for cfg in "MOLANN_DEBUG_RING=16,6,6,0" "MOLANN_DEBUG_RING=16,5,5,1" "MOLANN_DEBUG_RING=16,4,6,1" "MOLANN_DEBUG_RING=16,8,8,0" "MOLANN_DEBUG_RING=16,6,5,1" "MOLANN_DEBUG_RING=16,7,4,1" "MOLANN_DEBUG_RING=16,8,4,1" "MOLANN_DEBUG_RING=16,10,6,0" "MOLANN_DEBUG_RING=16,8,6,0" "MOLANN_DEBUG_RING=16,6,3,2" "MOLANN_DEBUG_RING=16,8,4,1 MOLANN_RING_BATCH=4" "MOLANN_DEBUG_RING=16,8,8,0 MOLANN_RING_BATCH=4"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 200 python tools/prof_one.py P1 2>&1 | tail -1
done
