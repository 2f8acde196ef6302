#!/bin/bash
# Run ON THE GPU BOX: ring geometry (slots, consumers, loaders, depth): P1 (B = 8) and C4 / C5 (B = 1)
cd "$(dirname "$0")/.."
export MOLANN_DIAG_LIB=1
for cfg in "MOLANN_DEBUG_RING=16,3,9,0" "MOLANN_DEBUG_RING=16,4,8,0" "MOLANN_DEBUG_RING=16,5,7,0" "MOLANN_DEBUG_RING=16,2,10,0" "MOLANN_DEBUG_RING=16,6,6,0"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 200 python tools/prof_one.py P1 2>&1 | tail -1
done
for cfg in "X=0" "MOLANN_DEBUG_RING=16,10,4,0" "MOLANN_DEBUG_RING=16,8,4,1" "MOLANN_DEBUG_RING=16,8,6,0" "MOLANN_DEBUG_RING=16,10,6,0" "MOLANN_DEBUG_RING=16,6,8,0"; do
  echo -n "[C4 $cfg] "
  env $cfg FRAMES=131072 timeout -k 10 200 python tools/prof_one.py C4 2>&1 | tail -1
done
for cfg in "X=0" "MOLANN_DEBUG_RING=11,6,3,0" "MOLANN_DEBUG_RING=11,5,4,0" "MOLANN_DEBUG_RING=11,7,4,0" "MOLANN_DEBUG_RING=11,5,3,1"; do
  echo -n "[C5 $cfg] "
  env $cfg FRAMES=131072 timeout -k 10 200 python tools/prof_one.py C5 2>&1 | tail -1
done
