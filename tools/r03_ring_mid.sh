#!/bin/bash
# Run ON THE GPU BOX: ring geometry (slots, consumers, loaders, depth) and consumer parts on a workload (default P2)
cd "$(dirname "$0")/.."
export MOLANN_DIAG_LIB=1
W=${1:-P2}
for cfg in "X=0" "MOLANN_DEBUG_RING_FLAGS=2" "MOLANN_DEBUG_RING_FLAGS=3" "MOLANN_DEBUG_RING=16,6,6,0" "MOLANN_DEBUG_RING=16,5,7,0" "MOLANN_DEBUG_RING=16,9,3,1" "MOLANN_DEBUG_RING=16,10,2,2"; do
  echo -n "[$cfg] "
  env $cfg FRAMES=262144 timeout -k 10 200 python tools/prof_one.py $W 2>&1 | tail -1
done
