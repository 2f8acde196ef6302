#!/bin/bash
# diagnostic: price the stages of frames_lane_kernel by skipping them (outputs are wrong on purpose)
WL=${1:-C3}
for m in ${ABLATE_LIST:-0 1 2 4 8 9 6 15}; do
  MOLANN_DEBUG_ABLATE=$m python bench.py --workload $WL --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ablate=$m  launch_ms_avg=%.4f' % d['roofline']['launch_ms_avg'])"
done
