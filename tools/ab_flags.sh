#!/bin/bash
# A/B of diagnostic settings on a bench workload (diagnostics library): tools/ab_flags.sh "NAME=VALUE ..." "NAME=VALUE ..." ...
# WL=C3 FRAMES=0 (workload default) REPS=2
cd "$(dirname "$0")/.."
export MOLANN_DIAG_LIB=1
WL=${WL:-C3}; FR=${FRAMES:-0}; REPS=${REPS:-2}
for rep in $(seq $REPS); do
for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 120 python bench.py --workload $WL --frames $FR --buffers ${BUFFERS:-0} --diagnostic --no-cpu-baseline --steps ${STEPS:-100} --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); f=d['config']['frames_per_gpu']; t=d['roofline']['launch_ms_avg']*1e3; print('%.2f us (%.2f per 1M frames)' % (t, t*1048576/f), d['config']['kernels'][40:95])"
done; done
