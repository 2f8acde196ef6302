#!/bin/bash
# A/B of diagnostic settings on the C3 bench (diagnostics library): tools/ab_flags.sh "NAME=VALUE ..." "NAME=VALUE ..." ...
cd "$(dirname "$0")/.."
export MOLANN_DIAG_LIB=1
WL=${WL:-C3}
for rep in 1 2; do
for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 120 python bench.py --workload $WL --diagnostic --no-cpu-baseline --steps 100 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.2f us' % (d['roofline']['launch_ms_avg']*1e3), d['config']['kernels'])"
done; done
