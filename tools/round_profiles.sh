#!/bin/bash
# Run ON THE GPU BOX: the round's bench lines and rocprofv3 summaries.   tools/round_profiles.sh r02 C3 C2 ...
R=$1; shift
mkdir -p gpurun_out/$R
for WL in "$@"; do
  EXTRA=""
  case $WL in C4|C5) EXTRA="--steps 5 --warmup 2";; esac
  timeout -k 10 600 python bench.py --workload $WL $EXTRA > gpurun_out/$R/bench_$WL.json 2> gpurun_out/$R/bench_$WL.err
  echo "bench $WL rc=$? $(cut -c1-160 gpurun_out/$R/bench_$WL.json)"
  case $WL in C4|C5) PEX="--steps 3 --warmup 1";; *) PEX="";; esac
  timeout -k 10 900 bash tools/profile.sh $WL ${R}_$WL "$PEX" > gpurun_out/$R/profile_$WL.log 2>&1
  echo "profile $WL rc=$?"
done
