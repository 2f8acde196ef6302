#!/bin/bash
# Run ON THE GPU BOX: memory-side PMC passes (separate runs) + cycle accounting of one bench workload, any env in front.
#   tools/pmc_mem.sh <workload> <tag> [extra bench args]
set -u
WL=${1:-C3}; TAG=${2:-$WL}; EXTRA=${3:-}
OUT=$PWD/gpurun_out/pmcm_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline $EXTRA"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_MISS_sum TCC_HIT_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 $ARGS > /dev/null 2> "$OUT/pmc$i.err"
  find "$OUT/pmc$i" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc${i}.csv" \;
  rm -rf "$OUT/pmc$i"
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pmc*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("void at::") or "elementwise" in k or "fillBuffer" in k or "distribution" in k or "copyBuffer" in k:
            continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, "launches", len(next(iter(cs.values()))))
    for c, v in cs.items():
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
PY
