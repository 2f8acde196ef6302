#!/bin/bash
# PMC of the alignment backward kernel on A4 (fewer frames to keep it short)
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/pmc_a4bwd; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/p$i" -- python3 tools/time_backward.py A4 > /dev/null 2> "$OUT/p$i.err"
  find "$OUT/p$i" -name "*counter_collection.csv" -exec cp {} "$OUT/p$i.csv" \;
  rm -rf "$OUT/p$i"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/p*.csv")):
    for r in csv.DictReader(open(f)):
        if "frames_align" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in cs.items(): print("   %-24s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
