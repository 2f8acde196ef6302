#!/bin/bash
# Run ON THE GPU BOX: the sub-line fetch microbenchmark (tools/micro/subline.hip), timed, then under rocprofv3 --pmc
# (separate passes; the program directly after `--`) for the size of the read requests L2 sends to memory.
set -u
OUT=$PWD/gpurun_out/subline
mkdir -p "$OUT"
export TMPDIR=/tmp
GIB=${1:-3}
./tools/micro/subline $GIB > "$OUT/timing.txt" 2>&1 || { echo "subline failed"; tail -5 "$OUT/timing.txt"; exit 1; }
i=0
for CTRS in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_MISS_sum" \
            "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum TCC_HIT_sum" \
            "FETCH_SIZE" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- ./tools/micro/subline $GIB > /dev/null 2> "$OUT/pmc$i.err"
  find "$OUT/pmc$i" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc${i}.csv" \;
  rm -rf "$OUT/pmc$i"
done
python3 - "$OUT" <<'PY' > "$OUT/counters.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pmc*.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("kernel (variant, stride); counters are averages over the launches of one run")
for k in acc:
    print(k[:40].ljust(40), "  ".join("%s=%.4g" % (c, sum(acc[k][c]) / len(acc[k][c])) for c in names if c in acc[k]))
PY
cat "$OUT/timing.txt"
