#!/bin/bash
# Run ON THE GPU BOX: PMC passes of any python tool, per kernel name (average over its launches).
#   tools/pmc_any.sh <tag> <kernel-name regex> tools/time_mlp_bwd.py [args]
set -u
TAG=$1; PAT=$2; shift 2
OUT=$PWD/gpurun_out/pmca_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 "$@" > /dev/null 2> "$OUT/pmc$i.err"
  find "$OUT/pmc$i" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc${i}.csv" \;
  rm -rf "$OUT/pmc$i"
done
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, re, sys, collections
out, pat = sys.argv[1], re.compile(sys.argv[2])
for f in sorted(glob.glob(out + "/pmc*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, "launches", len(next(iter(cs.values()))))
        for c, v in cs.items():
            print("   %-28s %14.0f" % (c, sum(v) / len(v)))
PY
