#!/bin/bash
# Run ON THE GPU BOX: two PMC passes (instruction counts, cycle accounting) of one bench workload, any env in front.
#   [MOLANN_DIAG_LIB=1 MOLANN_DEBUG_ABLATE=64] tools/pmc_quick.sh <workload> <tag> [extra bench args]
set -u
WL=${1:-C3}; TAG=${2:-$WL}; EXTRA=${3:-}
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline $EXTRA"
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 $ARGS > /dev/null 2> "$OUT/pmc$i.err"
  find "$OUT/pmc$i" -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "frames_|mlp_mfma|molann_lane_jit" "$1" >> "$2"' _ {} "$OUT/pmc${i}_counters.csv" \;
done
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/pmc[0-9]
cat "$OUT/summary.txt"
