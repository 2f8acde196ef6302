#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + PMC passes of one bench workload.
#   tools/profile.sh <workload> <tag>     -> gpurun_out/prof_<tag>/{stats,pmc_*}.csv summaries
# PMC passes are separate runs with nothing but --pmc (gpurun refuses --pmc mixed with trace domains).
set -u
WL=${1:-C3}; TAG=${2:-$WL}; EXTRA=${3:-}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/trace" -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > "$2"' _ {} "$OUT/kernel_trace_head.csv" \;
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F32" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 $ARGS > /dev/null 2> "$OUT/pmc$i.err"
  find "$OUT/pmc$i" -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "frames_|mlp_mfma|mlp_lane_kernel|molann_mlp_chain|pack_|molann_lane_jit" "$1" >> "$2"' _ {} "$OUT/pmc${i}_counters.csv" \;
done
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/trace "$OUT"/pmc[0-9] 
cat "$OUT/summary.txt"
