export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_bwd; mkdir -p $OUT
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F32" \
            "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/p$i" -- python3 tools/time_backward.py C3 > /dev/null 2> "$OUT/p$i.err"
  find "$OUT/p$i" -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "molann_lane_bwd|molann_lane_jit" "$1" >> "$2"' _ {} "$OUT/pmc${i}_counters.csv" \;
  rm -rf "$OUT/p$i"
done
python3 tools/summarize_pmc.py "$OUT"
