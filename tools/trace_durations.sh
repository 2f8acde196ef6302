#!/bin/bash
# Run ON THE GPU BOX: durations of one kernel's dispatches, in launch order, and the gaps between them.
#   tools/trace_durations.sh <kernel-name substring> bench.py [args]
set -u
PAT=$1; shift
OUT=$PWD/gpurun_out/trace_dur
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 "$@" > "$OUT/stdout.txt" 2> "$OUT/err.txt"
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
s = [int(r["Start_Timestamp"]) / 1e3 for r in rows]
print("n =", len(d))
print("durations us:", " ".join("%.1f" % v for v in d))
print("gaps us     :", " ".join("%.1f" % (s[i + 1] - s[i] - d[i]) for i in range(len(d) - 1)))
PY
rm -rf "$OUT/t"
