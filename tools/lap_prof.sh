#!/bin/bash
# per-kernel times of a resident chained 400-block batch (the bench's descriptors): the lap-parallel pre-pass and the row walks
# usage (GPU box): bash tools/lap_prof.sh <tag>
TAG="${1:-lap}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
for w in 3 1; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_w$w" -o trace -- python "$ROOT/tools/lap_probe.py" 400 0 $w > "$OUT/prof_w${w}_stdout.log" 2>&1 )
  f=$(find "$OUT/prof_w$w" -name "*kernel_stats.csv" | head -1)
  echo "== where $w"; tail -2 "$OUT/prof_w${w}_stdout.log"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %5s avg %10.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
