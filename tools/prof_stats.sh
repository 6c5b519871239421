#!/bin/bash
# rocprofv3 kernel-trace stats of a bench run: bash tools/prof_stats.sh <tag> [bench args...]
set -u
TAG="${1:-prof}"; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$ROOT/tools/kbench.py" "$@" > "$OUT/prof_stdout.log" 2>&1 )
python - "$OUT" <<'PY'
import sqlite3, glob, sys, os
for db in glob.glob(os.path.join(sys.argv[1], "prof", "*.db")):
    c = sqlite3.connect(db)
    for n, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-70s calls %5d  avg %10.1f us  %5.1f%%" % (n[:70], calls, avg/1000.0 if avg > 1e5 else avg, pct))
PY
tail -2 "$OUT/prof_stdout.log" | cut -c1-400
