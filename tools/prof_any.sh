#!/bin/bash
# rocprofv3 kernel-trace stats of any python command: bash tools/prof_any.sh <tag> <script> [args...]
set -u
TAG="$1"; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$ROOT/$1" "${@:2}" > "$OUT/prof_stdout.log" 2>&1 )
python - "$OUT" <<'PY'
import sqlite3, glob, sys, os
for db in glob.glob(os.path.join(sys.argv[1], "prof", "*.db")):
    c = sqlite3.connect(db)
    for n, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-70s calls %5d  avg %10.1f us  %5.1f%%" % (n[:70], calls, avg / 1000.0, pct))
PY
tail -8 "$OUT/prof_stdout.log" | grep -v amdgpu.ids | cut -c1-300
