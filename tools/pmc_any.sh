#!/bin/bash
# one counter pass over any command: bash tools/pmc_any.sh <tag> "<counters>" <cmd...>   (kernel-trace + pmc only)
set -u
TAG="$1"; CTRS="$2"; shift 2
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/pmc" -o pmc -- "$@" > "$OUT/pmc_stdout.log" 2>&1 )
python - "$OUT" <<'PY'
import sqlite3, glob, sys, os
for db in glob.glob(os.path.join(sys.argv[1], "pmc", "*.db")):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for k, n, v, cnt in rows:
        if "gpsbb" in k:
            print("%-12s %-28s %18.1f  (n=%d)" % (k.split("::")[1][:12], n, v, cnt))
PY
