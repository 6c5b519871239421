#!/bin/bash
# kernel timeline of any command: bash tools/timeline.sh <tag> <cmd...>
TAG="$1"; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$OUT/prof" -o trace -- "$@" > "$OUT/stdout.log" 2>&1 )
python - "$OUT" <<'PY'
import sqlite3, glob, sys, os
for db in glob.glob(os.path.join(sys.argv[1], "prof", "*.db")):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
    kt = [t for t in tabs if "kernel_dispatch" in t or t == "kernels"]
    rows = c.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
    t0 = rows[0][1]
    for n, s, e, st, q in rows[-80:]:
        print("%-22s start %9.3f ms  dur %8.3f ms  stream %s queue %s" % (n.split("(")[0][-22:], (s - t0) / 1e6, (e - s) / 1e6, st, q))
PY
