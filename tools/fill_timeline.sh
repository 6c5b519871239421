#!/bin/bash
# one drop-in call's kernels and copies on one time axis (GPU box):   [FILL_REG=1] [FILL_TL_EXTRA=--hip-runtime-trace] bash tools/fill_timeline.sh <tag>
TAG="${1:-fill_tl}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace $FILL_TL_EXTRA -d "$OUT/tl" -o tl -- python "$ROOT/tools/fill_timeline.py" > "$OUT/run.log" 2>&1 )
grep "^calls" "$OUT/run.log"
python - "$OUT/tl" <<'PY'
import sqlite3, glob, sys, os
for db in glob.glob(os.path.join(sys.argv[1], "*.db")):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    kt = [t for t in tabs if t == "kernels"]
    if kt:
        for name, s, e in c.execute("select name, start, end from kernels"):
            ev.append((s, e, name.split("(")[0].replace("gpsbb_impl::", "")[:40]))
    mt = [t for t in tabs if t == "memory_copies"]
    if mt:
        cols = [r[1] for r in c.execute("pragma table_info(memory_copies)")]
        szc = "size" if "size" in cols else None
        q = "select name, start, end%s from memory_copies" % (", size" if szc else "")
        for r in c.execute(q):
            ev.append((r[1], r[2], "copy %s %s" % (r[0], r[3] if szc else "")))
    for t in tabs:
        if t == "regions":
            cols = [r[1] for r in c.execute("pragma table_info(regions)")]
            try:
                for name, s, e in c.execute("select name, start, end from regions"):
                    ev.append((s, e, "    host " + name))
            except Exception as ex:
                print("regions:", cols, ex)
    ev.sort()
    # a call from the middle of the run: from 70 us before its plan kernel to 70 us before the next one's
    plans = [s for s, e, n in ev if "k_lap_plan2" in n]
    a, b = plans[-4] - 70000, plans[-3] - 70000
    for s, e, n in ev:
        if a <= s < b:
            print("  %8.1f .. %8.1f us  (%6.1f)  %s" % ((s - a) / 1e3, (e - a) / 1e3, (e - s) / 1e3, n))
PY
