#!/bin/bash
# One GPU-box session of the co-run diagnosis: the in-kernel workgroup trace of both synthesis kernels in every
# case (tools/corun_diag.py, trace build), then SQ counter passes of k_synth_ev alone and beside the lap passes.
#   bash tools/corun_session.sh <tag>
set -u
TAG="${1:-r06_corun}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
[ -f build/variants/libgpsbb_trace.so ] || make -C pluto-gps-sim_amd/csrc trace VARIANT_DIR=$PWD/build/variants > "$OUT/make_trace.log" 2>&1
timeout 900 python tools/corun_diag.py --out "$OUT" > "$OUT/corun_diag.txt" 2> "$OUT/corun_diag.err"
cat "$OUT/corun_diag.txt"; tail -5 "$OUT/corun_diag.err"
if [ "${2:-}" != "nopmc" ]; then
timeout 900 bash tools/pmc_sq.sh ${TAG}_sq_alone --blocks 400 --chain --smooth --synth-only > "$OUT/sq_alone.txt" 2>&1
timeout 900 bash tools/pmc_sq.sh ${TAG}_sq_beside --blocks 400 --chain --smooth > "$OUT/sq_beside.txt" 2>&1
grep -h "k_synth_ev " "$OUT/sq_alone.txt" | sed 's/^/alone  /'
grep -h "k_synth_ev " "$OUT/sq_beside.txt" | sed 's/^/beside /'
# do kernels overlap under --pmc at all?  (the kernel trace of the "beside" counter pass)
python - "$PWD/gpurun_out/${TAG}_sq_beside/sq1" <<'PY'
import sqlite3, glob, sys, os
for db in glob.glob(os.path.join(sys.argv[1], "*.db")):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    syn = [(s, e) for n, s, e in rows if "k_synth_ev" in n]
    lap = [(s, e) for n, s, e in rows if "k_lap_pass" in n]
    ov = sum(1 for s, e in lap if any(s < e2 and e > s2 for s2, e2 in syn))
    print("counter pass: %d k_synth_ev launches, %d lap-pass launches, %d of them overlap a k_synth_ev in time" % (len(syn), len(lap), ov))
PY
fi
