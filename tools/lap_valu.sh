#!/bin/bash
# vector instructions of the pre-pass kernels per launch, lap-parallel (where 3) against the row walks (where 1), for the headline
# geometry (400 chained blocks of 16 ch at 25 MS/s) and the reference's (1000 independent blocks of 12 ch at 2.6 MS/s)
#   bash tools/lap_valu.sh <tag>       (GPU box; GPSBB_PY_LIB=exp and the experiments knobs apply: e.g. GPSBB_LAP_NO_BURST=1)
TAG="${1:-lapvalu}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
run() { # name, kbench args
  local name=$1; shift
  if [ -n "$ONLY" ] && [[ ! "$name" =~ $ONLY ]]; then return; fi   # ONLY=laps: a subset by name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES -d "$OUT/$name" -o pmc -- python "$ROOT/tools/kbench.py" --steps 2 --warmup 1 "$@" > "$OUT/$name.log" 2>&1 )
  python - "$OUT/$name" "$name" <<'PY'
import sqlite3, glob, sys, os
tot = {}
for db in glob.glob(os.path.join(sys.argv[1], "*.db")):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for k, n, v, cnt in rows:
        if "gpsbb" in k:
            kk = k.split("::")[1].split("(")[0][:28]
            tot.setdefault(kk, {})[n] = v
print("==", sys.argv[2])
pre = 0.0
for k, d in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    print("  %-30s VALU %12.4g  SALU %12.4g  waves %10.4g" % (k, d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_SALU", 0), d.get("SQ_WAVES", 0)))
    if not k.startswith("k_synth"):
        pre += d.get("SQ_INSTS_VALU", 0)
print("  pre-pass VALU per launch (all kernels but the synthesis): %.4g" % pre)
PY
}
run s_laps --blocks 400 --chain --smooth --where 3
run s_rows --blocks 400 --chain --smooth --where 1
run m1_laps --fs 2.6e6 --nsamp 300000 --nch 12 --blocks 1000 --where 3
run m1_rows --fs 2.6e6 --nsamp 300000 --nch 12 --blocks 1000 --where 1
