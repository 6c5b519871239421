#!/bin/bash
# k_tiles' geometry (rows per lane and turn x lanes per workgroup): its vector instruction count and time per variant library,
# exact counters, one 400-block batch (exp: tools/pmc_sq.sh's first pass only).  usage: tools/tiles_geom.sh <lib tag> ...
export TMPDIR=/tmp
for t in "$@"; do
  if [ "$t" = "product" ]; then unset GPSBB_PY_LIB; else export GPSBB_PY_LIB=$t; fi
  OUT=$PWD/gpurun_out/tg_$t; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o pmc -- python /root/repo/tools/kbench.py --steps 2 --warmup 1 --no-cpu --blocks 400 > $OUT/log 2>&1 )
  python - $OUT $t <<'PY'
import sqlite3, glob, sys, os
for db in glob.glob(os.path.join(sys.argv[1], "*.db")):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    d = {}
    for k, n, v in rows:
        if "k_tiles" in k or "k_walk<2>" in k: d[(k.split("::")[1].split("(")[0], n)] = v
    print(sys.argv[2], {("%s %s" % k): "%.3e" % v for k, v in sorted(d.items())})
    try:
        t = c.execute("select name, avg(end-start) from kernels group by name").fetchall()
    except Exception: t = []
PY
done
