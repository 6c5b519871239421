#!/bin/bash
# HBM bytes the lap passes write per 400-block push of the headline geometry (WRITE_SIZE, KiB -> bytes), this build and another one
#   bash tools/pass_writes.sh [other libgpsbb.so]
export TMPDIR=/tmp; R=$PWD
for lib in product "$@"; do
  if [ "$lib" = product ]; then unset GPSBB_PY_LIB; else export GPSBB_PY_LIB=$lib; fi
  rm -rf $R/gpurun_out/pw; ( cd /tmp; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pw -o t -- python $R/tools/lap_probe.py 400 0 3 > $R/gpurun_out/pw.log 2>&1 )
  echo "== $lib"; python - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob("$R/gpurun_out/pw/*.db")[0])
for k,n,v in c.execute("select kernel_name,counter_name,avg(value) from counters_collection where kernel_name like '%k_lap_pass%' or kernel_name like '%k_synth_ev%' group by kernel_name,counter_name"): print("   %-18s %s %.4g bytes" % (k.split("::")[1][:16], n, v * 1024))
PY
done
