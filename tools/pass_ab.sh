export TMPDIR=/tmp; R=$PWD
for lib in product $R/build/variants/libgpsbb_head.so; do
  if [ "$lib" = product ]; then unset GPSBB_PY_LIB; else export GPSBB_PY_LIB=$lib; fi
  rm -rf $R/gpurun_out/pp; ( cd /tmp; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pp -o t -- python $R/tools/lap_probe.py 400 0 3 > $R/gpurun_out/pp.log 2>&1 )
  echo "== $lib"; python - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob("$R/gpurun_out/pp/*.db")[0])
for r in c.execute("select name,total_calls,total_duration,average from top_kernels limit 4"): print("  ", r[0][:50], r[1], "avg %.1f us" % r[3])
PY
  rm -rf $R/gpurun_out/pp2; ( cd /tmp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $R/gpurun_out/pp2 -o t -- python $R/tools/lap_probe.py 400 0 3 > $R/gpurun_out/pp2.log 2>&1 )
  python - <<PY
import sqlite3,glob
c=sqlite3.connect(glob.glob("$R/gpurun_out/pp2/*.db")[0])
for k,n,v in c.execute("select kernel_name,counter_name,avg(value) from counters_collection where kernel_name like '%k_lap_pass%' group by kernel_name,counter_name"): print("  ", k.split("::")[1][:16], n, "%.4g"%v)
PY
done
