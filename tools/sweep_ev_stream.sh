export GPSBB_PY_LIB=exp   # the environment knobs below exist in the experiments build only (libgpsbb_exp.so)
for o in 2 3 4 6; do for c in 1 2 4; do
r=$(GPSBB_EV_OVERSUB=$o GPSBB_EV_CHUNK=$c python bench.py --no-extras --steps 10 --repeats 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g S/s synth %.3f' % (d['value'], d['roofline']['ms_per_launch']))")
echo "oversub $o chunk $c: $r"
done; done
