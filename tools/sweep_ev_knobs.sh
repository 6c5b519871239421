export GPSBB_PY_LIB=exp
run() { a=$(python tools/kbench.py --no-cpu --steps 8 --synth-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['ms_per_launch'])"); p=$(python tools/kbench.py --no-cpu --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['roofline']['ms_per_launch'], d['ms_per_step']))"); echo "$1 alone $a pipelined $p"; }
run default
GPSBB_EV_CHUNK=1 run chunk1
GPSBB_EV_CHUNK=3 run chunk3
GPSBB_EV_CHUNK=4 run chunk4
GPSBB_EV_HELPERS=1 run helpers1x256
GPSBB_EV_HELPERS=3 run helpers3x256
GPSBB_EV_HELPERS=4 run helpers4x256
run default
