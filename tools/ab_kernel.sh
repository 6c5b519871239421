#!/bin/bash
# usage: tools/ab_kernel.sh tagA tagB ... : k_synth_ev alone and pipelined per library variant (libgpsbb_<tag>.so; "" = the product), interleaved, same box
for rep in 1 2 3; do
  for t in "$@"; do
    if [ "$t" = "product" ]; then export -n GPSBB_PY_LIB; unset GPSBB_PY_LIB; else export GPSBB_PY_LIB=$t; fi
    a=$(python tools/kbench.py --no-cpu --steps 8 --synth-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['ms_per_launch'])")
    p=$(python tools/kbench.py --no-cpu --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %.4f' % (d['roofline']['ms_per_launch'], d['ms_per_step']))")
    echo "$t alone $a pipelined(kernel,step) $p"
  done
done
