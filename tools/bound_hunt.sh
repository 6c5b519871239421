#!/bin/bash
# What is k_synth_ev's time made of?  Builds deliberately wrong variants of the kernel that take one resource out of the
# picture each (GPSBB_X_*: tools/experiments/bound_hunt_variants.patch) and times the synthesis kernel alone on the headline geometry, interleaved
# with the product, on this box.  usage: [KARGS="--fs 2.6e6 --nsamp 300000 --nch 12 --blocks 1000"] tools/bound_hunt.sh [variant flags ...]
# (run from the repo root on the GPU box; KARGS: kbench.py's geometry options, e.g. the reference's own for k_synth_pd)
# The variants are NOT in the product's sources: tools/experiments/bound_hunt_variants.patch puts their #ifdef GPSBB_X_* blocks
# into a scratch copy of csrc/ (made against round 5's kernels: if it no longer applies, the kernels have moved on and the
# variants have to be re-cut).
set -e
V=${@:-"NOATOMIC NOADD NOAMP NOSMEM NOSTORE"}
SCR=$(mktemp -d)
cp -r pluto-gps-sim_amd/csrc "$SCR/csrc"
( cd "$SCR" && patch -p2 -d . < "$OLDPWD/tools/experiments/bound_hunt_variants.patch" > /dev/null )
INC=$PWD/include
OUT=$PWD/pluto-gps-sim_amd
cd "$SCR/csrc"
for v in $V; do
  fl=""; for f in ${v//+/ }; do fl="$fl -DGPSBB_X_$f"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$INC -I. $fl -shared gpsbb.hip gpsbb_node.cpp -o $OUT/libgpsbb_x$v.so &
done
wait
cd "$OLDPWD"
rm -rf "$SCR"
for rep in 1 2; do
  for v in product $V; do
    if [ "$v" = "product" ]; then unset GPSBB_PY_LIB; else export GPSBB_PY_LIB=x$v; fi
    a=$(python tools/kbench.py --no-cpu --steps 8 --synth-only $KARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['ms_per_launch'])")
    echo "$v alone_ms $a"
  done
done
rm -f pluto-gps-sim_amd/libgpsbb_x*.so
