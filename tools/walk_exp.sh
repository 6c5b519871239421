#!/bin/bash
export GPSBB_PY_LIB=exp   # the environment knobs below exist in the experiments build only (libgpsbb_exp.so)
for m in "" "-DGPSBB_EXP_NOROWSTORE"; do
  make -C pluto-gps-sim_amd/csrc EXTRA="$m" -B >/dev/null 2>&1
  for l in 64 32 16 8; do
    echo "[$m] lanes/wave $l: $(GPSBB_WALK_LANES=$l python tools/seed_alone.py 400 2>&1 | grep 'run 5')"
  done
done
make -C pluto-gps-sim_amd/csrc -B >/dev/null 2>&1
