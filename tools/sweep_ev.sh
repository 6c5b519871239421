#!/bin/bash
export GPSBB_PY_LIB=exp   # the environment knobs below exist in the experiments build only (libgpsbb_exp.so)
# sweep launch parameters of k_synth_ev (synth-only timing)
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], "synth %.3f ms" % d["roofline"]["ms_per_launch"])'
for ov in ${OVS:-1 2 3 4 8 16}; do for ch in ${CHS:-1 2 4 8}; do for mw in ${MWS:-3}; do
  GPSBB_EV_MIN_WG=$mw GPSBB_EV_OVERSUB=$ov GPSBB_EV_CHUNK=$ch timeout 300 python tools/kbench.py --steps 6 --warmup 3 --no-cpu --synth-only 2>&1 | python -c "$P" "oversub $ov chunk $ch minwg $mw"
done; done; done
