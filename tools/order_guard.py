#!/usr/bin/env python3
"""Which stream owns what (the ordering contract), as an experiment: a chained stream whose second push
follows the first by more than GPSBB_X_PARK_NULL_MS.  Anything of the library's that still went through the null stream — the
zeroing of a fresh stream's carry did until round 4 — lands between the two pushes and the second one chains from a wiped
phase.  Run on the experiments build: prints one JSON line, "equal": whether every block equals the oracle's.
   GPSBB_PY_LIB=exp GPSBB_X_PARK_NULL_MS=60 [GPSBB_X_NULL_MEMSET=1] python tools/order_guard.py [where]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
import oracle_binding as ob
pkg = g.load_package()
where = int(sys.argv[1]) if len(sys.argv) > 1 else 0
park_ms = int(os.environ.get("GPSBB_X_PARK_NULL_MS", "0"))
fs, nsamp, nch, bps, npush = 25e6, 60000, 8, 4, 4
ch = pkg.synth_descriptors(bps * npush, nch=nch, seed=2024)
want, want_st, _ = ob.Oracle().fill_blocks(ch, 1 / fs, nsamp, chain=True)
out = {"lib": os.path.basename(pkg.LIB_PATH), "park_ms": park_ms, "null_memset": bool(os.environ.get("GPSBB_X_NULL_MEMSET")), "where": where, "runs": []}
for rep in range(3):
    with pkg.Synth(0) as s:   # a fresh handle every time: its first stream's carry is zeroed at the first push
        s.set_option(pkg.OPT_SEED_WHERE, where)
        st = s.stream(nch, 1 / fs, nsamp, bps, depth=2, flags=pkg.CHAIN_CARRIER)
        bad = 0
        for k in range(npush):
            st.push(ch[k * bps:(k + 1) * bps])
            iq, es = st.pop()
            bad += int((np.asarray(iq).reshape(bps, nsamp, 2) != want[k * bps:(k + 1) * bps]).any(axis=(1, 2)).sum())
            time.sleep(2.5 * park_ms / 1000.0)   # what was parked has landed before the next push reads the carry
        st.close()
        out["runs"].append({"blocks_that_differ": bad, "prepass": s.info(pkg.INFO_PREPASS)})
out["equal"] = all(r["blocks_that_differ"] == 0 for r in out["runs"])
print(json.dumps(out))
