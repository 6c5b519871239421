#!/usr/bin/env python3
"""Do the pre-passes leave the same BITS?  Every tile state, every tile's data bits and every end-of-block state of a batch
(gpsbb_test_table_digest, experiments build) as left by
  * the lap-parallel pre-pass (the default),
  * the same with every reference state pushed off the model by up to 1000 / 4e9 grid steps (GPSBB_LAP_JITTER: links that break,
    the repair kernel),
  * the row walks of rounds 1-4 (GPSBB_OPT_SEED_WHERE 1: another implementation altogether),
over workloads at both geometries: chained and independent blocks, Dopplers across seven decades and through zero, sign changes
from block to block, re-allocated and idle channels.  The IQ cannot tell a tile state that is off by one grid step (it changes no
sample): this compares the tables themselves.     python tools/table_check.py [--cases 12] [--seed 1]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workloads(pkg, np, ncases, seed):
    import bench
    rng = np.random.default_rng(seed)
    out = [("headline stream, 24 chained blocks", bench.stream_descriptors(pkg, 24, 16), 25e6, 2500000, pkg.CHAIN_CARRIER),
           ("reference geometry, 40 independent blocks", pkg.synth_descriptors(40, nch=12, seed=0xF00D), 2.6e6, 300000, 0),
           ("reference geometry, 40 chained blocks", bench.stream_descriptors(pkg, 40, 12, seed=0xF00D), 2.6e6, 300000, pkg.CHAIN_CARRIER)]
    for c in range(ncases):
        fs = float(rng.choice([2.6e6, 3e6, 4.092e6, 10e6, 16.368e6, 25e6]))
        nch = int(rng.integers(1, 17 if fs > 15e6 else 13))
        nb = int(rng.integers(2, 30))
        nsamp = int(rng.integers(3000, 400000))
        ch = pkg.synth_descriptors(nb, nch=nch, seed=int(rng.integers(1, 1 << 30)))
        # Dopplers: log-uniform magnitudes over seven decades, both signs, drifting and changing sign from block to block
        mag = 10.0 ** rng.uniform(-3.0, 3.7, nch)
        sign = rng.choice([-1.0, 1.0], nch)
        drift = rng.uniform(-0.02, 0.02, (nb, nch)).cumsum(axis=0)
        f = sign[None, :] * mag[None, :] * (1.0 + drift)
        flip = rng.random(nch) < 0.3
        f[nb // 2:, flip] *= -1.0
        ch["f_carr"] = f
        ch["f_code"] = 1.023e6 + f / 1540.0
        if rng.random() < 0.5:
            i = int(rng.integers(0, nch)); b0 = int(rng.integers(0, nb))
            ch["prn"][b0:, i] = int(rng.integers(1, 33))      # re-allocated
        if rng.random() < 0.4:
            i = int(rng.integers(0, nch)); b0 = int(rng.integers(0, nb)); b1 = int(rng.integers(b0, nb + 1))
            ch["prn"][b0:b1, i] = 0                             # idle for a while
        out.append(("random %d: fs %.4g, %d ch, %d blocks of %d" % (c, fs, nch, nb, nsamp), ch, fs, nsamp,
                    pkg.CHAIN_CARRIER if rng.random() < 0.7 else 0))
    return out


def child(a):
    import numpy as np
    from __graft_entry__ import load_package
    pkg = load_package()
    L = pkg.lib()
    L.gpsbb_test_table_digest.argtypes = [C.c_void_p, C.c_void_p]
    res = []
    with pkg.Synth(0) as s:
        s.set_option(pkg.OPT_SEED_WHERE, a.where)
        for name, ch, fs, nsamp, flags in workloads(pkg, np, a.cases, a.seed):
            b = s.batch(ch, 1.0 / fs, nsamp, flags=flags)
            b.run()
            s.sync()
            d = (C.c_ulonglong * 3)()
            rc = L.gpsbb_test_table_digest(b._b, d)
            res.append({"name": name, "rc": rc, "digest": [int(d[0]), int(d[1]), int(d[2])], "prepass": s.info(pkg.INFO_PREPASS),
                        "kernel": s.info(pkg.INFO_LAST_KERNEL), "links_broken": s.info(pkg.INFO_CHAIN_REPAIRS),
                        "laps_rewalked": s.info(pkg.INFO_CHAIN_FALLBACKS)})
            b.close()
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--where", type=int, default=3)
    a = ap.parse_args()
    if a.child:
        return child(a)
    modes = [("laps (default)", 3, {}), ("laps, references 1000 grid steps off", 3, {"GPSBB_LAP_JITTER": "1000"}),
             ("laps, references 4e9 grid steps off", 3, {"GPSBB_LAP_JITTER": "4000000000"}), ("row walks", 1, {})]
    runs = {}
    for name, where, env in modes:
        e = dict(os.environ, GPSBB_PY_LIB="exp", **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--where", str(where), "--cases", str(a.cases), "--seed", str(a.seed)],
                           env=e, capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit("mode %r failed" % name)
        runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    ref = runs[modes[0][0]]
    bad = 0
    for k, w in enumerate(ref):
        line = "%-52s" % w["name"][:52]
        for name, _, _ in modes:
            x = runs[name][k]
            same = x["digest"] == w["digest"] and x["rc"] == 0
            bad += not same
            line += " | %s %s" % ("==" if same else "DIFFERS", ("links broken %d (cum.), laps re-walked %d" % (x["links_broken"], x["laps_rewalked"])) if x["prepass"] == 3 else "rows")
        print(line)
    print("modes:", ", ".join(m[0] for m in modes))
    print("tables %s in every mode (%d workloads x %d modes)" % ("bit-identical" if not bad else "DIFFER", len(ref), len(modes)))
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
