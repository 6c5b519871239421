#!/usr/bin/env python3
"""The error budgets of the model kernels, MEASURED (gpsbb_modelerr.hip.h).

Run with the experiments build (GPSBB_PY_LIB=exp).  For a set of workloads at the corners of what k_synth_ev /
k_synth_ev_dense / k_synth_ev_fixed / k_synth_pd take — Dopplers at the edge of every breakpoint class, both signs,
near-zero steps, |f_carr*delt| up to the contract's 0.125, code rates up to the chip table's reach, chained batches cut
into segments, blocks whose last tile is partial, and the grazing descriptors of grazing_descriptors() — every batch is
rendered (checked against the CPU oracle where asked) and then every tile of it is replayed next to the reference's own
recurrence.  Prints one JSON object: per workload the realised maxima in units of 2^-32, the same as fractions of the
budget (EvConst::W resp. PD_BAND), the number of lanes the danger test flagged (== the kernel's own count), and the
number of unflagged decisions that differ from the truth (must be 0)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
try:
    import torch  # noqa: F401  (one HIP runtime per process: torch's, if torch is there)
except Exception:
    pass
from __graft_entry__ import load_package  # noqa: E402

ME_NQ, MEC_NQ = 11, 4
QNAMES = ["y0", "y0_over_W", "tk", "tk_over_W", "x0", "x0_over_W", "tc", "tc_over_W", "pure_y", "pure_x", "W_units"]
OFFSETS = [0, 1, -1, 2, -2, 3, -3, 4, -4, 6, -6, 8, -8, 10, -10, 12, -12, 13, -13, 14, -14, 16, -16, 20, -20, 24, -24, 28, -28,
           32, -32, 48, -48]


def measure(pkg, synth, ch, fs, nsamp, flags=0, check=None):
    L = pkg.lib()
    L.gpsbb_test_model_err.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    before = synth.info(pkg.INFO_EXACT_RUNS)
    b = synth.batch(ch, 1.0 / fs, nsamp, flags=flags)
    b.run()
    synth.sync()
    exact = synth.info(pkg.INFO_EXACT_RUNS) - before
    mx = np.zeros((pkg.MAX_CHAN, ME_NQ), np.float64)
    cnt = np.zeros((pkg.MAX_CHAN, MEC_NQ), np.uint64)
    which = C.c_int(0)
    rc = L.gpsbb_test_model_err(b._b, mx.ctypes.data, cnt.ctypes.data, C.byref(which))
    if rc != 0:
        b.close()
        return {"skipped": "not a model kernel (rc %d)" % rc}
    mismatch = None
    if check is not None:
        iq, _ = b.read()
        want_iq, _, _ = check.fill_blocks(ch, 1.0 / fs, nsamp, chain=bool(flags & pkg.CHAIN_CARRIER), fixed=bool(flags & pkg.FIXED_CARRIER))
        mismatch = int((iq != want_iq).sum())
    b.close()
    out = {"kernel": {1: "k_synth_ev*", 2: "k_synth_pd"}[which.value], "max": {q: float(mx[:, k].max()) for k, q in enumerate(QNAMES)},
           "bad_unflagged_decisions": int(cnt[:, 0].sum()), "lanes_flagged": int(cnt[:, 1].sum()), "lane_runs": int(cnt[:, 2].sum()),
           "always_exact": int(cnt[:, 3].sum()), "kernel_exact_runs": int(exact)}
    if mismatch is not None:
        out["iq_mismatches_vs_oracle"] = mismatch
    return out


def workloads(pkg, quick):
    """(name, ch, fs, nsamp, flags, check against the oracle)"""
    rng = np.random.default_rng(404)
    W = []
    nb = 4 if quick else 12

    def dopp(nblocks, nch, fs, values, seed, fixed=False):
        ch = pkg.synth_descriptors(nblocks, nch=nch, seed=seed)
        v = np.resize(np.asarray(values, np.float64), nblocks * nch).reshape(nblocks, nch)
        ch["f_carr"] = v
        ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
        if fixed:
            ch["carr_phase"] = np.floor(ch["carr_phase"] * 2.0 ** 32)
        return ch

    fs = 25e6
    # the headline workload's own Dopplers
    W.append(("ev_25MS_5kHz", pkg.synth_descriptors(nb, 16, seed=0x5EED), fs, 250000, 0, True))
    # the edges of the breakpoint classes: |S|*15.5 = 1, 2, 3, 4 index changes per run, either sign, just below and above
    edges = []
    for kc in (1, 2, 3, 4):
        f = kc / 15.5 / 512.0 * fs
        edges += [f * 0.9999, -f * 0.9999, f * 1.0001, -f * 1.0001]
    W.append(("ev_25MS_class_edges", dopp(nb, 16, fs, edges, 11), fs, 131071, 0, True))
    # steps so small that the change positions' bias grows (W ~ 2^-32 / S), down to the always-exact class, and zero
    tiny = [0.0, 1e-3, -1e-3, 0.05, -0.05, 0.7, -0.7, 3.0, -3.0, 11.0, -11.0, 40.0, -40.0, 150.0, -150.0, 600.0]
    W.append(("ev_25MS_tiny_steps", dopp(nb, 16, fs, tiny, 12), fs, 100000, 0, True))
    # chained, cut into segments (the tile states at segment starts come through the fix-up)
    W.append(("ev_25MS_chained_segments", pkg.synth_descriptors(3, 16, seed=77), fs, 2500000, pkg.CHAIN_CARRIER, False))
    # code rate at the breakpoint path's limit (one chip change per run) and a mixed batch just past it
    W.append(("ev_16p0MS_one_chip_per_run", pkg.synth_descriptors(nb, 16, seed=13), 16.0e6, 90001, 0, True))
    chm = pkg.synth_descriptors(nb, 12, seed=14)
    chm["f_code"] = 1.023e6 + np.where(np.arange(12) % 3 == 0, -40.0, 40.0)[None, :]
    W.append(("ev_dense_mixed_15p8565MS", chm, 15.8565e6, 70000, 0, True))
    W.append(("ev_30MS", pkg.synth_descriptors(nb, 16, seed=15), 30e6, 120000, 0, True))
    # the fixed-point carrier on the breakpoint kernel (code side only: the accumulator's model is exact)
    chf = pkg.synth_descriptors(nb, 16, seed=16)
    chf["carr_phase"] = np.floor(chf["carr_phase"] * 2.0 ** 32)
    W.append(("ev_fixed_25MS", chf, fs, 100000, pkg.FIXED_CARRIER, True))
    # grazing: states aimed within +-k units of an integer at a sample
    g, _ = pkg.grazing_descriptors(nb, 16, fs, 100000, OFFSETS, seed=21)
    W.append(("ev_25MS_grazing", g, fs, 100000, 0, True))
    g, _ = pkg.grazing_descriptors(nb, 16, fs, 100000, OFFSETS, seed=22, max_doppler=12000.0,
                                   samples=[16, 15, 1008, 1023, 1024, 1025, 2047, 99999, 99984, 5000, 777])
    W.append(("ev_25MS_grazing_run_and_tile_edges", g, fs, 100000, 0, True))

    # ---- k_synth_pd: the reference's own geometry and its corners ----
    fs = 2.6e6
    W.append(("pd_2p6MS_12ch", pkg.synth_descriptors(nb, 12, seed=4242), fs, 300000, 0, True))
    W.append(("pd_2p6MS_16ch", pkg.synth_descriptors(nb, 16, seed=4243), fs, 300000, 0, True))
    # |f_carr*delt| at the contract's limit (64 table entries per sample), either sign, and powers of two
    big = [0.124999 * fs, -0.124999 * fs, 0.1249 * fs, -0.1249 * fs, 0.0625 * fs, -0.03125 * fs, 1e5, -1e5]
    W.append(("pd_2p6MS_carrier_limit", dopp(nb, 12, fs, big, 31), fs, 100000, 0, True))
    W.append(("pd_1p97MS_code_limit", pkg.synth_descriptors(nb, 12, seed=32), 1.97e6, 100000, 0, True))
    W.append(("pd_10MS", pkg.synth_descriptors(nb, 16, seed=33), 10e6, 100000, 0, True))
    W.append(("pd_2p6MS_chained", pkg.synth_descriptors(6, 12, seed=34), fs, 300000, pkg.CHAIN_CARRIER, True))
    chf = pkg.synth_descriptors(nb, 12, seed=35)
    chf["carr_phase"] = np.floor(chf["carr_phase"] * 2.0 ** 32)
    W.append(("pd_fixed_2p6MS", chf, fs, 300000, pkg.FIXED_CARRIER, True))
    g, _ = pkg.grazing_descriptors(nb, 12, fs, 100000, OFFSETS, seed=41, max_doppler=20000.0)
    W.append(("pd_2p6MS_grazing", g, fs, 100000, 0, True))
    g, _ = pkg.grazing_descriptors(nb, 16, fs, 100000, OFFSETS, seed=42, max_doppler=300000.0,
                                   samples=[64, 63, 960, 1023, 1024, 1087, 99999, 99936, 4097])
    W.append(("pd_2p6MS_grazing_fast_carriers", g, fs, 100000, 0, True))
    del rng
    return W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="fewer blocks per workload (the -m gpu test)")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--out")
    a = ap.parse_args()
    pkg = load_package()
    assert os.environ.get("GPSBB_PY_LIB") == "exp", "run with GPSBB_PY_LIB=exp (the hook lives in the experiments build)"
    import oracle_binding as ob
    oracle = None if a.no_oracle else ob.Oracle()
    res = {}
    with pkg.Synth(0) as s:
        s.set_option(pkg.OPT_SEED_WHERE, int(os.environ.get("GPSBB_MODEL_ERR_WHERE", "0")))  # 0: the library's choice (lap-parallel), what production runs on
        for name, ch, fs, nsamp, flags, check in workloads(pkg, a.quick):
            res[name] = measure(pkg, s, ch, fs, nsamp, flags, oracle if check else None)
            res[name]["fs"] = fs
            res[name]["nsamp"] = nsamp
    worst = {}
    for r in res.values():
        for q, v in r.get("max", {}).items():
            worst[q] = max(worst.get(q, 0.0), v)
    bud = (C.c_double * 3)()
    pkg.lib().gpsbb_test_budgets(bud)
    doc = {"workloads": res, "worst": worst,
           "budgets": {"EV_MODEL_ERR_units": bud[0], "EV_T_EPS_units": bud[1], "PD_BAND_units": bud[2],
                       "note": "units of 2^-32; *_over_W = realised / budget of that channel"}}
    txt = json.dumps(doc, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    print(txt)
    print("MODEL_ERR_DONE")


if __name__ == "__main__":
    main()
