#!/usr/bin/env python3
"""Fuzz the error BUDGETS of the model kernels, not only their bytes (experiments build: GPSBB_PY_LIB=exp).

tools/model_err.py measures realised / budget on 19 hand-picked workloads; this runs thousands of random ones — sample rates
across what k_synth_ev / k_synth_ev_dense / k_synth_pd take, Dopplers log-uniform over seven decades with both signs, exact
binary steps, steps at the edges of the breakpoint classes, start phases on table-index and chip boundaries, 1 to 16 channels,
blocks whose last tile is partial — and for each replays the fast paths' arithmetic next to the reference's recurrence stepped
sample by sample (gpsbb_modelerr.hip.h), recording the largest realised / budget of everything the kernels test.  Prints one
JSON object: the distribution over the campaign, the ten worst workloads with their parameters, and whether any unflagged
decision differed from the truth or the replay's flag count ever differed from the kernel's own.
   GPSBB_PY_LIB=exp python tools/fuzz_budgets.py --cases 10000 --seed 5 --out profiles/r05_fuzz_budgets.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
try:
    import torch  # noqa: F401
except Exception:
    pass
import model_err  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

RATIOS = ("y0_over_W", "tk_over_W", "x0_over_W", "tc_over_W")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    pkg = load_package()
    rng = np.random.default_rng(a.seed)
    ratios, worst, skipped, bad, flag_mismatch = [], [], 0, 0, 0
    kernels = {}
    with pkg.Synth(0) as synth:
        for case in range(a.cases):
            low = rng.random() < 0.35
            fs = float(rng.choice([2.6e6, 3e6, 4.092e6, 8e6, 10e6, 15e6]) if low else rng.choice([16.368e6, 20e6, 25e6, 30e6, 2.0 ** 25, 50e6, 61.44e6]))
            nch = int(rng.integers(1, 17))
            nb = int(rng.integers(1, 4))
            nsamp = int(rng.choice([rng.integers(1100, 9000), rng.integers(9000, 60000), 1024 * int(rng.integers(2, 40)) + int(rng.integers(0, 2))]))
            ch = pkg.synth_descriptors(nb, nch=nch, seed=int(rng.integers(1, 2 ** 31)))
            fmax = (1.0 / 2100.0 if not low else 0.12) * fs
            kind = rng.random()
            if kind < 0.5:        # log-uniform over seven decades, either sign
                f = 10.0 ** rng.uniform(np.log10(fmax) - 7, np.log10(fmax), (nb, nch)) * np.where(rng.random((nb, nch)) < 0.5, -1.0, 1.0)
            elif kind < 0.7:      # exact binary steps
                f = np.sign(rng.uniform(-1, 1, (nb, nch))) * fs * 2.0 ** rng.integers(-24, -11 if not low else -4, (nb, nch))
            elif kind < 0.85 and not low:  # the edges of the breakpoint classes (1 .. 4 index changes per run of 16 samples)
                kc = rng.integers(1, 5, (nb, nch))
                f = kc / 15.5 / 512.0 * fs * (1.0 + rng.uniform(-2e-4, 2e-4, (nb, nch))) * np.where(rng.random((nb, nch)) < 0.5, -1.0, 1.0)
            else:                 # uniform up to the limit
                f = rng.uniform(-fmax, fmax, (nb, nch))
            ch["f_carr"] = f
            ch["f_code"] = 1.023e6 + f / 1540.0
            if rng.random() < 0.25:
                ch["code_phase"] = np.floor(ch["code_phase"])
            if rng.random() < 0.25:
                ch["carr_phase"] = np.floor(ch["carr_phase"] * 512.0) / 512.0
            chain = bool(rng.random() < 0.4) and nb > 1
            r = model_err.measure(pkg, synth, ch, fs, nsamp, flags=pkg.CHAIN_CARRIER if chain else 0)
            if "skipped" in r:
                skipped += 1
                continue
            kernels[r["kernel"]] = kernels.get(r["kernel"], 0) + 1
            m = max(r["max"][q] for q in RATIOS)
            ratios.append(m)
            bad += r["bad_unflagged_decisions"]
            if r["lanes_flagged"] != r["kernel_exact_runs"]:
                flag_mismatch += 1
            worst.append((m, {"case": case, "fs": fs, "nsamp": nsamp, "nch": nch, "nblocks": nb, "chained": chain, "kernel": r["kernel"],
                              "which": max(RATIOS, key=lambda q: r["max"][q]), "doppler_kind": ["log-uniform", "binary", "class edges", "uniform"][int(kind >= 0.5) + int(kind >= 0.7) + int(kind >= 0.85)],
                              "max_abs_f_carr": float(np.abs(f).max())}))
            worst.sort(key=lambda t: -t[0])
            del worst[10:]
    x = np.asarray(ratios)
    bud = (model_err.C.c_double * 3)()
    pkg.lib().gpsbb_test_budgets(bud)
    doc = {"campaign": {"cases": a.cases, "seed": a.seed, "measured": int(x.size), "not_a_model_kernel": skipped, "kernels": kernels},
           "realised_over_budget": {"max": float(x.max()), "p50": float(np.quantile(x, 0.5)), "p90": float(np.quantile(x, 0.9)), "p99": float(np.quantile(x, 0.99)),
                                    "p999": float(np.quantile(x, 0.999)),
                                    "histogram_edges": [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.75, 1.0, 1e9],
                                    "histogram": np.histogram(x, [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.75, 1.0, 1e9])[0].tolist()},
           "bad_unflagged_decisions": int(bad), "workloads_whose_flag_count_differs_from_the_kernels": flag_mismatch,
           "worst": [dict(w[1], realised_over_budget=w[0]) for w in worst],
           "budgets": {"EV_MODEL_ERR_units": bud[0], "EV_T_EPS_units": bud[1], "PD_BAND_units": bud[2], "unit": "2^-32"}}
    txt = json.dumps(doc, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
