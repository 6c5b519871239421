#!/usr/bin/env python3
"""Resident re-runs with the fixed-point carrier (GPSBB_FIXED_CARRIER, the reference built without FLOAT_CARR_PHASE): the
headline geometry and the reference's own.   python tools/fixed_rate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
import bench
with pkg.Synth(0) as s:
    for name, nb, nch, fs, nsamp in (("16 ch 25 MS/s", 400, 16, 25e6, 2500000), ("12 ch 2.6 MS/s", 1000, 12, 2.6e6, 300000)):
        ch = pkg.synth_descriptors(nb, nch=nch, seed=0xF1ED)
        ch["carr_phase"] = np.floor(ch["carr_phase"] * 2.0 ** 32)
        for flags, what in ((pkg.FIXED_CARRIER, "fixed-point carrier"), (0, "IEEE carrier")):
            if not flags:
                ch["carr_phase"] = ch["carr_phase"] / 2.0 ** 32
            r, _ = bench.resident_leg(pkg, s, torch, ch, 1.0 / fs, nsamp, flags, 10, 3, "cuda:0")
            print("%-16s %-20s %.4g S/s  step %.3f ms synth %.3f prepass %.3f kernel %d" % (name, what, r["value"], r["ms_per_step"], r["synth_kernel_ms"], r["prepass_ms"], s.info(pkg.INFO_LAST_KERNEL)))
