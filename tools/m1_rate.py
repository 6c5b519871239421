#!/usr/bin/env python3
"""Resident re-runs of the reference-faithful geometry (12 ch, 2.6 MS/s, 300 000-sample blocks, 1000
blocks per step), independent blocks and chained on the device.   python tools/m1_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
import bench
mch = pkg.synth_descriptors(1000, nch=12, seed=0xF00D)
with pkg.Synth(0) as s:
    for flags in (0, pkg.CHAIN_CARRIER):
        r, _ = bench.resident_leg(pkg, s, torch, mch, 1.0 / 2.6e6, 300000, flags, 20, 4, "cuda:0")
        print("flags", flags, "%.4g S/s  step %.3f ms synth %.3f prepass %.3f" % (r["value"], r["ms_per_step"], r["synth_kernel_ms"], r["prepass_ms"]))
    r, _ = bench.resident_leg(pkg, s, torch, mch, 1.0 / 2.6e6, 300000, 0, 20, 5, "cuda:0", synth_only=True)
    print("synthesis alone %.4g S/s  synth %.3f ms" % (r["value"], r["synth_kernel_ms"]))
