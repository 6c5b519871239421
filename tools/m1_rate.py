import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from __graft_entry__ import load_package
pkg = load_package()
import bench
mch = pkg.synth_descriptors(1000, nch=12, seed=0xF00D)
with pkg.Synth(0) as s:
    for flags in (0, pkg.CHAIN_CARRIER):
        r, _ = bench.resident_leg(pkg, s, torch, mch, 1.0 / 2.6e6, 300000, flags, 20, 4, "cuda:0")
        print("flags", flags, "%.4g S/s  step %.3f ms synth %.3f prepass %.3f" % (r["value"], r["ms_per_step"], r["synth_kernel_ms"], r["prepass_ms"]))
