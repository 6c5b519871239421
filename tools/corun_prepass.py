#!/usr/bin/env python3
"""A neighbour made of the library's own pre-pass kernels, for co-residency experiments (run it in its own process beside
tools/kbench.py --synth-only):
   python tools/corun_prepass.py chain <seconds>   walks + prefix + fix-up only (gpsbb_chain_carrier: no rows, no tile states)
   python tools/corun_prepass.py batch <seconds>   whole chained batches, synthesis included"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from __graft_entry__ import load_package
import bench
pkg = load_package()
mode, seconds = sys.argv[1], float(sys.argv[2])
nch, delt, nsamp = 16, 1 / 25e6, 2500000
ch = bench.stream_descriptors(pkg, 8001, nch)
with pkg.Synth(0) as s:
    s.shard_seed(ch, 64, delt, nsamp)
    print("corunner ready", flush=True)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        if mode == "chain":
            s.shard_seed(ch, 8000, delt, nsamp)
        n += 1
    print("corunner did %d rounds" % n)
