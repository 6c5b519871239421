#!/usr/bin/env python3
"""The reference's geometry (12 ch, 2.6 MS/s, 300 000-sample blocks, 1000 independent blocks per step), resident re-runs: what
k_synth_pd takes per launch beside each kind of pre-pass — lap-parallel, row walks, host threads (no pre-pass kernel on the device
at all: the tile states arrive by DMA) — and with the GPU to itself.   python tools/m1_where.py [nblocks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
import bench
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mch = pkg.synth_descriptors(nb, nch=12, seed=0xF00D)
with pkg.Synth(0) as s:
    for where, name in ((3, "lap-parallel"), (1, "row walks"), (2, "host threads")):
        s.set_option(pkg.OPT_SEED_WHERE, where)
        r, _ = bench.resident_leg(pkg, s, torch, mch, 1.0 / 2.6e6, 300000, 0, 20, 4, "cuda:0")
        print("%-14s %.4g S/s  step %.3f ms  k_synth_pd %.3f ms  pre-pass %.3f ms (pre-pass taken %d)" %
              (name, r["value"], r["ms_per_step"], r["synth_kernel_ms"], r["prepass_ms"], s.info(pkg.INFO_PREPASS)), flush=True)
    s.set_option(pkg.OPT_SEED_WHERE, 0)
    r, _ = bench.resident_leg(pkg, s, torch, mch, 1.0 / 2.6e6, 300000, 0, 20, 5, "cuda:0", synth_only=True)
    print("synthesis alone %.4g S/s  k_synth_pd %.3f ms" % (r["value"], r["synth_kernel_ms"]))
