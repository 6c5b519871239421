#!/usr/bin/env python3
"""When the workgroups of one k_synth_ev launch start, finish staging and leave (library built with -DGPSBB_EV_TIMING:
every workgroup writes its timestamps over the start of the IQ buffer).  python tools/wg_timing.py [blocks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nsamp = 2500000
ch = pkg.synth_descriptors(nb, nch=16, seed=0x5EED)
out = torch.zeros(nb * nsamp * 2, dtype=torch.int16, device="cuda:0")
with pkg.Synth(0) as s:
    b = s.batch(ch, 1 / 25e6, nsamp)
    for _ in range(4):
        b.run(out.data_ptr())
    s.sync()
    s.set_option(pkg.OPT_SKIP_SEED, 1)
    out.zero_()
    torch.cuda.synchronize()
    b.run(out.data_ptr())
    s.sync()
    t = b.timing()
    s.set_option(pkg.OPT_SKIP_SEED, 0)
    b.close()
raw = out[:8192 * 16].cpu().numpy().view(np.uint64).reshape(-1, 4)
# the log shares the buffer with real IQ: keep the rows that look like timestamps (a few ms apart, a sane tile count)
ok = (raw[:, 0] != 0) & (raw[:, 3] < 100000) & (raw[:, 2] >= raw[:, 1]) & (raw[:, 1] >= raw[:, 0]) & (raw[:, 2] - raw[:, 0] < 10 ** 7)
rows = raw[ok]
med = np.median(rows[:, 0])
rows = rows[np.abs(rows[:, 0].astype(np.float64) - med) < 1e7]
t0 = rows[:, 0].min()
tick = 1e-8  # wall_clock64: 100 MHz
ent, stg, ext, tiles = (rows[:, 0] - t0) * tick * 1e3, (rows[:, 1] - t0) * tick * 1e3, (rows[:, 2] - t0) * tick * 1e3, rows[:, 3]
work = tiles > 0
print("kernel %.3f ms by events; %d workgroups logged, %d did work" % (t["ms_synth"], len(rows), work.sum()))
print("entry   ms: first %.3f  median %.3f  last %.3f (workers: last %.3f)" % (ent.min(), np.median(ent), ent.max(), ent[work].max()))
print("staging ms: median %.4f max %.4f" % (np.median((stg - ent)[work]), (stg - ent)[work].max()))
print("exit    ms: first worker %.3f  median %.3f  last %.3f" % (ext[work].min(), np.median(ext[work]), ext[work].max()))
h, edges = np.histogram(ext[work], bins=12)
print("exit histogram:", " ".join("%.2f:%d" % (edges[i], h[i]) for i in range(len(h))))
h, edges = np.histogram(ent[work], bins=12)
print("entry histogram (workers):", " ".join("%.2f:%d" % (edges[i], h[i]) for i in range(len(h))))
busy = np.sum((ext - stg)[work]) / (256 * ext.max())
print("CU occupancy by working workgroups: %.1f %% of 256 CUs x %.3f ms" % (busy * 100, ext.max()))
