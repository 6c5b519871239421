#!/usr/bin/env python3
"""Latency of the drop-in call gpsbb_fill_block (host buffers in and out) on the reference-faithful block:
12 channels, 2.6 MS/s, 300000 samples = the unit the reference's loop produces every 100 ms."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    z = np.load(os.path.join(ROOT, "tests", "golden", "static_F.npz"))
    desc = z["desc"].view(pkg.CHAN_DTYPE).reshape(z["desc"].shape[0], -1)
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    with pkg.Synth(0) as s:
        for _ in range(3):
            s.fill_block(desc[0], 1 / fs, nsamp)
        ts = []
        for k in range(30):
            t0 = time.perf_counter()
            s.fill_block(desc[k % desc.shape[0]], 1 / fs, nsamp)
            ts.append(time.perf_counter() - t0)
        b = s.batch(desc[:1], 1 / fs, nsamp)
        b.run(); s.sync()
        b.run(); s.sync()
        tm = b.timing()
        b.close()
    ts = np.array(ts) * 1e3
    print(json.dumps({"fill_block_ms_median": float(np.median(ts)), "min": float(ts.min()), "max": float(ts.max()),
                      "realtime_factor": 100.0 / float(np.median(ts)), "kernels_ms": tm}))


if __name__ == "__main__":
    main()
