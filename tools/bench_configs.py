#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs other than the headline one, end to end from the RINEX file:
host front end (libgpsfe) -> exact carrier seeds on the host -> batches on one MI355X.
Writes one JSON object; quoted in DESIGN.md (the headline config is bench.py's)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SITE = (30.286502, 120.032669, 100.0)


def run(pkg, synth, name, nav, motion, max_chan, fs, nsamp, nblocks, batch_blocks):
    t0 = time.perf_counter()
    fe = pkg.FrontEnd(os.path.join(GOLD, nav), llh=SITE, motion=os.path.join(GOLD, motion) if motion else None,
                      max_chan=max_chan)
    ch = fe.generate(nblocks)
    fe.close()
    t_fe = time.perf_counter() - t0
    t0 = time.perf_counter()
    ch["carr_phase"] = pkg.chain_carrier_host(ch, 1.0 / fs, nsamp, 16)
    ch["carr_phase"][ch["prn"] <= 0] = 0.0
    t_chain = time.perf_counter() - t0
    batches = [synth.batch(ch[k:k + batch_blocks], 1.0 / fs, nsamp) for k in range(0, nblocks, batch_blocks)]
    for b in batches:  # every batch once before the clock starts: its device buffers are allocated on first use
        b.run()
        b.run()
    synth.sync()
    t0 = time.perf_counter()
    for b in batches:
        b.run()
    synth.sync()
    dt = time.perf_counter() - t0
    for b in batches:
        b.close()
    return {"config": name, "channels": int((ch["prn"][0] > 0).sum()), "fs": fs, "nsamp_per_block": nsamp,
            "blocks": nblocks, "signal_seconds": nblocks * 0.1, "front_end_s": t_fe, "host_carrier_chain_s": t_chain,
            "gpu_s": dt, "iq_samples_per_s": nblocks * nsamp / dt, "x_realtime": nblocks * 0.1 / dt}


def main():
    pkg = load_package()
    pkg.build_frontend()
    out = []
    with pkg.Synth(0) as s:
        out.append(run(pkg, s, "1/2 static, 2.6 MS/s, reference block (300000 samples)", "synth3540.14n", None, 12,
                       2.6e6, 300000, 3000, 1000))
        out.append(run(pkg, s, "4 user motion (10 Hz), 2.6 MS/s", "synth3540.14n", "circle_motion.csv", 12, 2.6e6,
                       300000, 3000, 1000))
        out.append(run(pkg, s, "3 geometry through the front end: 16 ch, 25 MS/s, 2.5 M-sample blocks", "dense3540.14n",
                       None, 16, 25e6, 2500000, 400, 200))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
