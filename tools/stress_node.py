#!/usr/bin/env python3
"""The reference scenario (RINEX -> front end -> 301 blocks) through the node driver, over and over in one process: 1 shard once
(the reference bytes), then N shards `runs` times; any block that differs is listed with where it differs.
usage: tools/stress_node.py [runs] [nshards] [depth] [reference|dense][-interleaved|-indexed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class Collect:
    def __init__(self, nblocks, nsamp):
        self.iq = np.zeros((nblocks, nsamp, 2), np.int16)
        self.nsamp = nsamp

    def __call__(self, iq_ptr, first, nb, shard):
        src = (C.c_int16 * (nb * self.nsamp * 2)).from_address(iq_ptr)
        self.iq[first:first + nb] = np.frombuffer(src, np.int16).reshape(nb, self.nsamp, 2)
        return 0


def render(pkg, ch, fs, nsamp, nshards, bps, depth, flags=0):
    sink = Collect(ch.shape[0], nsamp)
    with pkg.Node(nshards, ch.shape[1], 1.0 / fs, nsamp, bps, depth=depth, flags=flags, devices=[0] * nshards) as node:
        st = node.run(ch, sink)
    return sink.iq, st


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    nshards = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    mode = sys.argv[4] if len(sys.argv) > 4 else "reference"   # or: dense (16 ch, 25 MS/s: the breakpoint kernel), + "-interleaved"
    pkg = load_package()
    flags = pkg.NODE_INTERLEAVED if mode.endswith("-interleaved") else 0
    if mode.endswith("-indexed"):   # slots as they complete, from several producer threads at once
        flags = pkg.NODE_INDEXED | pkg.NODE_CONCURRENT
    if mode.startswith("dense"):
        fs, nsamp, bps = 25e6, 250000, 4
        ch = pkg.synth_descriptors(101, nch=16, seed=0x5EED)
        ch["f_carr"] = ch["f_carr"][0][None, :] + np.linspace(0.0, 3.0, ch.shape[0])[:, None]   # a stream: Doppler drifting slowly
        ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    else:
        pkg.build_frontend()
        fs, nsamp, bps = 2.6e6, 300000, 8
        fe = pkg.FrontEnd(os.path.join(GOLD, "synth3540.14n"), llh=(30.286502, 120.032669, 100.0), max_chan=12)
        ch = fe.generate(301)
        fe.close()
    ref, _ = render(pkg, ch, fs, nsamp, 1, bps, depth)
    bad = 0
    for r in range(runs):
        iq, st = render(pkg, ch, fs, nsamp, nshards, bps, depth, flags)
        if not (iq == ref).all():
            bad += 1
            blocks = [b for b in range(ch.shape[0]) if not (iq[b] == ref[b]).all()]
            what = []
            for b in blocks[:12]:
                d = np.nonzero((iq[b] != ref[b]).any(axis=1))[0]
                what.append((b, int(d.size), int(d[0]), int(d[-1]), bool((iq[b] == 0).all())))
            print("run %d: %d blocks differ %s; plan %s; (block, samples, first, last, all zero) %s"
                  % (r, len(blocks), blocks[:40], [s["first_block"] for s in st["shards"]], what), flush=True)
    print("%d of %d runs differed (%s, nshards %d, depth %d, lib %s)" % (bad, runs, mode, nshards, depth, os.environ.get("GPSBB_PY_LIB", "product")))


if __name__ == "__main__":
    main()
