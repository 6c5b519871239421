#!/usr/bin/env python3
"""What the seed of a time shard costs: gpsbb_chain_carrier (device) over the first B blocks of bench.py's stream, next to
gpsbb_chain_carrier_host (16 host threads) over the same blocks; both give the exact carr_phase at block B.
   python tools/seed_rate.py [--blocks 56000] [--host]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=56000)
    ap.add_argument("--host", action="store_true", help="also time the host chain (seconds at this size)")
    a = ap.parse_args()
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    import bench
    pkg = load_package()
    nch, delt, nsamp = 16, 1 / 25e6, 2500000
    ch = bench.stream_descriptors(pkg, a.blocks + 1, nch)
    with pkg.Synth(0) as s:
        s.shard_seed(ch, 64, delt, nsamp)
        for b0 in (400, 3200, 8000, a.blocks):
            if b0 > a.blocks:
                continue
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                seed = s.shard_seed(ch, b0, delt, nsamp)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            line = "blocks %6d  device chain %.4f s  (%.3g samples/s of stream)" % (b0, best, b0 * nsamp / best)
            if a.host and b0 <= a.blocks:
                t0 = time.perf_counter()
                want = pkg.chain_carrier_host(ch[:b0 + 1], delt, nsamp)[b0]
                line += "   host chain %.3f s   equal %s" % (time.perf_counter() - t0, want.tobytes() == seed.tobytes())
            print(line, flush=True)
        print("chain fallbacks", s.info(pkg.INFO_CHAIN_FALLBACKS))


if __name__ == "__main__":
    main()
