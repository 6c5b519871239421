#!/usr/bin/env python3
"""BASELINE configs[4] at its stated length through the product's node driver: 16 channels, 25 MS/s, 3600 s = 36 000 blocks
of 2.5 M samples = 360 GB of int16 IQ, every block delivered into pinned host memory and handed to ONE sink in stream order
(include/gpsbb_node.h), the first and last 64 KiB of every block digested on arrival.  Run once with one shard and once with
`--shards` shards on the same GPU (contiguous and interleaved): the digest of block digests must not depend on the layout.
    python tools/config5_node.py [--blocks 36000] [--shards 4] [--out profiles/r04_config5_node_full_length.json]"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
try:
    import torch  # noqa: F401
except Exception:
    pass
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=36000)
    ap.add_argument("--shards", type=int, default=4)
    ap.add_argument("--slot", type=int, default=32)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--out")
    a = ap.parse_args()
    pkg = load_package()
    nch, fs, nsamp = 16, 25e6, 2500000
    t0 = time.perf_counter()
    ch = bench.stream_descriptors(pkg, a.blocks, nch)
    t_desc = time.perf_counter() - t0
    blk_bytes = nsamp * 4
    res = {"blocks": a.blocks, "bytes": a.blocks * blk_bytes, "descriptor_generation_s": t_desc, "runs": []}
    for nshards, flags, name in ((1, 0, "one shard, ordered sink"), (a.shards, 0, "%d contiguous shards on one GPU, ordered sink" % a.shards),
                                 (a.shards, pkg.NODE_INTERLEAVED, "%d interleaved shards on one GPU, ordered sink" % a.shards)):
        digs = np.zeros(a.blocks, np.uint32)
        order_ok = [True]
        nxt = [0]

        def sink(iq_ptr, first, nb, shard):
            order_ok[0] = order_ok[0] and first == nxt[0]
            nxt[0] = first + nb
            for j in range(nb):
                base = iq_ptr + j * blk_bytes
                head = (C.c_char * (1 << 16)).from_address(base)
                tail = (C.c_char * (1 << 16)).from_address(base + blk_bytes - (1 << 16))
                digs[first + j] = zlib.crc32(tail, zlib.crc32(head))
            return 0
        with pkg.Node(nshards, nch, 1.0 / fs, nsamp, a.slot, depth=a.depth, flags=flags, devices=[0] * nshards) as node:
            st = node.run(ch, sink)
        res["runs"].append({"layout": name, "seconds": st["seconds"], "GBps_into_the_sink": a.blocks * blk_bytes / st["seconds"] / 1e9,
                            "samples_per_s": a.blocks * nsamp / st["seconds"], "in_stream_order": bool(order_ok[0]), "blocks": st["blocks"],
                            "digest_of_block_digests": int(zlib.crc32(digs.tobytes())),
                            "shards": [{k: s[k] for k in ("first_block", "nblocks", "seed_seconds", "busy_seconds", "wait_seconds", "numa_node", "cpus_bound")} for s in st["shards"]]})
    res["layouts_agree"] = len({r["digest_of_block_digests"] for r in res["runs"]}) == 1
    txt = json.dumps(res, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
