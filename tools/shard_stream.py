#!/usr/bin/env python3
"""BASELINE configs[4] (SURVEY section 8d, M4 "shard-S"), shortened: one long 16-channel 25 MS/s stream cut into
contiguous time shards, one per rank; every rank renders its shard through the streaming ring
(gpsbb_stream_*, carrier chained from the exact seed of its first block), checksums every block as it
arrives in pinned host memory and discards it.  No collective on the data path: ranks exchange only their
block digests at the end (torch.distributed gather of 32-byte hashes), and rank 0 prints the digest of the
digests, which does not depend on the number of ranks.

    python tools/shard_stream.py --seconds 36                       # one GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/shard_stream.py --seconds 36
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stream_descriptors(pkg, nblocks, nch, seed=0x5EED):
    """A stream continuous in time: per channel a slowly drifting Doppler (as a satellite pass gives), phases
    and nav state from the seeded generator; carr_phase of block 0 only (later blocks continue it)."""
    ch = pkg.synth_descriptors(nblocks, nch=nch, seed=seed)
    f0 = ch["f_carr"][0].copy()
    drift = (pkg.SplitMix64(seed ^ 0xD1F7).u01((nch,)) - 0.5) * 1.2          # Hz per block, both signs
    ch["f_carr"] = f0[None, :] + np.arange(nblocks)[:, None] * drift[None, :]
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    return ch


def shard_digests(pkg, synth, ch_all, seeds_all, b0, b1, delt, nsamp, bps, depth=3):
    """SHA-256 of every block of [b0, b1) rendered through the ring; returns (digests, seconds)."""
    nch = ch_all.shape[1]
    mine = ch_all[b0:b1].copy()
    mine["carr_phase"][0] = seeds_all[b0]          # the shard starts from the stream's exact phase
    nblk = b1 - b0
    pad = (-nblk) % bps
    if pad:                                         # the last slot is padded with blocks nobody looks at
        mine = np.concatenate([mine, np.repeat(mine[-1:], pad, axis=0)])
    nslots = mine.shape[0] // bps
    st = synth.stream(nch, delt, nsamp, bps, depth=depth, flags=pkg.CHAIN_CARRIER)
    out = []
    t0 = time.perf_counter()
    pushed = popped = 0
    while popped < nslots:
        while pushed < nslots and st.pending < depth:
            st.push(mine[pushed * bps:(pushed + 1) * bps])
            pushed += 1
        iq, _ = st.pop(copy=False)
        for k in range(bps):
            if popped * bps + k < nblk:
                out.append(hashlib.sha256(iq[k].tobytes()).digest())
        popped += 1
    dt = time.perf_counter() - t0
    st.close()
    return out, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=36.0, help="stream length (3600 in BASELINE configs[4])")
    ap.add_argument("--fs", type=float, default=25e6)
    ap.add_argument("--nch", type=int, default=16)
    ap.add_argument("--bps", type=int, default=16, help="blocks per ring slot")
    a = ap.parse_args()
    import torch  # first: the HIP runtime
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("GPSBB_BENCH_BACKEND", "gloo"))
    nsamp = int(round(a.fs / 10))
    nblocks = int(round(a.seconds * 10))
    delt = 1.0 / a.fs
    ch = stream_descriptors(pkg, nblocks, a.nch)
    seeds = pkg.chain_carrier_host(ch, delt, nsamp)      # exact phase at the start of every block (host threads)
    b0, b1 = pkg.shard_blocks(nblocks, rank, world)
    with pkg.Synth(local) as s:
        dig, dt = shard_digests(pkg, s, ch, seeds, b0, b1, delt, nsamp, a.bps)
    mine = b"".join(dig)
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (mine, dt))
        allbytes = b"".join(p[0] for p in parts)
        dt = max(p[1] for p in parts)
        dist.barrier()
        dist.destroy_process_group()
    else:
        allbytes = mine
    if rank == 0:
        print(json.dumps({"config": "shard-S: %d ch, fs %.3g, %d blocks of %d samples, %d ranks, %d-block slots"
                          % (a.nch, a.fs, nblocks, nsamp, world, a.bps),
                          "stream_digest": hashlib.sha256(allbytes).hexdigest(),
                          "iq_samples_per_s_to_host": nblocks * nsamp / dt, "seconds": dt}))


if __name__ == "__main__":
    main()
