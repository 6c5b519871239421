"""The N>1 path without GPUs: world_size-2 gloo processes each take a time shard whose carrier seeds come
from gpsbb_chain_carrier_host, render it with the CPU oracle (standing in for the device here), and the
gathered result must equal the single-stream render.  No data-path collective is involved in the product;
the all_gather below is the test's own check."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_package


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nblocks, nsamp, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    pkg = load_package()
    delt = 1 / 4.092e6
    ch = pkg.synth_descriptors(nblocks, nch=6, seed=4242)
    ch["prn"][nblocks // 2:, 1] = 17           # a re-allocated channel inside rank 1's shard
    mine = pkg.shard_descriptors(ch, rank, world, delt, nsamp)
    b0, b1 = pkg.shard_blocks(nblocks, rank, world)
    assert mine.shape[0] == b1 - b0
    iq, st, _ = ob.Oracle().fill_blocks(mine, delt, nsamp, chain=False)   # every block stands alone
    t = torch.from_numpy(iq.astype(np.int32).reshape(-1))  # gloo has no int16
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    if rank == 0:
        whole, _, _ = ob.Oracle().fill_blocks(ch, delt, nsamp, chain=True)
        got = torch.cat(parts).numpy().reshape(whole.shape)
        ret["equal"] = bool((got == whole).all())
        ret["blocks"] = [list(pkg.shard_blocks(nblocks, r, world)) for r in range(world)]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_time_shards_reproduce_the_stream(pkg):
    world, nblocks, nsamp = 2, 6, 30000
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, nblocks, nsamp, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        assert ret["equal"] is True
        assert ret["blocks"] == [[0, 3], [3, 6]]


def test_shard_ranges_cover_everything(pkg):
    for n in (1, 7, 36000):
        for w in (1, 2, 4, 8):
            r = [pkg.shard_blocks(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
