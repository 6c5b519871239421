#!/usr/bin/env python3
"""bench.py — the GPS L1 C/A IQ buffer-fill path on MI355X, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

Workload (config.workload = "synth16-S as one stream", BASELINE configs[2] in the shape of configs[4]): ONE
time-continuous 16-channel stream at fs = 25 MS/s in 0.1 s blocks of 2.5 M samples — seeded descriptors
 with a slowly varying Doppler per channel, |f_carr| <= 5 kHz — whose carrier is chained
exactly from block to block (GPSBB_CHAIN_CARRIER, c:2741-2746 never re-seeds carr_phase).  One "step" is one
pass of the hot path over the next 3200 blocks of that stream (8e9 IQ samples, 32 GB of int16 IQ), from
descriptors the library plans afresh: per 400-block push the host validates and plans the descriptors and uploads
them, the device builds the exact NCO states of every tile with the lap-parallel pre-pass (k_lap_plan, k_lap_pass1, k_lap_scan,
k_lap_pass2, k_lap_repair for the code chains and for the carriers, the carrier chained exactly across the blocks and from the
phase the push before left on the device) and synthesises (k_synth_ev); the IQ lands in the ring's HBM slots
(GPSBB_STREAM_DEVICE_ONLY).  The stream has T = K*3200 blocks; the warm-up and every one of the --repeats
timed regions walk through its pushes cyclically (a region of K steps is one pass over a rank's whole shard).  With
N ranks the stream is cut into N contiguous time shards (rank r renders blocks [r*T/N, (r+1)*T/N): strong scaling,
no data-path collective); a shard starts from the stream's exact carrier phase there, which every rank computes for
itself on its GPU (gpsbb_chain_carrier over the blocks before its shard: `shard_seed_s` = the slowest rank's,
`value_incl_seed` charges it to one pass over the stream); torch.distributed (RCCL) carries the barrier, the
max-over-ranks times and the digests only.

After the timed regions the same kind of ring (GPSBB_STREAM_DEVICE_ONLY) renders the rank's shard once more in
order and blocks of it are read back from HBM and compared with the CPU oracle, bit for bit (`parity`): the leading
blocks of the shard (oracle chained from the shard's seed) and one block of its last push (oracle from the phase
the stream itself reports there).  A mismatch on any rank makes the run exit non-zero.

Prints ONE JSON line on rank 0.  `value` = IQ samples/s over all GPUs, from the MEDIAN of --repeats timed regions
of K steps each, every region bracketed by barrier + synchronize on both sides with the ring drained (min / max
beside it).  `roofline` = algorithmic HBM bytes (4 B per IQ sample) of the synthesis kernel / its HIP-event
duration against the 8 TB/s HBM peak.  Beside it, measured in the same invocation:
  resident      re-runs of one 400-block batch whose descriptors and plans stay in HBM (independent blocks and
                chained): what round 1 reported as its value
  gather        the same shard through a ring with the pinned D2H gather (PCIe-inclusive; never `value`)
  m1            BASELINE.md section 3's other leg, the reference's own geometry: 12 ch, 2.6 MS/s, 300 000-sample blocks
                (plutogpssim.c:43-45).  `m1.stream` (and `m1.value`, `m1.roofline_stream`) AT EVERY N: a time-sharded stream of
                fresh chained pushes like the headline's (1000 blocks per push), every block cross-checked against the per-sample
                kernel by device-side digests, leading blocks against the oracle; `m1.gpu*` (resident re-runs of one batch) at N = 1
  fill_block    the drop-in call itself, gpsbb_fill_block_ref on the reference's channel_t layout with a pageable iq_buff: median /
                p99 of 200 calls for the reference's block and for the headline's, each checked against the oracle once (rank 0)
  cpu_baseline  the CPU restatement (oracle, 1 core) on a bounded sample of the same blocks, AT EVERY N: rank 0 starts it as a
                process of its own (bench.py --cpu-legs) once the timed regions are over; the GPU legs that follow run meanwhile
`resident`, `m1.gpu*`, `roofline.alone` and the write ceiling run at N = 1 only (with N > 1 they would keep N - 1 GPUs idle behind
rank 0 for most of the command).
  parity        blocks of the timed mode's ring against the CPU oracle (a handful: the oracle renders 2e7 samples/s) AND every
                block of every shard against the per-sample kernel's rendering, by device-side digests (`blocks_cross_checked`)
  node_driver   the product's one-process driver (include/gpsbb_node.h): the same pushes through one shard on this rank's GPU,
                and — rank 0, after the timed regions — the whole stream over ALL visible GPUs, contiguous shards into an indexed
                sink and interleaved slots into the ordered one, every slot digested and compared with what the ranks rendered
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time
import zlib

import numpy as np

# The library keeps up to nine HIP streams busy (four pre-pass streams, upload, two synthesis streams, gather, the null stream); the
# runtime's default of four hardware queues would make unrelated streams share a queue and run one after the other.
# Must be set before the HIP runtime starts (import torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PUSH_BLOCKS = 400      # blocks per push (1e9 samples, 4 GB of IQ)
STEP_PUSHES = 8        # pushes per step over all ranks: a step is 3200 blocks
# BASELINE.md section 3's other geometry, the reference's own (plutogpssim.c:43-45; README.md:123): 12 channels (MAX_CHAN, h:21),
# 2.6 MS/s, NUM_SAMPLES 300 000 per block; as a stream: pushes of 1000 blocks (3e8 samples, 1.2 GB of IQ), 8 per step
M1_NCH, M1_FS, M1_NSAMP, M1_PUSH_BLOCKS, M1_SEED, M1_DESC_STEPS = 12, 2.6e6, 300000, 1000, 0xF00D, 4


def stream_descriptors(pkg, nblocks, nch, seed=0x5EED, max_doppler=5000.0, first=0, count=None, fields=None):
    """A stream continuous in time: everything from the seeded generator (M2) except the Doppler, which moves
    slowly and smoothly per channel like a satellite pass (period 1-2 h, both signs, through zero).
    first / count: blocks [first, first+count) of the nblocks-block stream only (the same rows: the generator is
    counter-based), so that a rank builds what it renders; fields: see synth_descriptors (the blocks BEFORE a shard are only
    chained through: prn, f_carr, carr_phase)."""
    count = nblocks - first if count is None else count
    ch = pkg.synth_descriptors(nblocks, nch=nch, seed=seed, max_doppler=max_doppler, first=first, count=count, fields=fields)
    g = pkg.SplitMix64(seed ^ 0xD0BB1E5)
    ph = g.u01((nch,)) * 2.0 * np.pi
    per = 36000.0 * (1.0 + g.u01((nch,)))
    amp = max_doppler * (0.35 + 0.65 * g.u01((nch,)))
    b = np.arange(first, first + count, dtype=np.float64)[:, None]
    ch["f_carr"] = amp[None, :] * np.sin(ph[None, :] + 2.0 * np.pi * b / per[None, :])
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    return ch


def parity_check(pkg, ob, synth, mine, delt, nsamp, PB, nch, ncheck, nspot=8):
    """Bytes of evidence on the mode the headline is timed in: the shard through a fresh GPSBB_STREAM_DEVICE_ONLY ring in
    order; the first `ncheck` blocks, one block of every (npush / nspot)-th push and one of the last push read back from the slots' HBM and compared with
    the oracle (c:2690-2756 restated), IQ and end-of-block carrier phase, bit for bit (ob None: no oracle, only the
    digests).  Also returns one 32-bit digest per block of its end-of-block NCO states: the concatenation over the
    ranks must not depend on how many ranks the stream was cut into (every shard starts from its seed)."""
    orc = ob.Oracle() if ob else None
    npush = mine.shape[0] // PB
    spot_every = max(1, npush // max(1, nspot))  # one block of every spot_every-th push, from the phase the stream reports there
    st = synth.stream(nch, delt, nsamp, PB, depth=3, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
    checked, bad, digs = 0, [], []
    iq_digs = np.zeros(npush * PB, np.uint64)  # gpsbb_device_digest of every block as the timed mode renders it
    pushed = 0
    for k in range(npush):
        while pushed < npush and st.pending < 3:
            st.push(mine[pushed * PB:(pushed + 1) * PB])
            pushed += 1
        dptr, ends = st.pop(copy=False)
        iq_digs[k * PB:(k + 1) * PB] = synth.device_digest(dptr, PB, nsamp)
        act = mine["prn"][k * PB:(k + 1) * PB] > 0
        cp = np.where(act, ends["carr_phase"], 0.0)
        xp = np.where(act, ends["code_phase"], 0.0)
        digs.extend(zlib.crc32(xp[j].tobytes(), zlib.crc32(cp[j].tobytes())) for j in range(PB))
        if orc is None:
            continue
        if k == 0:
            want_iq, want_st, _ = orc.fill_blocks(mine[:ncheck], delt, nsamp, chain=True)
            got = synth.device_read(dptr, (ncheck, nsamp, 2))
            for j in range(ncheck):
                ok_iq = bool((got[j] == want_iq[j]).all())
                ok_ph = ends["carr_phase"][j].tobytes() == want_st["carr_phase"][j].tobytes()
                checked += 1
                if not (ok_iq and ok_ph):
                    bad.append(j)
                    sys.stderr.write("bench.py: block %d of the shard: IQ %s (%d samples differ), end-of-block carr_phase %s\n" %
                                     (j, "equal" if ok_iq else "DIFFERS", int((got[j] != want_iq[j]).any(axis=1).sum()),
                                      "equal" if ok_ph else "DIFFERS"))
        if npush > 1 and (k == npush - 1 or (k % spot_every == spot_every - 1 and k < npush - 1)):
            j = (PB // 2 + 7 * k) % PB or 1  # a different block of the push every time, never its first
            one = mine[k * PB + j:k * PB + j + 1].copy()
            cont = (one["prn"][0] > 0) & (one["prn"][0] == mine["prn"][k * PB + j - 1])
            one["carr_phase"][0] = np.where(cont, ends["carr_phase"][j - 1], one["carr_phase"][0])
            want_iq, want_st, _ = orc.fill_blocks(one, delt, nsamp)
            got = synth.device_read(dptr + j * nsamp * 4, (nsamp, 2))
            ok = (got == want_iq[0]).all() and ends["carr_phase"][j].tobytes() == want_st["carr_phase"][0].tobytes()
            checked += 1
            if not ok:
                bad.append(k * PB + j)
    st.close()
    return checked, bad, digs, iq_digs


def cross_check(pkg, synth, mine, delt, nsamp, PB, nch, iq_digs):
    """EVERY block of the shard, not the handful the oracle has time for: the shard once more through the same kind of ring
    with the per-sample kernel forced (GPSBB_OPT_SYNTH_KERNEL 1: k_synth steps both NCOs of every sample with genuine IEEE
    adds from rows built by the row walks, k_seed — another algorithm AND another pre-pass than the timed mode's model
    kernels on the lap-parallel pre-pass, itself tested against the oracle), one device-side digest per block
    (gpsbb_device_digest) compared with the timed mode's.  Returns the blocks whose digests differ."""
    npush = mine.shape[0] // PB
    synth.set_option(pkg.OPT_SYNTH_KERNEL, 1)
    try:
        st = synth.stream(nch, delt, nsamp, PB, depth=3, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
        other = np.zeros(npush * PB, np.uint64)
        pushed = 0
        for k in range(npush):
            while pushed < npush and st.pending < 3:
                st.push(mine[pushed * PB:(pushed + 1) * PB])
                pushed += 1
            dptr, _ = st.pop(copy=False)
            other[k * PB:(k + 1) * PB] = synth.device_digest(dptr, PB, nsamp)
        kernel = synth.info(pkg.INFO_LAST_KERNEL)
        st.close()
    finally:
        synth.set_option(pkg.OPT_SYNTH_KERNEL, 0)
    return np.nonzero(other != iq_digs)[0].tolist(), kernel


def cpu_baseline(ob, ch, delt, nsamp, budget_s=10.0):
    """The oracle (kind "port": bit-identical CPU restatement of plutogpssim.c:2690-2756, gcc -O2
    -ffp-contract=off, 1 thread) on as many leading blocks of the stream as fit the time budget."""
    orc = ob.Oracle()
    orc.fill_blocks(ch[:1], delt, min(nsamp, 1000))  # one-time table/code generation out of the timing
    t0 = time.perf_counter()
    orc.fill_blocks(ch[:1], delt, nsamp)
    per_block = time.perf_counter() - t0
    nb = int(max(1, min(ch.shape[0], budget_s / max(per_block, 1e-9))))
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        orc.fill_blocks(ch[:nb], delt, nsamp)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out = {"value": nb * nsamp / best, "unit": "IQ samples/s", "cores": 1, "kind": "port",
           "sample": "%d of the stream's blocks (%d ch x %d samples each), best of 2, gcc -O2 -ffp-contract=off" %
                     (nb, ch.shape[1], nsamp)}
    if ob.have_ref():
        # the reference's own loop statements, built with its Makefile's flags (-O0), on a smaller sample
        ref = ob.RefLoop()
        t0 = time.perf_counter()
        ref.fill(ch[0], delt, nsamp)
        out["reference_loop_O0"] = {"value": nsamp / (time.perf_counter() - t0), "unit": "IQ samples/s",
                                    "cores": 1, "sample": "1 block, verbatim plutogpssim.c:2690-2756, -std=c11 -O0"}
    return out


def cpu_legs_main(args):
    """`bench.py --cpu-legs`: the two CPU baselines in a process of their own (no GPU, no torch) — rank 0 starts it once the timed
    regions are over and reads its one JSON line at the end, so that the oracle's gigabyte of output and its core are not inside the
    process whose threads drive the GPU legs (as a thread of rank 0 it cost the node driver's legs 10 - 30 %: one address space,
    one allocator, one interpreter lock)."""
    from __graft_entry__ import load_package
    pkg = load_package()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_binding as ob
    delt, nsamp, nch = 1.0 / args.fs, args.nsamp, args.nch
    total = args.steps * STEP_PUSHES * args.push_blocks
    n_main = int(max(1, min(total, 4 + args.cpu_budget * 3.0e7 / max(nsamp, 1))))   # more than the budget can take at 2.3e7 samples/s
    mine = stream_descriptors(pkg, total, nch, first=0, count=n_main)
    m_total = max(1, min(args.steps, M1_DESC_STEPS)) * STEP_PUSHES * args.m1_push_blocks
    n_m1 = int(max(1, min(m_total, 4 + args.cpu_budget / 2 * 4.0e7 / max(args.m1_nsamp, 1))))
    m1_mine = stream_descriptors(pkg, m_total, M1_NCH, seed=M1_SEED, first=0, count=n_m1)
    out = {}
    try:
        out["cpu_baseline"] = cpu_baseline(ob, mine, delt, nsamp, budget_s=args.cpu_budget)
        out["m1_cpu"] = cpu_baseline(ob, m1_mine, 1.0 / M1_FS, args.m1_nsamp, budget_s=args.cpu_budget / 2)
    except Exception as e:
        out["error"] = repr(e)
    print(json.dumps(out))


def block_digest(a):
    """32-bit digest of the first and the last 64 KiB of one block as it sits in host memory: proof of arrival, cheap
    enough not to bound the gather (at ~2 GB/s of CRC a longer prefix would: full-length digests are the tests' business)."""
    m = memoryview(a).cast("B")
    return zlib.crc32(m[-(1 << 16):], zlib.crc32(m[:1 << 16]))


def resident_leg(pkg, synth, torch, ch, delt, nsamp, flags, steps, warmup, dev, synth_only=False):
    """round 1's measurement: one batch resident in HBM, run again and again; synth_only: the pre-pass is skipped once
    every table set has been built (GPSBB_OPT_SKIP_SEED), i.e. the synthesis kernel with the GPU to itself"""
    out = torch.empty(ch.shape[0] * nsamp * 2, dtype=torch.int16, device=dev)
    batch = synth.batch(ch, delt, nsamp, flags=flags)
    for _ in range(warmup):
        batch.run(out.data_ptr())
    synth.sync()
    if synth_only:
        synth.set_option(pkg.OPT_SKIP_SEED, 1)
    torch.cuda.synchronize()
    batch.timing_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run(out.data_ptr())
    synth.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = batch.timing_stats(reset=True)
    if synth_only:
        synth.set_option(pkg.OPT_SKIP_SEED, 0)
    ceil_ms = synth.fill_ceiling(out.data_ptr(), out.numel() * 2, iters=10)
    batch.close()
    del out
    n = ch.shape[0] * nsamp
    return {"value": n * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
            "synth_kernel_ms": st["ms_synth_sum"] / max(st["runs"], 1), "prepass_ms": st["ms_seed_sum"] / max(st["runs"], 1),
            "chained": bool(flags & pkg.CHAIN_CARRIER)}, n * 4 / (ceil_ms * 1e-3) / 1e9


def m1_stream_leg(pkg, synth, torch, dist, use_dist, backend, dev, rank, world, ob, K, W, R, depth, PB, nsamp, parity_blocks, parity_spots):
    """The reference's geometry AT EVERY N (north_star: throughput at 2.6 MS/s and 25 MS/s at 1, 2, 4 and 8 GPUs): one
    time-continuous 12-channel stream at 2.6 MS/s in 300 000-sample blocks, cut into `world` contiguous time shards like the
    headline's, every rank seeding its shard on its own GPU; K steps of 8 / world fresh chained pushes of PB blocks per timed
    region (the descriptors of min(K, M1_DESC_STEPS) steps, walked through cyclically), R regions, max over ranks, median.
    Then the shard once more in order: leading blocks and spots against the oracle (ob; None: skipped), every block's digest
    against the per-sample kernel's.  Returns (result dict for rank 0, this rank's shard descriptors, mismatches)."""
    nch, delt = M1_NCH, 1.0 / M1_FS
    ppr = STEP_PUSHES // world
    kd = max(1, min(K, M1_DESC_STEPS))
    total = kd * STEP_PUSHES * PB
    b0, b1 = pkg.shard_blocks(total, rank, world)
    mine = stream_descriptors(pkg, total, nch, seed=M1_SEED, first=b0, count=b1 - b0)
    before = np.concatenate([stream_descriptors(pkg, total, nch, seed=M1_SEED, first=0, count=b0, fields=("prn", "f_carr", "carr_phase")), mine[:1]])
    synth.shard_seed(before, min(b0, 64), delt, nsamp)
    synth.sync()
    t_seed = time.perf_counter()
    mine["carr_phase"][0] = synth.shard_seed(before, b0, delt, nsamp)
    t_seed = time.perf_counter() - t_seed
    del before
    npush = mine.shape[0] // PB

    def barrier():
        synth.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def over_ranks(x, op):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())

    def run_ring(st, first, count):
        for j in range(count):
            if st.pending >= depth:
                st.pop(copy=False)
            k = (first + j) % npush
            st.push(mine[k * PB:(k + 1) * PB])
        while st.pending:
            st.pop(copy=False)

    st = synth.stream(nch, delt, nsamp, PB, depth=depth, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
    run_ring(st, 0, max(W, 1) * ppr)
    pos = max(W, 1) * ppr
    barrier()
    st.timing_stats(reset=True)
    times = []
    for _ in range(R):
        barrier()
        t0 = time.perf_counter()
        run_ring(st, pos, K * ppr)
        synth.sync()
        torch.cuda.synchronize()
        times.append(over_ranks(time.perf_counter() - t0, dist.ReduceOp.MAX if use_dist else None))
        pos += K * ppr
    barrier()
    stats = st.timing_stats(reset=True)
    kernel = {1: "k_synth", 2: "k_synth_pd"}.get(synth.info(pkg.INFO_LAST_KERNEL), "?")
    prepass = synth.info(pkg.INFO_PREPASS)
    st.close()
    # bytes of evidence: the shard in order through a fresh ring, against the oracle and against the per-sample kernel
    n_ok, bad, digs, iq_digs = parity_check(pkg, ob, synth, mine, delt, nsamp, PB, nch, max(1, min(parity_blocks, PB)), parity_spots)
    cross_bad, cross_kernel = cross_check(pkg, synth, mine, delt, nsamp, PB, nch, iq_digs)
    n_all = int(over_ranks(float(n_ok), dist.ReduceOp.SUM if use_dist else None))
    n_bad = int(over_ranks(float(len(bad)), dist.ReduceOp.SUM if use_dist else None))
    n_cross_bad = int(over_ranks(float(len(cross_bad)), dist.ReduceOp.SUM if use_dist else None))
    seed_max = over_ranks(t_seed, dist.ReduceOp.MAX if use_dist else None)
    if use_dist:
        allg = [None] * world
        dist.all_gather_object(allg, (digs, iq_digs))
        digs = [d for part in allg for d in part[0]]
        iq_digs = np.concatenate([part[1] for part in allg])
    if bad or cross_bad:
        sys.stderr.write("bench.py: rank %d, 2.6 MS/s stream: %d blocks differ from the oracle, %d from the per-sample kernel\n" % (rank, len(bad), len(cross_bad)))
    elapsed = statistics.median(times)
    samples_per_step = STEP_PUSHES * PB * nsamp
    ms_synth = stats["ms_synth_sum"] / max(stats["runs"], 1)
    achieved = 4.0 * PB * nsamp / (ms_synth * 1e-3) / 1e9
    res = {"value": samples_per_step * K / elapsed, "unit": "IQ samples/s", "ms_per_step": elapsed / K * 1e3, "steps": K, "n_gpus": world,
           "seconds": times, "synth_kernel_ms": ms_synth, "prepass_ms": stats["ms_seed_sum"] / max(stats["runs"], 1),
           "synthesis_kernel": kernel, "prepass": {3: "lap-parallel", 2: "host threads", 1: "row walks"}.get(prepass, "?"),
           "shard_seed_s": seed_max, "value_incl_seed": samples_per_step * K / (elapsed + seed_max),
           "workload": "%d ch, fs %.3g S/s, %d-sample blocks as ONE chained stream: %d blocks per step (%d fresh pushes of %d), descriptors of %d "
                       "blocks walked through cyclically, cut into %d contiguous time shards" % (nch, M1_FS, nsamp, STEP_PUSHES * PB, STEP_PUSHES, PB, total, world),
           "roofline": {"bound": "valu+lds issue (the package's power limit: DESIGN.md 3.1)", "priced_against": "hbm", "kernel": kernel, "ms_per_launch": ms_synth,
                        "launches_timed": stats["runs"], "algorithmic_bytes_per_launch": 4 * PB * nsamp, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS},
           "parity": {"checked_blocks": n_all, "mismatching_blocks": n_bad, "blocks_cross_checked": int(iq_digs.shape[0]),
                      "cross_mismatching_blocks": n_cross_bad, "blocks_digested": len(digs),
                      "cross_check_against": "k_synth on k_seed's rows (GPSBB_OPT_SYNTH_KERNEL 1)" if cross_kernel == 1 else "?",
                      "stream_iq_digest": zlib.crc32(iq_digs.tobytes()), "stream_end_state_digest": zlib.crc32(np.asarray(digs, np.uint32).tobytes())}}
    return res, mine, n_bad + n_cross_bad


def fill_block_leg(pkg, synth, ob, cases, calls=200):
    """The drop-in call itself (plutogpssim.c:2689-2759 replaced by ONE call): gpsbb_fill_block_ref on the reference's own
    channel_t[] / gain[] (offsetof layout), IQ into a pageable iq_buff, state updated in place — median / p99 / min of `calls`
    calls, every case checked against the oracle once (IQ and the updated channel state)."""
    out = {}
    for name, nch, fs, nsamp, seed in cases:
        d = pkg.synth_descriptors(8, nch=nch, seed=seed)
        iq = np.zeros((nsamp, 2), np.int16)           # pageable, like the reference's iq_buff (c:84)
        lay = pkg.ref_layout()
        for k in range(4):
            chan, gain = pkg.ref_channels(d[k % 8])
            synth.fill_block_ref(chan, gain, 1.0 / fs, nsamp, iq, lay)
        ok = None
        if ob is not None:
            chan, gain = pkg.ref_channels(d[3])
            synth.fill_block_ref(chan, gain, 1.0 / fs, nsamp, iq, lay)
            want_iq, want_st, _ = ob.Oracle().fill_blocks(d[3:4], 1.0 / fs, nsamp)
            act = d["prn"][3] > 0
            ok = bool((iq == want_iq[0]).all()) and all(chan[f][act].tobytes() == want_st[f][0][act].astype(chan[f].dtype).tobytes()
                                                         for f in ("carr_phase", "code_phase", "iword", "ibit", "icode", "dataBit", "codeCA"))
        def timed():
            ts = []
            for k in range(calls):
                chan, gain = pkg.ref_channels(d[k % 8])
                t0 = time.perf_counter()
                synth.fill_block_ref(chan, gain, 1.0 / fs, nsamp, iq, lay)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            return {"median_ms": ts[len(ts) // 2] * 1e3, "p99_ms": ts[min(len(ts) - 1, int(len(ts) * 0.99))] * 1e3, "min_ms": ts[0] * 1e3,
                    "samples_per_s_at_median": nsamp / ts[len(ts) // 2], "real_time_factor": (nsamp / fs) / ts[len(ts) // 2]}
        out[name] = dict(timed(), calls=calls, nch=nch, fs=fs, nsamp=nsamp, equals_oracle=ok,
                         prepass={3: "lap-parallel, on the device", 2: "host threads", 1: "row walks, on the device"}.get(synth.info(pkg.INFO_PREPASS), "?"))
        # ... and with iq_buff registered once (gpsbb_host_register: what INTEGRATION.md adds after the reference's calloc, c:2604):
        # the kernel renders straight into it, no copy at the end of the call
        before = hashlib.sha256(iq.tobytes()).hexdigest()
        synth.host_register(iq)
        try:
            iq[:] = 0
            chan, gain = pkg.ref_channels(d[(calls - 1) % 8])
            synth.fill_block_ref(chan, gain, 1.0 / fs, nsamp, iq, lay)
            same = hashlib.sha256(iq.tobytes()).hexdigest() == before
            out[name]["registered"] = dict(timed(), same_bytes_as_copied=same)
        finally:
            synth.host_unregister(iq)
    out["what"] = ("gpsbb_fill_block_ref (include/gpsbb.h): the reference's channel_t[] and gain[] in, int16 IQ into a pageable host buffer, "
                   "channel state updated in place; wall clock around the call, Python's ctypes overhead (~10 us) included; "
                   "`registered`: the same calls after gpsbb_host_register(iq_buff) - rendered straight into the caller's buffer")
    return out


def dry_run(args, pkg, dist, world, rank):
    """No GPU: the sharding, the shard seeds and the digest exchange of the N > 1 path with the CPU oracle rendering
    (tiny blocks).  What tests/test_shard_gloo.py runs under gloo with world_size 2."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hashlib
    import oracle_binding as ob
    delt, nsamp, nch = 1.0 / args.fs, args.nsamp, args.nch
    total = args.steps * STEP_PUSHES * args.push_blocks
    ch = stream_descriptors(pkg, total, nch)
    b0, b1 = pkg.shard_blocks(total, rank, world)
    seeds = pkg.chain_carrier_host(ch[:b0 + 1], delt, nsamp)
    mine = ch[b0:b1].copy()
    mine["carr_phase"][0] = seeds[b0]
    iq, _, _ = ob.Oracle().fill_blocks(mine, delt, nsamp, chain=True)
    dig = [hashlib.sha256(iq[k].tobytes()).digest() for k in range(b1 - b0)]
    # ... and the 2.6 MS/s stream of m1_stream_leg: the same cut, the same seeds, 12 channels
    m_delt, m_nsamp = 1.0 / M1_FS, args.m1_nsamp
    m_total = max(1, min(args.steps, M1_DESC_STEPS)) * STEP_PUSHES * args.m1_push_blocks
    mch = stream_descriptors(pkg, m_total, M1_NCH, seed=M1_SEED)
    m0, m1 = pkg.shard_blocks(m_total, rank, world)
    m_mine = mch[m0:m1].copy()
    m_mine["carr_phase"][0] = pkg.chain_carrier_host(mch[:m0 + 1], m_delt, m_nsamp)[m0]
    m_iq, _, _ = ob.Oracle().fill_blocks(m_mine, m_delt, m_nsamp, chain=True)
    m_dig = [hashlib.sha256(m_iq[k].tobytes()).digest() for k in range(m1 - m0)]
    if world > 1:
        allg = [None] * world
        dist.all_gather_object(allg, (dig, m_dig))
        dig = [d for part in allg for d in part[0]]
        m_dig = [d for part in allg for d in part[1]]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_ranks": world, "blocks": total,
                          "stream_digest": hashlib.sha256(b"".join(dig)).hexdigest(),
                          "m1": {"blocks": m_total, "nch": M1_NCH, "fs": M1_FS, "nsamp": m_nsamp,
                                 "stream_digest": hashlib.sha256(b"".join(m_dig)).hexdigest()}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps per region; a step is 3200 blocks of the stream")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps; the median is reported")
    ap.add_argument("--push-blocks", type=int, default=PUSH_BLOCKS, help="0.1 s blocks per push")
    ap.add_argument("--depth", type=int, default=6, help="ring slots")
    ap.add_argument("--nch", type=int, default=16)
    ap.add_argument("--fs", type=float, default=25e6)
    ap.add_argument("--nsamp", type=int, default=2500000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU legs (baselines and the parity check against the oracle)")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work per CPU baseline leg (the M1 leg gets half)")
    ap.add_argument("--parity-blocks", type=int, default=8, help="leading blocks of the shard compared with the oracle (+ one block in each of --parity-spots pushes spread over the shard, + 1 in the last push)")
    ap.add_argument("--parity-spots", type=int, default=8)
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement (profiling runs)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: sharding / seeds / digest exchange with the CPU oracle")
    ap.add_argument("--m1-push-blocks", type=int, default=M1_PUSH_BLOCKS, help="blocks per push of the 2.6 MS/s stream leg")
    ap.add_argument("--m1-nsamp", type=int, default=M1_NSAMP, help="samples per block of the 2.6 MS/s legs (the reference's NUM_SAMPLES)")
    ap.add_argument("--fill-calls", type=int, default=200, help="gpsbb_fill_block_ref calls timed per case of the fill_block leg")
    ap.add_argument("--no-m1-stream", action="store_true", help="skip the 2.6 MS/s stream leg (measurement aid)")
    ap.add_argument("--cpu-legs", action="store_true", help="(internal) only the CPU baselines, as a process of their own: what rank 0 starts beside its GPU legs")
    args = ap.parse_args()
    if args.cpu_legs:
        return cpu_legs_main(args)

    import torch  # first: it brings the HIP runtime the library then shares
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if STEP_PUSHES % world:
        raise SystemExit("the number of ranks must divide %d" % STEP_PUSHES)
    # one rank per GPU; GPSBB_BENCH_BACKEND=gloo lets the N>1 code path be exercised where ranks share a device
    backend = os.environ.get("GPSBB_BENCH_BACKEND", "nccl")
    if args.dry_run:
        backend = "gloo"
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but %d GPUs" % (world, ndev))
    local = local % max(ndev, 1)
    # GPSBB_BENCH_FORCE_DIST: take the N > 1 code path (process group, RCCL all_reduce / all_gather / barrier) with ONE rank
    # too — what `torchrun --nproc-per-node 1` on a 1-GPU box can exercise of the path an 8-GPU node runs
    use_dist = world > 1 or (os.environ.get("GPSBB_BENCH_FORCE_DIST") and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.dry_run:
        dry_run(args, pkg, dist, world, rank)
        if use_dist:
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    # this rank next to its GPU: the thread that builds descriptors, pushes, and touches the pinned ring of the gather leg runs
    # on the CPUs local to the GPU's PCI function (the reference pins its threads: plutogpssim.c:2045-2056); the node driver's
    # producer threads do the same for themselves (gpsbb_node.h)
    affinity = {"numa_node": -1, "cpus_bound": 0}
    if not os.environ.get("GPSBB_BENCH_NO_AFFINITY"):
        try:
            node, cpulist = pkg.device_affinity(local)
            cpus = set()
            for part in filter(None, cpulist.split(",")):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
                affinity = {"numa_node": node, "cpus_bound": len(cpus)}
        except (OSError, RuntimeError, ValueError):
            pass

    delt, nsamp, nch, PB = 1.0 / args.fs, args.nsamp, args.nch, args.push_blocks
    K, W, R = args.steps, args.warmup, args.repeats
    ppr = STEP_PUSHES // world                 # pushes per rank and step
    total = K * STEP_PUSHES * PB               # blocks of the stream
    # ---- the stream and this rank's shard of it (set-up, not timed as throughput; the seed is timed on its own) ----
    # Every rank builds ITS shard in full and, of the blocks before it, only what the carrier chain reads (prn, f_carr,
    # carr_phase: three of the 69 words of a descriptor) — not the whole stream on every rank.
    b0, b1 = pkg.shard_blocks(total, rank, world)
    t_gen = time.perf_counter()
    mine = stream_descriptors(pkg, total, nch, first=b0, count=b1 - b0)
    before = np.concatenate([stream_descriptors(pkg, total, nch, first=0, count=b0, fields=("prn", "f_carr", "carr_phase")), mine[:1]])
    t_gen = time.perf_counter() - t_gen
    synth = pkg.Synth(local)
    synth.shard_seed(before, min(b0, 64), delt, nsamp)   # first use: scratch allocation, kernels loaded (not part of the seed's cost)
    synth.sync()
    t_seed = time.perf_counter()
    seed0 = synth.shard_seed(before, b0, delt, nsamp)    # exact carrier phase at the shard's first block: the device chains the b0 blocks before it
    t_seed = time.perf_counter() - t_seed
    mine["carr_phase"][0] = seed0              # the shard starts from the stream's exact phase
    npush = mine.shape[0] // PB                # = K * ppr
    del before

    def barrier():
        synth.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def run_ring(st, first, count, depth):
        """`count` pushes starting at push number `first` of the shard (cyclically), every slot popped; the ring ends drained"""
        for j in range(count):
            if st.pending >= depth:
                st.pop(copy=False)
            k = (first + j) % npush
            st.push(mine[k * PB:(k + 1) * PB])
        while st.pending:
            st.pop(copy=False)

    # ---- headline: K steps of fresh pushes, chained on the device, IQ into HBM ----
    st = synth.stream(nch, delt, nsamp, PB, depth=args.depth, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
    run_ring(st, 0, W * ppr, args.depth)
    pos = W * ppr
    barrier()
    st.timing_stats(reset=True)
    times = []
    for _ in range(R):
        barrier()
        t0 = time.perf_counter()
        run_ring(st, pos, K * ppr, args.depth)
        synth.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
        pos += K * ppr
    barrier()
    stats = st.timing_stats(reset=True)
    chain_dev = synth.info(pkg.INFO_CHAIN_ON_DEVICE)
    last_kernel = synth.info(pkg.INFO_LAST_KERNEL)
    chain_info = {"blocks_walked_sequentially": synth.info(pkg.INFO_CHAIN_FALLBACKS), "of_block_channels": pos * PB * nch,
                  "wrap_ties_recorded": synth.info(pkg.INFO_CHAIN_TIES),
                  "lane_runs_recomputed_exactly_by_k_synth_ev": synth.info(pkg.INFO_EXACT_RUNS)}
    st.close()
    elapsed = statistics.median(times)
    samples_per_step = STEP_PUSHES * PB * nsamp
    samples_per_launch = PB * nsamp
    ms_synth = stats["ms_synth_sum"] / max(stats["runs"], 1)
    ms_seed = stats["ms_seed_sum"] / max(stats["runs"], 1)
    achieved = 4.0 * samples_per_launch / (ms_synth * 1e-3) / 1e9  # GB/s, algorithmic bytes / kernel time

    # ---- PCIe-inclusive: the same shard through a ring with the pinned gather (never `value`) ----
    gather = None
    if not args.no_extras:
        gb, gdepth, gslots = 32, 8, 32
        gst = synth.stream(nch, delt, nsamp, gb, depth=gdepth, flags=pkg.CHAIN_CARRIER)
        dig = []

        def gather_run(nslots, keep):
            pushed = popped = 0
            while popped < nslots:
                while pushed < nslots and gst.pending < gdepth:
                    k = pushed % (mine.shape[0] // gb)
                    gst.push(mine[k * gb:(k + 1) * gb])
                    pushed += 1
                iq, _ = gst.pop(copy=False)
                if keep:
                    dig.extend(block_digest(iq[j]) for j in range(gb))
                popped += 1
        gather_run(gdepth, False)
        barrier()
        t0 = time.perf_counter()
        gather_run(gslots, True)
        gdt = time.perf_counter() - t0
        gst.close()
        per_rank = gslots * gb * nsamp * 4 / gdt / 1e9
        mydig = zlib.crc32(np.asarray(dig, np.uint32).tobytes())
        if use_dist:
            allg = [None] * world
            dist.all_gather_object(allg, (per_rank, mydig))
            rates, digs = [a[0] for a in allg], [a[1] for a in allg]
        else:
            rates, digs = [per_rank], [mydig]
        gather = {"value": sum(rates) / 4.0 * 1e9, "unit": "IQ samples/s", "per_rank_GBps_to_host": rates, "node_GBps_to_host": sum(rates),
                  "expectation": "per GPU: PCIe Gen5 x16, 63 GB/s raw, ~51 measured; per node min(N x 51 GB/s, what the host's DRAM takes: "
                                 "8 x 51 = 0.41 TB/s is about the write bandwidth of a 2-socket DDR5 host, so beyond ~4 GPUs the pinned rings "
                                 "have to sit on their GPU's own NUMA node (gpsbb_node.h binds them there) to keep scaling)",
                  "slot_blocks": gb, "depth": gdepth, "slots": gslots, "chained": True,
                  "digest_of_block_digests": zlib.crc32(np.asarray(digs, np.uint32).tobytes()),
                  "note": "IQ stored into pinned host memory by a copy kernel on the side stream, the first and last 64 KiB of every block digested on arrival; PCIe Gen5 x16 = 63 GB/s raw"}

    # ---- the reference's own geometry as a time-sharded stream, at every N (all ranks: its regions are bracketed by barriers) ----
    m1_stream, m1_mine, m1_bad, ob_all = None, None, 0, None
    if not args.no_cpu and not args.no_extras:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_binding as ob_all
        ob_all.Oracle().fill_blocks(mine[:1], delt, 1000)   # the oracle's one-time tables, before any thread uses it
    if not args.no_extras and not args.no_m1_stream:
        m1_stream, m1_mine, m1_bad = m1_stream_leg(pkg, synth, torch, dist, use_dist, backend, dev, rank, world, ob_all, K, W, R, args.depth,
                                                   args.m1_push_blocks, args.m1_nsamp, max(1, min(args.parity_blocks, 8)), min(args.parity_spots, 4))

    # ---- the CPU baselines, at every N: rank 0 starts them as a PROCESS of their own (bench.py --cpu-legs: no GPU, its own address
    # space and interpreter) once the timed regions are over; its one JSON line is read before this rank prints its own.  The
    # ranks' GPU legs below go on meanwhile: nobody waits for the CPU ----
    cpu_proc, cpu_out = None, {}
    if rank == 0 and ob_all is not None:
        import subprocess
        cpu_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-legs", "--steps", str(K), "--push-blocks", str(PB), "--nch", str(nch),
                                     "--fs", repr(args.fs), "--nsamp", str(nsamp), "--cpu-budget", repr(args.cpu_budget),
                                     "--m1-push-blocks", str(args.m1_push_blocks), "--m1-nsamp", str(args.m1_nsamp)],
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    # ---- the same pushes through the PRODUCT's node driver (include/gpsbb_node.h): one process, one producer thread +
    # handle + ring per shard, one sink.  N = 1 on this rank's GPU must agree with the headline (same ring, same pushes,
    # driven from C instead of from this script); with one rank and several GPUs visible, also all of them in one process,
    # every shard seeded by the device-side chain, the sink taking slots as they complete ----
    node = None
    if rank == 0 and not args.no_extras:
        def node_leg(devices):
            nflags = pkg.NODE_DEVICE_ONLY | pkg.NODE_INDEXED | pkg.NODE_CONCURRENT
            with pkg.Node(len(devices), nch, delt, nsamp, PB, depth=args.depth, flags=nflags, devices=devices) as nd:
                cnt = [0]

                def sink(iq, first, nb, shard):
                    cnt[0] += nb
                    return 0
                nd.run(mine[:min(mine.shape[0], 2 * PB * len(devices))], sink)   # warm-up: rings allocated, kernels loaded
                best = None
                for _ in range(3):
                    stn = nd.run(mine, sink)
                    best = stn if best is None or stn["seconds"] < best["seconds"] else best
            return {"value": mine.shape[0] * nsamp / best["seconds"], "unit": "IQ samples/s", "shards": len(devices), "devices": list(devices),
                    "seconds": best["seconds"], "blocks": mine.shape[0], "seed_seconds_max": max(x["seed_seconds"] for x in best["shards"]),
                    "numa_nodes": [x["numa_node"] for x in best["shards"]], "cpus_bound": [x["cpus_bound"] for x in best["shards"]]}
        try:
            node = {"one_shard": node_leg([local]),
                    "note": "gpsbb_node_run (C, one producer thread per shard bound to its GPU's NUMA node), GPSBB_NODE_DEVICE_ONLY rings of the "
                            "headline's geometry, one pass over this rank's shard, best of 3; includes the shard seeds"}
            node["one_shard"]["vs_headline"] = node["one_shard"]["value"] / (samples_per_step * K / elapsed / world)
        except Exception as e:  # the headline stands on its own
            node = {"error": repr(e)}

    # ---- the seed of the slowest rank; bytes of evidence on the timed mode's output ----
    def over_ranks(x, op):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())
    seed_max = over_ranks(t_seed, dist.ReduceOp.MAX)
    parity = None
    if not args.no_extras:
        ob = ob_all
        n_ok, bad, digs, iq_digs = parity_check(pkg, ob, synth, mine, delt, nsamp, PB, nch, max(1, min(args.parity_blocks, PB)), args.parity_spots)
        t_cross = time.perf_counter()
        cross_bad, cross_kernel = cross_check(pkg, synth, mine, delt, nsamp, PB, nch, iq_digs)
        t_cross = time.perf_counter() - t_cross
        n_all = int(over_ranks(float(n_ok), dist.ReduceOp.SUM))
        n_bad = int(over_ranks(float(len(bad)), dist.ReduceOp.SUM))
        n_cross_bad = int(over_ranks(float(len(cross_bad)), dist.ReduceOp.SUM))
        if use_dist:
            allg = [None] * world
            dist.all_gather_object(allg, (digs, iq_digs))
            digs = [d for part in allg for d in part[0]]
            iq_digs = np.concatenate([part[1] for part in allg])
        parity = {"checked_blocks": n_all, "mismatching_blocks": n_bad, "per_rank": n_ok,
                  # every block of every rank's shard: the timed mode's rendering (model kernels, lap-parallel pre-pass) against the
                  # per-sample kernel's (genuine IEEE steps, row-walk pre-pass), one device-side digest per block
                  "blocks_cross_checked": int(iq_digs.shape[0]), "cross_mismatching_blocks": n_cross_bad,
                  "cross_check": {"against": "k_synth on k_seed's rows (GPSBB_OPT_SYNTH_KERNEL 1)" if cross_kernel == 1 else "?",
                                  "digest": "gpsbb_device_digest (64 bits per block, on the device)", "seconds_rank0": t_cross,
                                  "stream_iq_digest": zlib.crc32(iq_digs.tobytes())},
                  "stream_end_state_digest": zlib.crc32(np.asarray(digs, np.uint32).tobytes()), "blocks_digested": len(digs),
                  "what": "int16 IQ and end-of-block carr_phase of blocks read back from the HBM slots of a "
                          "GPSBB_STREAM_DEVICE_ONLY ring (the timed mode) vs the CPU oracle, bit for bit: the first blocks of "
                          "every rank's shard, one block of every (pushes / %d)-th push and one of its last push" % args.parity_spots}
        if n_bad:
            sys.stderr.write("bench.py: rank %d: IQ differs from the oracle in blocks %s of its shard\n" % (rank, bad))
        if cross_bad:
            sys.stderr.write("bench.py: rank %d: the model kernels and the per-sample kernel differ in %d blocks of its shard, first %s\n" %
                             (rank, len(cross_bad), cross_bad[:8]))

    # ---- the PRODUCT's multi-device path on every GPU this process can see, checked against what the ranks rendered: rank 0
    # runs gpsbb_node_run over the WHOLE stream on all visible GPUs — contiguous shards into an indexed sink, then slots dealt
    # round the GPUs into the ordered sink (the reference's one consumer, c:2146-2158) — digests every slot on the GPU that
    # rendered it and compares with the per-block digests gathered from the ranks.  Under the driver's torchrun invocation
    # (one rank per GPU, every rank sees all GPUs) this is where the node driver first meets several physical devices. ----
    if rank == 0 and not args.no_extras and parity is not None and node is not None and "error" not in node:
        try:
            full = mine if world == 1 else stream_descriptors(pkg, total, nch)
            devs = list(range(max(ndev, 1))) if backend == "nccl" or world == 1 else [local]
            legs = {}
            for name, nflags in (("contiguous_indexed", pkg.NODE_DEVICE_ONLY | pkg.NODE_INDEXED | pkg.NODE_CONCURRENT),
                                 ("interleaved_ordered", pkg.NODE_DEVICE_ONLY | pkg.NODE_INTERLEAVED | pkg.NODE_DIGESTS)):
                got = np.zeros(full.shape[0], np.uint64)
                seen = np.zeros(full.shape[0], np.int32)
                order = []
                with pkg.Node(len(devs), nch, delt, nsamp, PB, depth=args.depth, flags=nflags, devices=devs) as nd:
                    if name == "contiguous_indexed":
                        # the driver's OWN sink (gpsbb_node_run_digest: C, every shard's producer thread digests its slots on its GPU
                        # while its ring keeps rendering): what the driver delivers when the consumer is not a Python callback
                        nd.run_digest(full[:min(full.shape[0], 2 * PB * len(devs))])
                        stn, got = nd.run_digest(full)
                        seen += 1
                        order = list(range(0, full.shape[0], PB))
                    else:
                        # the ordered consumer, as a host would write it in Python: one callback per slot, in stream order
                        def sink(iq, first, nb, shard):
                            # (GPSBB_NODE_DIGESTS: the pushes were rendered with their digests; reading the slot back instead -
                            # helpers[devs[shard]].slot_digest(iq, nb, nsamp) - costs the synthesis behind it a third of its rate)
                            got[first:first + nb] = nd.slot_digests(shard, nb)
                            seen[first:first + nb] += 1
                            order.append(first)
                            return 0
                        stn = nd.run(full, sink)
                legs[name] = {"value": full.shape[0] * nsamp / stn["seconds"], "unit": "IQ samples/s (a digest of every block included)",
                              "sink": "the driver's own: gpsbb_node_run_digest (C; pushes rendered with their digests)" if name == "contiguous_indexed" else "a Python callback per slot, in stream order, taking the digests the slot was rendered with (GPSBB_NODE_DIGESTS, gpsbb_node_slot_digests)",
                              "seconds": stn["seconds"], "blocks": int(full.shape[0]), "devices": devs,
                              "every_block_once": bool((seen == 1).all()), "in_stream_order": order == sorted(order),
                              "digests_equal_the_ranks": bool((got == iq_digs).all()),
                              "blocks_that_differ": int((got != iq_digs).sum()),
                              "shards": [{k: x[k] for k in ("first_block", "nblocks", "device", "numa_node", "cpus_bound", "seed_seconds", "busy_seconds", "wait_seconds")}
                                         for x in stn["shards"]]}
            all_equal = all(legs[k]["digests_equal_the_ranks"] and legs[k]["every_block_once"] for k in legs)
            node["all_gpus"] = legs
            node["all_gpus"]["expectation"] = ("N GPUs: contiguous shards into the driver's own digest sink scale with N (every GPU renders its pushes WITH their digests - "
                                               "GPSBB_PUSH_DIGEST: the synthesis kernel adds them up as it renders, nothing is read back - and its shard's producer thread "
                                               "copies them out in C: the 1-GPU figure is ~0.9 x one_shard); the ordered sink over interleaved slots delivers in stream order "
                                               "at the same rate as long as the consumer keeps up - here ONE Python callback per slot (it takes the digests the slot was rendered with: "
                                               "GPSBB_NODE_DIGESTS), which at N = 8 is what bounds that leg, not the driver")
            if not all_equal:
                parity["node_driver_mismatch"] = True
        except Exception as e:
            node["all_gpus"] = {"error": repr(e)}

    res = None
    if rank == 0:
        res = {
            "metric": "IQ samples/sec (whole node) at 16 channels; bit-exact int16 IQ vs CPU ref",
            "value": samples_per_step * K / elapsed,
            "unit": "IQ samples/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64 NCOs (exact), i32 accumulate of packed int16 I/Q", "data": "synthetic",
            "config": {"carrier_nco": "IEEE double (reference default), chained across all blocks of the stream",
                       "workload": "synth16-S as one stream: %d ch, fs %.3g S/s, %d-sample blocks, seeded descriptors "
                                   "(splitmix64 0x5EED) with a slowly varying Doppler, %d blocks per step (%d pushes of %d "
                                   "fresh blocks), %d blocks in all, cut into %d contiguous time shards" %
                                   (nch, args.fs, nsamp, STEP_PUSHES * PB, STEP_PUSHES, PB, total, world),
                       "global_samples_per_step": samples_per_step, "parallelism": "time-shard x%d" % world,
                       "ring_depth": args.depth, "carrier_chain": "device" if chain_dev else "host threads",
                       "synthesis_kernel": "k_synth_ev" if last_kernel == 2 else "k_synth"},
            "repeats": {"n": R, "seconds": times, "ms_per_step_min": min(times) / K * 1e3,
                        "ms_per_step_max": max(times) / K * 1e3, "value_min": samples_per_step * K / max(times),
                        "value_max": samples_per_step * K / min(times)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_synth_ev" if last_kernel == 2 else "k_synth", "ms_per_launch": ms_synth,
                         "launches_timed": stats["runs"], "algorithmic_bytes_per_launch": 4 * samples_per_launch,
                         "note": "not HBM-bound: 31 VALU instructions per channel-run, and the package at its power limit (1.33 - 1.35 kW of 1.4 kW, shader "
                                 "clock 2.05 - 2.3 GHz measured inside the kernel): the chip delivers the same vector-instruction rate with and without "
                                 "the pre-pass of the next pushes beside this kernel, so `ms_per_launch` here = (this kernel's + the pre-pass's "
                                 "instructions) at that rate; `alone` = this kernel's only (tools/corun_diag.py, tools/power_probe.sh, DESIGN.md 3.1)"},
            "prepass_ms_per_launch": ms_seed,
            "device_chain": chain_info,
            "shard_seed_s": seed_max, "descriptor_generation_s": t_gen,
            "shard_seed": {"seconds_max_over_ranks": seed_max, "seconds_rank0": t_seed,
                           "blocks_before_the_last_shard": pkg.shard_blocks(total, world - 1, world)[0],
                           "how": "gpsbb_chain_carrier on each rank's GPU over the blocks before its shard (exact, parallel over the blocks)"},
            # one pass over the whole stream (K steps) with the slowest rank's seed charged to it
            "value_incl_seed": samples_per_step * K / (elapsed + seed_max),
            "parity": parity,
            "parity_checked_blocks": parity["checked_blocks"] if parity else 0,
            "dist": {"process_group": bool(use_dist), "backend": backend if use_dist else None, "world": world,
                     "hw_queues": synth.info(pkg.INFO_HW_QUEUES), "streams_of_the_handle": synth.info(pkg.INFO_STREAMS),
                     "rank0_affinity": affinity},
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                res["roofline"]["traffic"] = pj.get("k_synth_hbm_bytes_per_launch")
                res["roofline"]["traffic_source"] = "PMC passes of an earlier run, %s" % pj.get("source", "profiles/pmc_latest.json")
            except Exception:
                pass
        if gather:
            res["gather"] = gather
        if node:
            res["node_driver"] = node
        sq = os.path.join(ROOT, "profiles", "sq_latest.json")
        if os.path.exists(sq):
            try:
                # the bound the kernel actually runs against: VALU issue.  Wave-instructions per launch from the kept SQ counter
                # pass (SQ_INSTS_VALU, scaled to this launch's samples), cycles per wave-instruction per SIMD from the
                # micro-benchmark of the kernel's instruction mix (profiles/r03_valu_rates_ubench.txt), 1024 SIMDs at the clock
                # measured under load: the time the SIMDs need just to ISSUE the kernel's vector instructions
                sj = json.load(open(sq))
                insts = sj["k_synth_ev_valu_wave_insts_per_sample"] * samples_per_launch
                issue_ms = insts * sj["cycles_per_valu_wave_inst"] / (sj["simds"] * sj["clock_ghz"] * 1e9) * 1e3
                res["roofline_valu"] = {"bound": "valu_issue", "kernel": "k_synth_ev", "valu_wave_insts_per_launch": insts,
                                        "cycles_per_wave_inst": sj["cycles_per_valu_wave_inst"], "simds": sj["simds"], "clock_ghz": sj["clock_ghz"],
                                        "issue_ms_per_launch": issue_ms, "ms_per_launch": ms_synth, "frac": issue_ms / ms_synth,
                                        "note": "priced at the 2.4 GHz peak clock; under this kernel the package sits at its power limit and the clock measured "
                                                "inside the kernel is 2.05 - 2.3 GHz (DESIGN.md 3.1)",
                                        "lds_bank_conflict_share_of_lds_active": sj.get("lds_bank_conflict_share"),
                                        "source": sj.get("source")}
            except Exception:
                pass
    if rank == 0 and m1_stream is not None:
        res["m1"] = {"value": m1_stream["value"], "unit": "IQ samples/s", "n_gpus": world, "stream": m1_stream, "roofline_stream": m1_stream["roofline"],
                     "note": "the reference's own geometry (plutogpssim.c:43-45: 12 ch, 2.6 MS/s, 300 000-sample blocks); `value` = the time-sharded stream of "
                             "fresh chained pushes over all ranks (m1.stream), at every N; m1.gpu* = resident re-runs of one 1000-block batch, N = 1 only"}
    if rank == 0 and not args.no_extras and world == 1:
        # (at N = 1 only: with N > 1 these legs would keep N - 1 GPUs idle behind a barrier for most of the command's wall time)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        rch = mine[:PB]  # the shard's first push: the headline's own descriptors, resident
        r0, ceil_gbs = resident_leg(pkg, synth, torch, rch, delt, nsamp, 0, 20, 4, dev)
        r1, _ = resident_leg(pkg, synth, torch, rch, delt, nsamp, pkg.CHAIN_CARRIER, 20, 5, dev)
        res["resident"] = {"independent_blocks": r0, "chained": r1, "unit": "IQ samples/s",
                           "note": "the stream's first push (%d blocks) as one batch re-run 20 times, descriptors and plans resident in HBM (round 1's kind of value)" % PB}
        # the reference built without FLOAT_CARR_PHASE (h:12): the 32-bit fixed-point carrier on the same descriptors
        fxr = rch.copy()
        fxr["carr_phase"] = np.floor(fxr["carr_phase"] * 2.0 ** 32)
        r3, _ = resident_leg(pkg, synth, torch, fxr, delt, nsamp, pkg.CHAIN_CARRIER | pkg.FIXED_CARRIER, 20, 4, dev)
        r3["synthesis_kernel"] = {1: "k_synth", 2: "k_synth_ev_fixed"}.get(synth.info(pkg.INFO_LAST_KERNEL), "?")
        res["resident"]["fixed_point_carrier_chained"] = r3
        res["roofline"]["write_ceiling_measured_GBs"] = ceil_gbs
        res["roofline"]["frac_of_measured_ceiling"] = achieved / ceil_gbs
        # the same kernel with the GPU to itself (tables already built): what the pre-passes running beside it cost
        r2, _ = resident_leg(pkg, synth, torch, rch, delt, nsamp, 0, 20, 5, dev, synth_only=True)
        alone = 4.0 * samples_per_launch / (r2["synth_kernel_ms"] * 1e-3) / 1e9
        res["roofline"]["alone"] = {"ms_per_launch": r2["synth_kernel_ms"], "achieved": alone, "frac": alone / HBM_PEAK_GBS,
                                    "note": "k_synth_ev on resident tables, no pre-pass running beside it"}
        if "roofline_valu" in res:
            res["roofline_valu"]["frac_alone"] = res["roofline_valu"]["issue_ms_per_launch"] / r2["synth_kernel_ms"]
        # the reference-faithful geometry (12 ch, 2.6 MS/s, 300 000-sample blocks)
        mns = args.m1_nsamp
        mch = pkg.synth_descriptors(args.m1_push_blocks, nch=12, seed=0xF00D)
        m1, _ = resident_leg(pkg, synth, torch, mch, 1.0 / 2.6e6, mns, 0, 30, 4, dev)
        m1_kernel = {1: "k_synth", 2: "k_synth_pd (every channel evaluated per sample on the in-tile model)"}.get(
            synth.info(pkg.INFO_LAST_KERNEL), "?")
        m1s, _ = resident_leg(pkg, synth, torch, mch, 1.0 / 2.6e6, mns, 0, 30, 5, dev, synth_only=True)
        m1c, _ = resident_leg(pkg, synth, torch, mch, 1.0 / 2.6e6, mns, pkg.CHAIN_CARRIER, 30, 4, dev)
        m1_alg = 4.0 * mch.shape[0] * mns
        res.setdefault("m1", {"unit": "IQ samples/s"})
        res["m1"].update({"gpu": m1, "gpu_chained_on_the_device": m1c,
                     "roofline": {"bound": "valu+lds issue", "priced_against": "hbm", "kernel": "k_synth_pd", "ms_per_launch": m1["synth_kernel_ms"],
                                  "achieved": m1_alg / (m1["synth_kernel_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": m1_alg / (m1["synth_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "alone": {"ms_per_launch": m1s["synth_kernel_ms"],
                                            "frac": m1_alg / (m1s["synth_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                  "bound_measured": "valu+lds issue",
                                  "note": "VALU- and LDS-issue-bound (about 27 VALU issue cycles and 12 bytes of LDS reads per channel-sample), not HBM-bound: "
                                          "DESIGN.md 2.4; the stores cost it 7 % (tools/bound_hunt.sh PD_NOSTORE)"},
                     "workload": "12 ch, fs 2.6e6 S/s, %d-sample blocks, %d independent blocks per step (resident re-runs); synthesis kernel %s" % (mns, mch.shape[0], m1_kernel)})
        # the reference built without FLOAT_CARR_PHASE (h:12): 32-bit fixed-point carrier, the same M1 geometry
        fch = mch.copy()
        fch["carr_phase"] = np.floor(fch["carr_phase"] * 2.0 ** 32)
        m1f, _ = resident_leg(pkg, synth, torch, fch, 1.0 / 2.6e6, mns, pkg.FIXED_CARRIER, 20, 4, dev)
        m1f["synthesis_kernel"] = {1: "k_synth", 2: "k_synth_pd"}.get(synth.info(pkg.INFO_LAST_KERNEL), "?")
        res["m1"]["gpu_fixed_point_carrier"] = m1f
    fill_bad = False
    if cpu_proc is not None:
        try:
            so, se = cpu_proc.communicate(timeout=600)
            cpu_out = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
        except Exception as e:  # reported in the line, never fatal for the GPU figures
            cpu_out = {"error": repr(e)}
        if "cpu_baseline" in cpu_out:
            res["cpu_baseline"] = cpu_out["cpu_baseline"]
            res["cpu_baseline"]["concurrency"] = "a process of its own (bench.py --cpu-legs, 1 thread) started by rank 0 after the timed regions, beside that rank's remaining GPU legs"
        if "m1_cpu" in cpu_out and "m1" in res:
            res["m1"]["cpu"] = cpu_out["m1_cpu"]
        if "error" in cpu_out:
            res["cpu_baseline_error"] = cpu_out["error"]
    if rank == 0 and not args.no_extras:
        # the drop-in call itself, after the CPU legs have left the cores alone
        try:
            res["fill_block"] = fill_block_leg(pkg, synth, ob_all, (("reference_block_12ch_2.6MSps_300000", 12, 2.6e6, args.m1_nsamp, 0xBEEF),
                                                                     ("headline_block_16ch_25MSps", nch, args.fs, nsamp, 0xBEE5)), calls=args.fill_calls)
            fill_bad = any(v.get("equals_oracle") is False for v in res["fill_block"].values() if isinstance(v, dict))
        except Exception as e:
            res["fill_block"] = {"error": repr(e)}
    if use_dist:
        dist.barrier()   # the other ranks wait for rank 0's node-driver leg
    if rank == 0:
        print(json.dumps(res))
    synth.close()
    if use_dist:
        dist.destroy_process_group()
    if parity and (parity["mismatching_blocks"] or parity["cross_mismatching_blocks"] or parity.get("node_driver_mismatch")):
        raise SystemExit(3)
    if m1_bad or fill_bad:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
