"""include/gpsbb_node.h: one process, N producer threads / handles / rings, ONE sink (the reference has one consumer of one
stream, plutogpssim.c:2146-2158).  On a one-GPU test box the N shards share device 0; on a box with several GPUs they are
spread over them (node_devices below: the first execution of the driver on distinct physical devices should be a test, not a
user); what is checked is what N GPUs have to get right: contiguous time shards seeded with the exact carrier phase by the device-side chain, blocks
delivered once each and — in the default mode — strictly in stream order, the same bytes whatever N is, equal to the golden
vectors of the reference's own code and to the CPU oracle."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Collect:
    """a sink that copies every delivery into one host array and records the order of arrival"""

    def __init__(self, nblocks, nsamp):
        self.iq = np.zeros((nblocks, nsamp, 2), np.int16)
        self.calls = []
        self.nsamp = nsamp

    def __call__(self, iq_ptr, first, nb, shard):
        src = (C.c_int16 * (nb * self.nsamp * 2)).from_address(iq_ptr)
        self.iq[first:first + nb] = np.frombuffer(src, np.int16).reshape(nb, self.nsamp, 2)
        self.calls.append((first, nb, shard))
        return 0


def node_devices(nshards):
    """HIP device ordinals for nshards shards: round the GPUs of the box where it has several, device 0 for all otherwise"""
    try:
        import torch
        ndev = max(1, torch.cuda.device_count())
    except Exception:
        ndev = 1
    return [k % ndev for k in range(nshards)]


def render(pkg, ch, fs, nsamp, nshards, bps, depth=2, flags=0):
    sink = Collect(ch.shape[0], nsamp)
    with pkg.Node(nshards, ch.shape[1], 1.0 / fs, nsamp, bps, depth=depth, flags=flags, devices=node_devices(nshards)) as node:
        st = node.run(ch, sink)
    assert st["blocks"] == ch.shape[0] and sum(c[1] for c in sink.calls) == ch.shape[0]
    return sink, st


def test_reference_scenario_in_1_2_and_4_shards_equals_the_golden_vectors(pkg):
    """BASELINE configs 1/2 end to end: RINEX -> front end -> 301 blocks (30.1 s, across the 30 s nav refresh) -> node
    driver.  The golden blocks 0, 1, 2, 149, 299, 300 of the reference's own code fall into different shards."""
    pkg.build_frontend()
    z = np.load(os.path.join(GOLDEN, "static_F.npz"))
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    blocks = [int(b) for b in z["blocks"]]
    fe = pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=(30.286502, 120.032669, 100.0), max_chan=12)
    ch = fe.generate(max(blocks) + 1)
    fe.close()
    first = None
    for nshards in (1, 2, 4):
        sink, st = render(pkg, ch, fs, nsamp, nshards, bps=8)
        # one ordered stream: every delivery starts where the one before ended
        pos = 0
        for f, nb, _ in sink.calls:
            assert f == pos
            pos += nb
        assert len({c[2] for c in sink.calls}) == nshards
        for k, blk in enumerate(blocks):
            assert sha(sink.iq[blk]) == str(z["iq_sha256"][k]), (nshards, blk)
        if first is None:
            first = sink.iq
        else:
            assert (sink.iq == first).all(), nshards
        plan = pkg.node_plan(ch.shape[0], nshards, 8)
        assert [s["first_block"] for s in st["shards"]] == plan[:-1]
        assert all(s["seed_seconds"] > 0 for s in st["shards"][1:]) and st["shards"][0]["seed_seconds"] == 0


def test_fresh_streams_four_to_a_gpu_over_and_over(pkg):
    """A statistical guard (tools/stress_node.py is the long version): the library's streams are non-blocking ones, and the
    cleared carry of a fresh stream used to be a null-stream hipMemset — which could land after the first push's fix-up had
    written the exact end phase, leaving every later block of the shard off by one push's phase: about one run in a hundred
    with four handles on one GPU, never with one.  Forty fresh nodes of four shards against the one-shard bytes."""
    pkg.build_frontend()
    fs, nsamp = 2.6e6, 300000
    fe = pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=(30.286502, 120.032669, 100.0), max_chan=12)
    ch = fe.generate(301)
    fe.close()
    ref, _ = render(pkg, ch, fs, nsamp, 1, bps=8)
    for r in range(40):
        sink, _ = render(pkg, ch, fs, nsamp, 4, bps=8)
        bad = [b for b in range(ch.shape[0]) if not (sink.iq[b] == ref.iq[b]).all()]
        assert not bad, (r, bad[:8])


def test_dense_25_MSps_blocks_indexed_sink_and_oracle(pkg, oracle):
    """config 3/5 geometry (16 channels, 25 MS/s, 2.5 M-sample blocks) through the front end: golden blocks 0 and 1, then
    every block against N = 1 and a sample of them against the oracle's sequential render; the sink takes slots in
    completion order (GPSBB_NODE_INDEXED) from several threads at once (GPSBB_NODE_CONCURRENT)."""
    pkg.build_frontend()
    z = np.load(os.path.join(GOLDEN, "dense_S.npz"))
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    fe = pkg.FrontEnd(os.path.join(GOLDEN, "dense3540.14n"), llh=(30.286502, 120.032669, 100.0), max_chan=16)
    ch = fe.generate(8)
    fe.close()
    one, _ = render(pkg, ch, fs, nsamp, 1, bps=1)
    for k in (0, 1):
        assert sha(one.iq[k]) == str(z["iq_sha256"][k]), k
    four, st = render(pkg, ch, fs, nsamp, 4, bps=1, flags=pkg.NODE_INDEXED | pkg.NODE_CONCURRENT)
    assert (four.iq == one.iq).all()
    assert sorted(c[0] for c in four.calls) == list(range(8))
    # the first 200 000 samples of blocks 0..6 from their exact start phases: cheap for the oracle
    seeds = pkg.chain_carrier_host(ch, 1.0 / fs, nsamp)
    chs = ch.copy()
    chs["carr_phase"] = seeds
    want, _, _ = oracle.fill_blocks(chs[:7], 1.0 / fs, 200000)
    assert (four.iq[:7, :200000] == want).all()


def test_shard_boundaries_prn_changes_padding_fixed_carrier_and_stop(pkg, oracle):
    """Small streams against the oracle's sequential render: a channel that changes satellite exactly at a shard
    boundary (it must start from its own phase, c:1956-1964), one that goes idle and comes back, a block count that is
    not a multiple of the push size (the last slot is padded), the 32-bit accumulator; a sink that stops the run."""
    fs, nsamp, nch, nb, bps = 4.0e6, 30000, 6, 23, 3
    ch = pkg.synth_descriptors(nb, nch=nch, seed=99)
    ch["f_carr"] = ch["f_carr"][0][None, :] + np.arange(nb)[:, None] * 3.0
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    plan = pkg.node_plan(nb, 3, bps)
    assert plan == [0, 6, 15, 23]
    ch["prn"][plan[1]:, 2] = 30          # hand-over exactly at the first boundary
    ch["prn"][plan[2] - 2:plan[2] + 1, 4] = 0   # idle across the second one
    want, _, _ = oracle.fill_blocks(ch, 1.0 / fs, nsamp, chain=True)
    for nshards in (1, 3):
        sink, _ = render(pkg, ch, fs, nsamp, nshards, bps=bps)
        assert (sink.iq == want).all(), nshards
    chf = ch.copy()
    chf["carr_phase"] = np.floor(chf["carr_phase"] * 2.0 ** 32)
    wantf, _, _ = oracle.fill_blocks(chf, 1.0 / fs, nsamp, chain=True, fixed=True)
    sink, _ = render(pkg, chf, fs, nsamp, 3, bps=bps, flags=pkg.NODE_FIXED_CARRIER)
    assert (sink.iq == wantf).all()
    # the sink stops the stream after the third delivery (plutogpssim.c:2153-2157: a negative push ends the loop)
    seen = []

    def stopper(iq_ptr, first, n, shard):
        seen.append(first)
        return -1 if len(seen) == 3 else 0

    with pkg.Node(3, nch, 1.0 / fs, nsamp, bps, depth=2, devices=node_devices(3)) as node:
        st = node.run(ch, stopper, expect_stop=True)
        assert st["rc"] == -7 and seen == [0, 3, 6] and st["blocks"] == 9
        sink = Collect(nb, nsamp)      # ... and the node is usable again afterwards
        node.run(ch, sink)
        assert (sink.iq == want).all()


def test_interleaved_shards_keep_the_order_and_the_bytes(pkg, oracle):
    """GPSBB_NODE_INTERLEAVED: the slots of the stream go round the shards, every slot a chain of its own from the exact
    phase of its first block (gpsbb_chain_carrier over the whole stream + GPSBB_PUSH_NEW_CHAIN) — the layout in which an
    ORDERED consumer gets N GPUs' rate.  Same bytes as the contiguous layout, the golden blocks and the oracle's sequential
    render; deliveries strictly in stream order, the shards taking turns; PRN changes and idle channels at slot edges; a
    block count that is not a multiple of the slot size; the 32-bit accumulator."""
    pkg.build_frontend()
    z = np.load(os.path.join(GOLDEN, "static_F.npz"))
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    blocks = [int(b) for b in z["blocks"]]
    fe = pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=(30.286502, 120.032669, 100.0), max_chan=12)
    ch = fe.generate(max(blocks) + 1)
    fe.close()
    for nshards in (1, 3):
        sink, st = render(pkg, ch, fs, nsamp, nshards, bps=8, flags=pkg.NODE_INTERLEAVED)
        pos = 0
        for k, (f, nb, shard) in enumerate(sink.calls):
            assert f == pos and shard == k % nshards
            pos += nb
        for k, blk in enumerate(blocks):
            assert sha(sink.iq[blk]) == str(z["iq_sha256"][k]), (nshards, blk)
        assert sum(s["nblocks"] for s in st["shards"]) == ch.shape[0]
    fs, nsamp, nch, nb, bps = 4.0e6, 30000, 6, 23, 3
    ch = pkg.synth_descriptors(nb, nch=nch, seed=98)
    ch["f_carr"] = ch["f_carr"][0][None, :] + np.arange(nb)[:, None] * 3.0
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    ch["prn"][6:, 2] = 30          # hand-over exactly at a slot edge
    ch["prn"][11, 1] = 27          # ... and inside a slot
    ch["prn"][13:16, 4] = 0        # idle across one
    want, _, _ = oracle.fill_blocks(ch, 1.0 / fs, nsamp, chain=True)
    for nshards, flags in ((2, pkg.NODE_INTERLEAVED), (4, pkg.NODE_INTERLEAVED | pkg.NODE_INDEXED | pkg.NODE_CONCURRENT)):
        sink, _ = render(pkg, ch, fs, nsamp, nshards, bps=bps, flags=flags)
        assert (sink.iq == want).all(), nshards
    chf = ch.copy()
    chf["carr_phase"] = np.floor(chf["carr_phase"] * 2.0 ** 32)
    wantf, _, _ = oracle.fill_blocks(chf, 1.0 / fs, nsamp, chain=True, fixed=True)
    sink, _ = render(pkg, chf, fs, nsamp, 3, bps=bps, flags=pkg.NODE_INTERLEAVED | pkg.NODE_FIXED_CARRIER)
    assert (sink.iq == wantf).all()


def test_a_push_that_starts_a_new_chain(pkg, oracle):
    """gpsbb_stream_push_ex(GPSBB_PUSH_NEW_CHAIN) on a plain stream: pushes that are NOT consecutive in time, each from exact
    seeds in its descriptors, through the device-side chain and through the host-side one."""
    fs, nsamp, nch, bps = 25e6, 40000, 8, 4
    ch = pkg.synth_descriptors(6 * bps, nch=nch, seed=321)
    seeds = pkg.chain_carrier_host(ch, 1.0 / fs, nsamp)
    want, _, _ = oracle.fill_blocks(ch, 1.0 / fs, nsamp, chain=True)
    order = [4, 1, 5, 0, 3, 2]
    for where in (1, 2):
        with pkg.Synth(0) as s:
            s.set_option(pkg.OPT_SEED_WHERE, where)
            st = s.stream(nch, 1.0 / fs, nsamp, bps, depth=2, flags=pkg.CHAIN_CARRIER)
            for k in order:
                slot = ch[k * bps:(k + 1) * bps].copy()
                slot["carr_phase"][0] = seeds[k * bps]
                st.push(slot, new_chain=True)
                iq, _ = st.pop()
                assert (iq == want[k * bps:(k + 1) * bps]).all(), (where, k)
            st.close()


def test_a_failed_run_leaves_the_node_usable(pkg, oracle):
    """A run that fails half-way — a descriptor outside the contract in the middle of the stream (GPSBB_E_BADCHAN from that
    push), in the middle of a shard with slots in flight — must not leave the node dead: the failing shard drains its ring (or
    gets a new handle and ring where the library closed the stream), the others wind down, and the next gpsbb_node_run on the
    same node delivers the good stream bit for bit.  Both layouts."""
    fs, nsamp, nch, bps, nb = 25e6, 30000, 6, 2, 24
    ch = pkg.synth_descriptors(nb, nch=nch, seed=77)
    want, _, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=True)
    bad = ch.copy()
    bad["prn"][13, 2] = 40  # not a PRN: gpsbb_stream_push refuses the slot that holds block 13
    for flags in (0, pkg.NODE_INTERLEAVED):
        with pkg.Node(3, nch, 1 / fs, nsamp, bps, depth=3, flags=flags, devices=node_devices(3)) as node:
            with pytest.raises(pkg.GpsbbError):
                node.run(bad, lambda *a: 0)
            for _ in range(2):
                sink = Collect(nb, nsamp)
                st = node.run(ch, sink)
                assert st["blocks"] == nb and (sink.iq == want).all()
                assert [c[0] for c in sink.calls] == sorted(c[0] for c in sink.calls)
            with pytest.raises(pkg.GpsbbError):
                node.run(bad, lambda *a: 0)
            sink = Collect(nb, nsamp)
            node.run(ch, sink)
            assert (sink.iq == want).all()


def test_the_stream_fed_incrementally(pkg, oracle):
    """gpsbb_node_begin / _feed / _end: the stream as the reference's loop makes it (c:2655-2687: one block's descriptors, render,
    round again) — fed in pieces of any size (1 block, 7, a few slots, a piece that ends inside a slot), the driver cuts it
    into slots itself and chains the carrier across everything it is fed; the bytes are gpsbb_node_run's, the oracle's and the
    golden vectors' (the reference scenario across its 30 s nav refresh), in stream order; PRN changes at and inside slot and
    feed boundaries, an idle stretch, a short last slot, the fixed-point carrier; the sink stops the run; a second run on the
    same node."""
    z = np.load(os.path.join(GOLDEN, "static_F.npz"))
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    desc = z["desc"].view(pkg.CHAN_DTYPE).reshape(z["desc"].shape[0], -1)
    blocks = [int(b) for b in z["blocks"]]
    # the golden file holds selected blocks; a stream needs all: the dense synthetic one below, and the golden ones one by one
    rng = np.random.default_rng(8)
    fs, nsamp, nch, bps, nb = 25e6, 25000, 7, 3, 41
    ch = pkg.synth_descriptors(nb, nch=nch, seed=808)
    ch["f_carr"] = rng.uniform(-5000, 5000, (1, nch)) + rng.uniform(-2, 2, (nb, nch))
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    ch["prn"] = np.arange(1, nch + 1)[None, :]
    ch["prn"][9:, 1] = 21      # at a slot boundary
    ch["prn"][10:, 2] = 22     # inside a slot
    ch["prn"][17:26, 3] = 0    # idle for a while
    want, _, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=True)
    for nshards, pieces in ((1, [1] * nb), (3, [7, 1, 2, 9, 22]), (4, [nb]), (2, [5] * 8 + [1])):
        assert sum(pieces) == nb
        sink = Collect(nb, nsamp)
        with pkg.Node(nshards, nch, 1 / fs, nsamp, bps, depth=2, devices=node_devices(nshards)) as node:
            for _ in range(2):  # twice on the same node
                sink.calls.clear()
                sink.iq[:] = 0
                node.begin(sink)
                k = 0
                for p in pieces:
                    node.feed(ch[k:k + p])
                    k += p
                st = node.end()
                assert st["blocks"] == nb and (sink.iq == want).all(), (nshards, pieces)
                assert [c[0] for c in sink.calls] == sorted(c[0] for c in sink.calls)   # ordered: the reference's one consumer
                assert sorted({c[2] for c in sink.calls}) == list(range(min(nshards, (nb + bps - 1) // bps)))  # the slots went round the shards
            # ... and gpsbb_node_run still works on it afterwards
            sink2 = Collect(nb, nsamp)
            node.run(ch, sink2)
            assert (sink2.iq == want).all()
    # the sink stops the run: feed reports it, end winds the run down
    with pkg.Node(2, nch, 1 / fs, nsamp, bps, depth=2, devices=node_devices(2)) as node:
        seen = []

        def stopper(iq, first, n, shard):
            seen.append(first)
            return -1 if first >= 9 else 0
        node.begin(stopper)
        with pytest.raises(pkg.GpsbbError):
            for k in range(0, nb, 2):
                node.feed(ch[k:k + 2])
        node.end(expect_stop=True)
        assert max(seen) < nb - bps
    # the fixed-point carrier: the accumulator is carried across feeds in integer arithmetic
    fch = ch.copy()
    fch["carr_phase"] = np.floor(fch["carr_phase"] * 2.0 ** 32)
    fwant, _, _ = oracle.fill_blocks(fch, 1 / fs, nsamp, chain=True, fixed=True)
    sink = Collect(nb, nsamp)
    with pkg.Node(2, nch, 1 / fs, nsamp, bps, depth=2, flags=pkg.NODE_FIXED_CARRIER, devices=node_devices(2)) as node:
        node.begin(sink)
        for k in range(0, nb, 4):
            node.feed(fch[k:k + 4])
        node.end()
    assert (sink.iq == fwant).all()
    del desc, blocks


def test_placement_is_reported(pkg):
    """The producer threads bind themselves next to their GPU before they allocate (plutogpssim.c:2045-2056 pins the
    reference's two threads): the statistics say where."""
    ch = pkg.synth_descriptors(4, nch=4, seed=5)
    devs = node_devices(2)
    with pkg.Node(2, 4, 1 / 4e6, 20000, 2, devices=devs) as node:
        st = node.run(ch, lambda *a: 0)
    for s, d in zip(st["shards"], devs):
        node_id, cpus = pkg.device_affinity(d)
        assert s["device"] == d and s["numa_node"] == node_id
        if cpus:
            assert s["cpus_bound"] > 0
    with pkg.Node(1, 4, 1 / 4e6, 20000, 2, flags=pkg.NODE_NO_AFFINITY) as node:
        st = node.run(ch, lambda *a: 0)
    assert st["shards"][0]["cpus_bound"] == 0
    with pkg.Synth(0) as s:
        s.fill_block(ch[0], 1 / 4e6, 20000)
        assert s.info(pkg.INFO_HW_QUEUES) == int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        assert 4 <= s.info(pkg.INFO_STREAMS) <= 12


def test_gpsbb_sim_over_several_shards_writes_the_same_file(pkg, tmp_path):
    """gpsbb-sim -G N: the C program renders through the node driver (pwrite sink for a file, ordered fwrite for a pipe);
    the file equals the single-handle -F output and the golden blocks."""
    pkg.build_frontend()
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "gpsbb-sim")
    z = np.load(os.path.join(GOLDEN, "static_F.npz"))
    nsamp = int(z["nsamp"])
    common = [exe, "-e", os.path.join(GOLDEN, "synth3540.14n"), "-l", "30.286502,120.032669,100", "-s", "2600000", "-d", "30.1"]
    ref = str(tmp_path / "one.bin")
    subprocess.run(common + ["-F", "-o", ref], check=True, stderr=subprocess.DEVNULL)
    want = np.fromfile(ref, np.int16)
    for k, blk in enumerate(int(b) for b in z["blocks"]):
        assert sha(want.reshape(-1, nsamp, 2)[blk]) == str(z["iq_sha256"][k]), blk
    for n in (1, 3):
        # -G N: the stream fed incrementally (gpsbb_node_begin / _feed / _end); -C: contiguous shards, everything up front
        for extra in ([], ["-C"]):
            out = str(tmp_path / ("g%d%s.bin" % (n, "c" if extra else "")))
            subprocess.run(common + ["-G", str(n), "-g", ",".join(["0"] * n), "-o", out] + extra, check=True, stderr=subprocess.DEVNULL)
            assert (np.fromfile(out, np.int16) == want).all(), (n, extra)
    piped = subprocess.run(common + ["-G", "3", "-I", "-g", "0,0,0", "-o", "-"], check=True, stderr=subprocess.DEVNULL, stdout=subprocess.PIPE).stdout
    assert np.frombuffer(piped, np.int16).tobytes() == want.tobytes()   # interleaved slots, ordered pipe
    piped = subprocess.run(common + ["-G", "2", "-g", "0,0", "-o", "-"], check=True, stderr=subprocess.DEVNULL, stdout=subprocess.PIPE).stdout
    assert np.frombuffer(piped, np.int16).tobytes() == want.tobytes()
    piped = subprocess.run(common + ["-G", "2", "-C", "-g", "0,0", "-o", "-"], check=True, stderr=subprocess.DEVNULL, stdout=subprocess.PIPE).stdout
    assert np.frombuffer(piped, np.int16).tobytes() == want.tobytes()


def test_gpsbb_sim_feeds_the_node_in_bounded_memory(pkg, tmp_path):
    """gpsbb-sim -G: the front end runs a queue ahead of the rings and no further — the process's peak memory does not grow with
    the duration (it did: all descriptors up front, 296 bytes x channels per block).  Forty times the signal (40 minutes more: + 830 MB
    of descriptors, were they made up front), the same peak RSS but for the runtime's own one-off step; the first minute of both outputs is the same bytes (a pipe, so that nothing is kept on disk)."""
    import resource
    pkg.build_frontend()
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "gpsbb-sim")
    common = [exe, "-e", os.path.join(GOLDEN, "synth3540.14n"), "-l", "30.286502,120.032669,100", "-s", "2600000", "-n", "26000",
              "-G", "2", "-g", "0,0", "-o", "-"]

    def run(seconds, keep):
        cmd = "exec %s -d %s | head -c %d | sha256sum; true" % (" ".join(common), seconds, keep)
        # peak RSS of the generator: /usr/bin/time is not in the image; a wrapper that reports ru_maxrss of its child
        code = ("import resource,subprocess,sys,hashlib\n"
                "p=subprocess.Popen(%r+['-d',%r],stdout=subprocess.PIPE,stderr=subprocess.DEVNULL)\n"
                "h=hashlib.sha256(); n=0\n"
                "while True:\n"
                "    b=p.stdout.read(1<<20)\n"
                "    if not b: break\n"
                "    if n<%d: h.update(b[:%d-n])\n"
                "    n+=len(b)\n"
                "p.wait()\n"
                "print(h.hexdigest(), n, resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss, p.returncode)\n") % (common, str(seconds), keep, keep)
        del cmd
        out = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        dig, n, rss_kb, rc = out.stdout.split()
        assert int(rc) == 0
        return dig, int(n), int(rss_kb)
    keep = 600 * 26000 * 4          # the first minute
    d1, n1, rss1 = run(600, keep)
    d2, n2, rss2 = run(24000, keep)
    assert n1 == 6000 * 26000 * 4 and n2 == 240000 * 26000 * 4 and d1 == d2
    # kB.  With all descriptors up front (-C) 234 000 blocks more are + 830 MB (12 ch x 296 B each).  Fed as it goes the peak does
    # not move with the duration — but for ONE step of ~190 MB that the runtime takes when the last lazily created streams and
    # scratch buffers of the handles come to exist, which a short run may or may not reach (measured: 2 947 300 or 3 141 800 kB at
    # 300 s, 3 133 600 .. 3 150 100 from 1 500 s on): hence the margin
    assert abs(rss2 - rss1) < 300 * 1024, (rss1, rss2)


def test_the_drivers_own_digest_sink(pkg, synth, oracle):
    """gpsbb_node_run_digest: the driver's own sink — every slot digested on the GPU that rendered it by the shard's producer
    thread (gpsbb_slot_digest: the ring behind the slot keeps rendering), no callback in the data path.  The digests equal the
    oracle's bytes' (block_digest_host) for 1, 2 and 3 shards, contiguous and interleaved, device-only rings and rings in host
    memory (digested on the host by the same threads); gpsbb_slot_digest equals gpsbb_device_digest on a popped slot."""
    nch, fs, nsamp, nb, bps = 7, 4.092e6, 40000, 24, 4
    ch = pkg.synth_descriptors(nb, nch=nch, seed=8181)
    ch["prn"][10:, 2] = 21                       # a re-allocated channel inside the stream
    want_iq, _, _ = oracle.fill_blocks(ch, 1.0 / fs, nsamp, chain=True)
    want = pkg.block_digest_host(want_iq)
    for nshards in (1, 2, 3):
        for flags in (pkg.NODE_DEVICE_ONLY, pkg.NODE_DEVICE_ONLY | pkg.NODE_INTERLEAVED, 0, pkg.NODE_INTERLEAVED):
            with pkg.Node(nshards, nch, 1.0 / fs, nsamp, bps, depth=2, flags=flags, devices=node_devices(nshards)) as node:
                for _ in range(2):   # twice on one node: the rings are kept
                    st, digs = node.run_digest(ch)
                    assert st["blocks"] == nb and (digs == want).all(), (nshards, flags)
                # and an ordinary run on the same node afterwards still delivers in order
                sink = Collect(nb, nsamp)
                if not (flags & pkg.NODE_DEVICE_ONLY):
                    node.run(ch, sink)
                    assert (sink.iq == want_iq).all() and [c[0] for c in sink.calls] == sorted(c[0] for c in sink.calls)
    st_ = synth.stream(nch, 1.0 / fs, nsamp, bps, depth=2, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
    st_.push(ch[:bps])
    st_.push(ch[bps:2 * bps])
    dptr, _ = st_.pop(copy=False)
    a = synth.slot_digest(dptr, bps, nsamp)
    assert (a == synth.device_digest(dptr, bps, nsamp)).all() and (a == want[:bps]).all()
    st_.close()


@pytest.mark.gpu
def test_a_sink_of_the_hosts_own_takes_the_digests_the_slots_were_rendered_with(pkg, oracle):
    """GPSBB_NODE_DIGESTS: every push carries the digests of its blocks (the synthesis kernel adds them up as it renders) and
    gpsbb_node_slot_digests hands them to a sink from inside its callback — equal to the digests of the oracle's bytes for 1 and 2
    shards, contiguous and interleaved, rings in HBM and in host memory (where the sink also sees the bytes); outside a sink, and on
    a node without the flag, the call is refused."""
    nch, fs, nsamp, bps, nb = 9, 25e6, 30011, 4, 24
    ch = pkg.synth_descriptors(nb, nch=nch, seed=99)
    want_iq, _, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=True)
    want = pkg.block_digest_host(want_iq)
    for nshards in (1, 2):
        for layout in (0, pkg.NODE_INTERLEAVED):
            for ring in (pkg.NODE_DEVICE_ONLY, 0):
                with pkg.Node(nshards, nch, 1 / fs, nsamp, bps, depth=3, flags=layout | ring | pkg.NODE_DIGESTS | pkg.NODE_INDEXED, devices=[0] * nshards) as nd:
                    got = np.zeros(nb, np.uint64)

                    def sink(iq, first, n, shard):
                        got[first:first + n] = nd.slot_digests(shard, n)
                        if not ring:
                            import ctypes as C
                            seen_iq = np.frombuffer((C.c_int16 * (n * nsamp * 2)).from_address(iq), np.int16).reshape(n, nsamp, 2)
                            assert (seen_iq == want_iq[first:first + n]).all()
                        return 0
                    nd.run(ch, sink)
                    assert (got == want).all(), (nshards, layout, ring)
                    with pytest.raises(pkg.GpsbbError):
                        nd.slot_digests(0, bps + 1)
    with pkg.Node(1, nch, 1 / fs, nsamp, bps, depth=3, flags=pkg.NODE_DEVICE_ONLY | pkg.NODE_INDEXED, devices=[0]) as nd:
        seen = []

        def sink2(iq, first, n, shard):
            try:
                nd.slot_digests(shard, n)
                seen.append("handed out")
            except pkg.GpsbbError:
                seen.append("refused")
            return 0
        nd.run(ch[:bps * 2], sink2)
        assert seen == ["refused", "refused"]
