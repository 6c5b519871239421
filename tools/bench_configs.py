#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs other than the headline one, end to end from the RINEX file:
host front end (libgpsfe) -> a chained stream of pushes on one MI355X (IQ left in HBM).
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


def run(pkg, synth, name, nav, motion, max_chan, fs, nsamp, nblocks, push_blocks):
    """front end on the host, then the descriptors through a chained stream (IQ left in HBM): the carrier is chained
    inside the stream — on the device where the breakpoint kernel takes the push, on host threads otherwise"""
    t0 = time.perf_counter()
    fe = pkg.FrontEnd(os.path.join(GOLD, nav), llh=SITE, motion=os.path.join(GOLD, motion) if motion else None,
                      max_chan=max_chan)
    ch = fe.generate(nblocks)
    fe.close()
    t_fe = time.perf_counter() - t0
    depth = 6
    st = synth.stream(ch.shape[1], 1.0 / fs, nsamp, push_blocks, depth=depth, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)

    def render():
        pushed = popped = 0
        n = nblocks // push_blocks
        while popped < n:
            while pushed < n and st.pending < depth:
                st.push(ch[pushed * push_blocks:(pushed + 1) * push_blocks])
                pushed += 1
            st.pop(copy=False)
            popped += 1
        synth.sync()

    render()  # once before the clock starts: device buffers are allocated on first use
    t0 = time.perf_counter()
    render()
    dt = time.perf_counter() - t0
    on_dev = synth.info(pkg.INFO_CHAIN_ON_DEVICE)
    st.close()
    return {"config": name, "channels": int((ch["prn"][0] > 0).sum()), "fs": fs, "nsamp_per_block": nsamp,
            "blocks": nblocks, "signal_seconds": nblocks * 0.1, "front_end_s": t_fe, "render_s": dt,
            "carrier_chain": "device" if on_dev else "host threads inside push()",
            "render_iq_samples_per_s": nblocks * nsamp / dt,
            "end_to_end_iq_samples_per_s": nblocks * nsamp / (dt + t_fe), "x_realtime_end_to_end": nblocks * 0.1 / (dt + t_fe)}


def run_overlapped(pkg, synth, nav, motion, max_chan, fs, nsamp, nblocks, push_blocks, span):
    """the same work with the front end as a producer thread `span` blocks ahead of the pushes (what gpsbb-sim's loop does:
    the ring is asynchronous, the host builds the next descriptors while the GPU renders): RINEX file -> IQ in HBM, one clock"""
    import queue
    import threading
    depth = 6
    # the ring and its device buffers exist before the clock starts (as in run(): allocated on first use), warmed by one
    # ring's worth of pushes from a front end of their own
    fe0 = pkg.FrontEnd(os.path.join(GOLD, nav), llh=SITE, motion=os.path.join(GOLD, motion) if motion else None,
                       max_chan=max_chan)
    ch0 = fe0.generate(depth * push_blocks)
    fe0.close()
    st = synth.stream(ch0.shape[1], 1.0 / fs, nsamp, push_blocks, depth=depth, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
    for k in range(depth):
        st.push(ch0[k * push_blocks:(k + 1) * push_blocks])
    for k in range(depth):
        st.pop(copy=False)
    synth.sync()
    st.reset()
    t0 = time.perf_counter()
    fe = pkg.FrontEnd(os.path.join(GOLD, nav), llh=SITE, motion=os.path.join(GOLD, motion) if motion else None,
                      max_chan=max_chan)
    q = queue.Queue(maxsize=8)

    def produce():
        for _ in range(nblocks // span):
            q.put(fe.generate(span))
        q.put(None)

    th = threading.Thread(target=produce)
    th.start()
    pushed = popped = 0
    while True:
        ch = q.get()
        if ch is None:
            break
        for k in range(span // push_blocks):
            if st.pending >= depth:
                st.pop(copy=False)
                popped += 1
            st.push(ch[k * push_blocks:(k + 1) * push_blocks])
            pushed += 1
    while popped < pushed:
        st.pop(copy=False)
        popped += 1
    synth.sync()
    dt = time.perf_counter() - t0
    th.join()
    fe.close()
    st.close()
    return {"end_to_end_overlapped_s": dt, "end_to_end_overlapped_iq_samples_per_s": nblocks * nsamp / dt,
            "x_realtime_overlapped": nblocks * 0.1 / dt, "front_end_span_blocks": span}


def run_node_feed(pkg, nav, motion, max_chan, fs, nsamp, nblocks, push_blocks, span, nshards=1):
    """RINEX file -> the PRODUCT's node driver, fed incrementally (gpsbb_node_begin / _feed / _end: the front end one queue ahead
    of the rings, memory independent of the duration), IQ left in HBM, one clock: what `gpsbb-sim -G` does"""
    fe0 = pkg.FrontEnd(os.path.join(GOLD, nav), llh=SITE, motion=os.path.join(GOLD, motion) if motion else None, max_chan=max_chan)
    warm = fe0.generate(4 * push_blocks)
    fe0.close()
    flags = pkg.NODE_DEVICE_ONLY | pkg.NODE_INDEXED | pkg.NODE_CONCURRENT
    with pkg.Node(nshards, warm.shape[1], 1.0 / fs, nsamp, push_blocks, depth=6, flags=flags, devices=[0] * nshards) as nd:
        nd.run(warm, lambda *a: 0)   # rings allocated, kernels loaded
        t0 = time.perf_counter()
        fe = pkg.FrontEnd(os.path.join(GOLD, nav), llh=SITE, motion=os.path.join(GOLD, motion) if motion else None, max_chan=max_chan)
        nd.begin(lambda *a: 0)
        for _ in range(nblocks // span):
            nd.feed(fe.generate(span))
        st = nd.end()
        dt = time.perf_counter() - t0
        fe.close()
    return {"node_feed_s": dt, "node_feed_iq_samples_per_s": nblocks * nsamp / dt, "x_realtime_node_feed": nblocks * 0.1 / dt,
            "node_feed_shards": nshards, "node_feed_chain_s": st["shards"][0]["seed_seconds"], "node_feed_blocks": st["blocks"]}


NB_F = 24000  # 40 minutes of signal: long enough for the ring to reach its steady state (a 3000-block run is over in 13 ms)


def main():
    pkg = load_package()
    pkg.build_frontend()
    out = []
    with pkg.Synth(0) as s:
        if len(sys.argv) > 1:  # e.g. 3: stream pushes keep pass A (gpsbb.h, GPSBB_OPT_CHAIN_WHERE)
            s.set_option(pkg.OPT_CHAIN_WHERE, int(sys.argv[1]))
        out.append(run(pkg, s, "1/2 static, 2.6 MS/s, reference block (300000 samples)", "synth3540.14n", None, 12,
                       2.6e6, 300000, NB_F, 1000))
        out[-1].update(run_overlapped(pkg, s, "synth3540.14n", None, 12, 2.6e6, 300000, NB_F, 1000, 1000))
        out[-1].update(run_node_feed(pkg, "synth3540.14n", None, 12, 2.6e6, 300000, NB_F, 1000, 1000))
        out.append(run(pkg, s, "4 user motion (10 Hz), 2.6 MS/s", "synth3540.14n", "circle_motion.csv", 12, 2.6e6,
                       300000, NB_F, 1000))
        out[-1].update(run_overlapped(pkg, s, "synth3540.14n", "circle_motion.csv", 12, 2.6e6, 300000, NB_F, 1000, 1000))
        out[-1].update(run_node_feed(pkg, "synth3540.14n", "circle_motion.csv", 12, 2.6e6, 300000, NB_F, 1000, 1000))
        out.append(run(pkg, s, "3 geometry through the front end: 16 ch, 25 MS/s, 2.5 M-sample blocks", "dense3540.14n",
                       None, 16, 25e6, 2500000, 400, 100))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
