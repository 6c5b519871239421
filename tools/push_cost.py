#!/usr/bin/env python3
"""host time of gpsbb_stream_push (400-block pushes, chained on the device, HBM-only ring)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from __graft_entry__ import load_package
pkg = load_package()
import bench
PB = int(os.environ.get("PB", "400"))
NP = int(os.environ.get("NP", "48"))
D = int(os.environ.get("DEPTH", "8"))
ch = bench.stream_descriptors(pkg, PB * NP, 16)
with pkg.Synth(0) as s:
    st = s.stream(16, 1 / 25e6, 2500000, PB, depth=D, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
    tp, tq = [], []
    t_all = time.perf_counter()
    for k in range(NP):
        if st.pending >= D:
            t0 = time.perf_counter(); st.pop(copy=False); tq.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); st.push(ch[k * PB:(k + 1) * PB]); tp.append(time.perf_counter() - t0)
    while st.pending:
        st.pop(copy=False)
    s.sync()
    t_all = time.perf_counter() - t_all
    print("push ms:", " ".join("%.2f" % (t * 1e3) for t in tp))
    print("pop  ms:", " ".join("%.2f" % (t * 1e3) for t in tq))
    print("total %.1f ms for %d pushes of %d blocks: %.2f ms per push; host in push %.1f, in pop %.1f" % (t_all * 1e3, NP, PB, t_all * 1e3 / NP, sum(tp) * 1e3, sum(tq) * 1e3))
    st.close()
