import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from __graft_entry__ import load_package
pkg = load_package()
import bench
gb = int(os.environ.get("GB", "32")); depth = int(os.environ.get("DEPTH", "5")); nsl = int(os.environ.get("NSL", "32"))
flags = pkg.CHAIN_CARRIER if os.environ.get("CHAIN", "1") == "1" else 0
ch = bench.stream_descriptors(pkg, gb * 16, 16)
with pkg.Synth(0) as s:
    st = s.stream(16, 1 / 25e6, 2500000, gb, depth=depth, flags=flags)
    def run(n):
        pushed = popped = 0
        while popped < n:
            while pushed < n and st.pending < depth:
                k = pushed % 16
                st.push(ch[k * gb:(k + 1) * gb]); pushed += 1
            st.pop(copy=False); popped += 1
    run(depth)
    t0 = time.perf_counter(); run(nsl); dt = time.perf_counter() - t0
    print("gb %d depth %d chained %d: %.1f GB/s to host, %.2f ms per slot" % (gb, depth, flags != 0, nsl * gb * 2500000 * 4 / dt / 1e9, dt / nsl * 1e3))
    st.close()
