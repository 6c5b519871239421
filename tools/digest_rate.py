import sys, time
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
import numpy as np
pkg = load_package()
import bench
ch = bench.stream_descriptors(pkg, 400, 16)
with pkg.Synth(0) as s:
    b = s.batch(ch, 1 / 25e6, 2500000, flags=pkg.CHAIN_CARRIER)
    b.run(); s.sync()
    d = b.device_iq()
    for k in range(3):
        t0 = time.perf_counter(); a = s.slot_digest(d, 400, 2500000); dt = time.perf_counter() - t0
        print("slot_digest of 1e9 samples: %.3f ms  (%.2f TB/s)" % (dt * 1e3, 4e9 / dt / 1e12))
    b.close()
