#!/usr/bin/env python3
"""The headline workload's carrier chain against the host: the first N pushes of bench.py's stream through the ring,
the end-of-block carrier phases the device chain produced compared bit for bit with gpsbb_chain_carrier_host's exact
walk of the same descriptors (block b's end = block b+1's start).  python tools/verify_bench_chain.py [pushes]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from __graft_entry__ import load_package
pkg = load_package()
import bench
npush = int(sys.argv[1]) if len(sys.argv) > 1 else 20
PB, nch, delt, nsamp = 400, 16, 1 / 25e6, 2500000
ch = bench.stream_descriptors(pkg, PB * npush + 1, nch)
starts = pkg.chain_carrier_host(ch, delt, nsamp)          # [block, channel]: exact carr_phase at the start of every block
with pkg.Synth(0) as s:
    st = s.stream(nch, delt, nsamp, PB, depth=8, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)
    ends = []
    for k in range(npush):
        if st.pending == 8:
            ends.append(st.pop(copy=False)[1])
        st.push(ch[k * PB:(k + 1) * PB])
    while st.pending:
        ends.append(st.pop(copy=False)[1])
    st.close()
    on_dev, fb, ties = s.info(pkg.INFO_CHAIN_ON_DEVICE), s.info(pkg.INFO_CHAIN_FALLBACKS), s.info(pkg.INFO_CHAIN_TIES)
ends = np.concatenate(ends)["carr_phase"]
want = starts[1:PB * npush + 1]
bad = np.argwhere(ends.view(np.uint64) != want.view(np.uint64))
print("%d blocks x %d channels of the bench stream: chained on the device %d, %d end phases differ from the host's exact chain "
      "(blocks walked sequentially %d, ties recorded %d)" % (PB * npush, nch, on_dev, len(bad), fb, ties))
if len(bad):
    print("first:", bad[0], ends[tuple(bad[0])], want[tuple(bad[0])])
    sys.exit(1)
