#!/usr/bin/env python3
"""PCIe-inclusive rate of the streaming surface (gpsbb_stream_*): descriptors in, int16 IQ out in pinned
host memory, D2H gather on the side stream overlapped with the next slot's kernels.  Not bench.py's
`value` (which keeps the IQ in HBM); quoted in DESIGN.md."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    nch, fs, nsamp, bps, depth, nslots = 16, 25e6, 2500000, 16, 3, 24
    for a in sys.argv[1:]:  # --depth=N --bps=N --slots=N
        if a.startswith("--depth="):
            depth = int(a.split("=")[1])
        if a.startswith("--bps="):
            bps = int(a.split("=")[1])
        if a.startswith("--slots="):
            nslots = int(a.split("=")[1])
    ch = pkg.synth_descriptors(bps * nslots, nch=nch, seed=0x5EED)
    with pkg.Synth(0) as s:
        flags = pkg.CHAIN_CARRIER if "--chain" in sys.argv else 0
        st = s.stream(nch, 1.0 / fs, nsamp, bps, depth=depth, flags=flags)
        # warm-up
        for k in range(depth):
            st.push(ch[k * bps:(k + 1) * bps])
        while st.pending:
            st.pop(copy=False)
        t0 = time.perf_counter()
        pushed = popped = 0
        while popped < nslots:
            while pushed < nslots and st.pending < depth:
                st.push(ch[pushed * bps:(pushed + 1) * bps])
                pushed += 1
            st.pop(copy=False)
            popped += 1
        dt = time.perf_counter() - t0
        st.close()
    samples = nslots * bps * nsamp
    print(json.dumps({"stream_samples_per_s": samples / dt, "GBps_to_host": samples * 4 / dt / 1e9,
                      "chained_carrier": bool(flags), "slot_blocks": bps, "depth": depth, "slots": nslots, "seconds": dt}))


if __name__ == "__main__":
    main()
