#!/usr/bin/env python3
"""Where the drop-in call's time goes: run under rocprofv3 --kernel-trace --memory-copy-trace (tools/fill_timeline.sh), this makes
40 gpsbb_fill_block_ref calls of the reference's block (12 ch, 2.6 MS/s, 300 000 samples) and prints the wall clock of each;
the .sh then lists the last call's kernels and copies on one time axis."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
nch, fs, nsamp = 12, 2.6e6, 300000
d = pkg.synth_descriptors(8, nch=nch, seed=0xF00D)
iq = np.zeros((nsamp, 2), np.int16)
lay = pkg.ref_layout()
if os.environ.get("FILL_PIN"):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipHostRegister(ctypes.c_void_p(iq.ctypes.data), ctypes.c_size_t(iq.nbytes), 0)
    print("hipHostRegister", rc)
with pkg.Synth(0) as s:
    if os.environ.get("FILL_REG"):
        s.host_register(iq)
    ts = []
    for k in range(int(os.environ.get("FILL_CALLS", "40"))):
        chan, gain = pkg.ref_channels(d[k % 8])
        t0 = time.perf_counter()
        s.fill_block_ref(chan, gain, 1.0 / fs, nsamp, iq, lay)
        ts.append(time.perf_counter() - t0)
    import hashlib
    print("sha of the last block", hashlib.sha256(iq.tobytes()).hexdigest()[:16])
    print("calls (us):", " ".join("%.0f" % (t * 1e6) for t in ts[-12:]))
    tt = sorted(ts[len(ts) // 5:])
    print("median %.1f us  min %.1f  p90 %.1f  (%d calls)" % (tt[len(tt) // 2] * 1e6, tt[0] * 1e6, tt[int(len(tt) * 0.9)] * 1e6, len(tt)))
