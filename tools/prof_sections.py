#!/usr/bin/env python3
"""Where a k_synth wavefront's wall time goes: section timers compiled in with -DGPSBB_PROF.

    make -C pluto-gps-sim_amd/csrc EXTRA=-DGPSBB_PROF   (then rebuild without it!)
    python tools/prof_sections.py [--blocks 100] [--synth-only]

Each wavefront sums the wall cycles (s_memtime) between marks; the sums over all wavefronts are printed as
fractions of the wavefronts' total lifetime.  With 4 wavefronts per SIMD a section's wall time includes the
time its wavefront was not issuing, so the fractions show where wavefronts wait, not instruction counts."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (first: the HIP runtime)
from __graft_entry__ import load_package

NAMES = ["table staging (per workgroup)", "chunk start (atomic, tile index, first rows)",
         "channel prelude (row lookup, wrap tests)", "staging of the next tile's rows (issue)", "walk",
         "wait for the next tile's rows", "stores"]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=100)
    ap.add_argument("--fs", type=float, default=25e6)
    ap.add_argument("--nsamp", type=int, default=2500000)
    ap.add_argument("--nch", type=int, default=16)
    ap.add_argument("--synth-only", action="store_true")
    a = ap.parse_args()
    pkg = load_package()
    lib = pkg.lib()
    lib.gpsbb_test_read_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    lib.gpsbb_test_read_prof.restype = C.c_int
    s = pkg.Synth(0)
    ch = pkg.synth_descriptors(a.blocks, a.nch)
    b = pkg.Batch(s, ch, 1.0 / a.fs, a.nsamp)
    for _ in range(3):
        b.run()
    s.sync()
    out = (C.c_ulonglong * 16)()
    lib.gpsbb_test_read_prof(s._h, out, 1)
    if a.synth_only:
        lib.gpsbb_test_skip_seed(1)
    for _ in range(4):
        b.run()
    s.sync()
    lib.gpsbb_test_skip_seed(0)
    print("k_synth %.3f ms (last run)" % b.timing()["ms_synth"])
    lib.gpsbb_test_read_prof(s._h, out, 1)
    v = np.array(list(out), dtype=np.float64)
    if v[15] == 0:
        raise SystemExit("library was not built with -DGPSBB_PROF")
    for k in range(7):
        print("%-48s %6.2f %%" % (NAMES[k], 100.0 * v[k] / v[15]))
    print("%-48s %6.2f %%" % ("unattributed", 100.0 * (v[15] - v[:7].sum()) / v[15]))
    ntile = max(v[10], 1.0)
    print("tiles %d, staged per channel (rows exceed the slice) %.1f %%, wall cycles per tile %.0f"
          % (ntile, 100.0 * v[11] / ntile, v[15] / ntile))


if __name__ == "__main__":
    main()
