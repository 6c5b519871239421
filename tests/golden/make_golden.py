#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference code.

Needs oracle/_ref/ (oracle/ref/build_ref.sh: the reference's own statements compiled from
/root/reference, -O0 as in its Makefile).  Run in the build container only; the .npz files it writes
are committed and are what the tests use on the GPU box, where /root/reference does not exist.

  static_F.npz   BASELINE configs 1/2: static -l 30.286502,120.032669,100, synth3540.14n, fs 2.6 MS/s,
                 300000-sample blocks, MAX_CHAN 12; blocks 0,1,2,149,299,300 (300 = first block after the
                 30 s nav-message refresh, plutogpssim.c:2764-2772)
  motion_F.npz   config 4: user motion (circle_motion.csv, 10 Hz), otherwise as above; blocks 0,1,2,299,300
  dense_S.npz    config 3 geometry through the real front end: dense3540.14n, MAX_CHAN 16, fs 25 MS/s,
                 2.5 M-sample blocks 0,1
  static_F_fixed.npz  as static_F with the reference's fixed-point carrier NCO (`#ifndef FLOAT_CARR_PHASE`)
  loop_M2.npz    the verbatim sample loop (plutogpssim.c:2690-2756) on the seeded M2 descriptor set
                 (16 channels, fs 25 MS/s), 3 blocks of 100000 samples
  rinex3_F.npz   as static_F through the reference's RINEX 3 reader (readRinex3, c:1241-1610): synth3540_v3.rnx, -3
  toverwrite_F.npz  as static_F with -t 2014/12/21,10:00:00 -T: TOC/TOE of every ephemeris overwritten (c:2523-2553)
  motion_ref_F.npz  config 4 on its named input: the reference's own circle.csv (a copy of /root/reference/circle.csv
                 lives here as circle.csv: a data fixture), blocks 0,1,2,299,300
  swap_S.npz     dense3540.14n, MAX_CHAN 16, -t 2014/12/20,01:20:00, 2000-sample blocks around block 1500, where the
                 30 s maintenance gives channel 10 from PRN 11 (set) straight to PRN 18 (risen): allocateChannel's
                 fresh carrier phase must not be replaced by the departed satellite's

Each file holds, per kept block: the descriptors the reference's front end produced, the first 4096 IQ
samples, the SHA-256 of the whole block's IQ bytes, and the channel state after the block.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from conftest import load_package  # noqa: E402

PREFIX = 4096
SITE = ("30.286502", "120.032669", "100")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def keep(iq, desc, st, blocks):
    return dict(blocks=np.array(blocks), desc=desc[blocks], end_state=st[blocks],
                iq_prefix=iq[blocks, :PREFIX].copy(), iq_sha256=np.array([sha(iq[b]) for b in blocks]))


def main():
    assert ob.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    nav = os.path.join(HERE, "synth3540.14n")
    dense = os.path.join(HERE, "dense3540.14n")
    motion = os.path.join(HERE, "circle_motion.csv")

    iq, desc, st = ob.run_ref_sim(nav, 301, 300000, 2600000, llh=SITE, max_chan=12)
    np.savez_compressed(os.path.join(HERE, "static_F.npz"), fs=2600000, nsamp=300000,
                        **keep(iq, desc, st, [0, 1, 2, 149, 299, 300]))
    print("static_F: prns", desc["prn"][0])

    iq, desc, st = ob.run_ref_sim(nav, 301, 300000, 2600000, motion=motion, max_chan=12)
    np.savez_compressed(os.path.join(HERE, "motion_F.npz"), fs=2600000, nsamp=300000,
                        **keep(iq, desc, st, [0, 1, 2, 299, 300]))
    print("motion_F: prns", desc["prn"][0])

    iq, desc, st = ob.run_ref_sim(dense, 2, 2500000, 25000000, llh=SITE, max_chan=16)
    np.savez_compressed(os.path.join(HERE, "dense_S.npz"), fs=25000000, nsamp=2500000,
                        **keep(iq, desc, st, [0, 1]))
    print("dense_S: prns", desc["prn"][0])

    # the reference's fixed-point carrier variant (built against the header without h:12)
    iq, desc, st = ob.run_ref_sim(nav, 301, 300000, 2600000, llh=SITE, max_chan=12, opt="_fixed")
    np.savez_compressed(os.path.join(HERE, "static_F_fixed.npz"), fs=2600000, nsamp=300000,
                        **keep(iq, desc, st, [0, 1, 2, 300]))
    print("static_F_fixed: carr_phase", desc["carr_phase"][0][:3])

    v3 = os.path.join(HERE, "synth3540_v3.rnx")
    iq, desc, st = ob.run_ref_sim(v3, 301, 300000, 2600000, llh=SITE, max_chan=12, extra=("-3",))
    np.savez_compressed(os.path.join(HERE, "rinex3_F.npz"), fs=2600000, nsamp=300000, **keep(iq, desc, st, [0, 1, 300]))
    print("rinex3_F: prns", desc["prn"][0])

    iq, desc, st = ob.run_ref_sim(nav, 301, 300000, 2600000, llh=SITE, max_chan=12, extra=("-t", "2014/12/21,10:00:00", "-T"))
    np.savez_compressed(os.path.join(HERE, "toverwrite_F.npz"), fs=2600000, nsamp=300000, **keep(iq, desc, st, [0, 1, 300]))
    print("toverwrite_F: prns", desc["prn"][0])

    iq, desc, st = ob.run_ref_sim(nav, 301, 300000, 2600000, motion=os.path.join(HERE, "circle.csv"), max_chan=12)
    np.savez_compressed(os.path.join(HERE, "motion_ref_F.npz"), fs=2600000, nsamp=300000,
                        **keep(iq, desc, st, [0, 1, 2, 299, 300]))
    print("motion_ref_F: prns", desc["prn"][0])

    iq, desc, st = ob.run_ref_sim(dense, 1504, 2000, 2600000, llh=SITE, max_chan=16, extra=("-t", "2014/12/20,01:20:00"))
    assert desc["prn"][1499][10] == 11 and desc["prn"][1500][10] == 18, desc["prn"][1498:1502, 10]
    d = keep(iq, desc, st, [1498, 1499, 1500, 1501, 1502, 1503])
    d["iq_prefix"] = iq[[1498, 1499, 1500, 1501, 1502, 1503]].copy()     # the whole (short) blocks
    np.savez_compressed(os.path.join(HERE, "swap_S.npz"), fs=2600000, nsamp=2000, **d)
    print("swap_S: channel 10", desc["prn"][1498:1504, 10])

    pkg = load_package()
    ch = pkg.synth_descriptors(3, nch=16, seed=0x5EED)
    ref = ob.RefLoop()
    nsamp = 100000
    iqs, sts = [], []
    for b in range(3):
        i, s = ref.fill(ch[b], 1.0 / 25e6, nsamp)
        iqs.append(i)
        sts.append(s)
    iq, st = np.stack(iqs), np.stack(sts)
    np.savez_compressed(os.path.join(HERE, "loop_M2.npz"), fs=25000000, nsamp=nsamp,
                        **keep(iq, ch, st, [0, 1, 2]))
    print("done")


if __name__ == "__main__":
    main()
