#!/usr/bin/env python3
"""Replay descriptors a fuzz campaign saved (gpurun_out/fuzz*_fail_ch.npy) as ONE batch through a given pre-pass, against the CPU
oracle: which blocks differ, which channels' end states, and — every channel rendered alone — which channel it is.

    python tools/replay_fail.py <ch.npy> <fs> <nsamp> [--chain] [--where 3] [--first B0 --count NB]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npy")
    ap.add_argument("fs", type=float)
    ap.add_argument("nsamp", type=int)
    ap.add_argument("--chain", action="store_true")
    ap.add_argument("--where", type=int, default=3)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=0)
    ap.add_argument("--alone", action="store_true", help="every channel alone as well")
    ap.add_argument("--chan", type=int, default=-1, help="this channel only")
    a = ap.parse_args()
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    import oracle_binding as ob
    pkg = load_package()
    oracle = ob.Oracle()
    ch = np.load(a.npy)
    if a.npy.endswith(".npz"):
        ch = ch["ch"]
    if a.count:
        ch = ch[a.first:a.first + a.count].copy()
    if a.chan >= 0:
        ch = ch[:, a.chan:a.chan + 1].copy()
    nb, nch = ch.shape
    flags = pkg.CHAIN_CARRIER if a.chain else 0
    with pkg.Synth(0) as synth:
        def render(c, where):
            synth.set_option(pkg.OPT_SEED_WHERE, where)
            b = synth.batch(c, 1.0 / a.fs, a.nsamp, flags=flags)
            b.run(); synth.sync()
            iq, st = b.read(); b.close()
            return iq.reshape(len(c), -1), st, synth.info(pkg.INFO_PREPASS), synth.info(pkg.INFO_LAST_KERNEL)
        want_iq, want_st, _ = oracle.fill_blocks(ch, 1.0 / a.fs, a.nsamp, chain=a.chain, fixed=False)
        want_iq = want_iq.reshape(nb, -1)
        for where in (a.where, 1):
            iq, st, pp, kern = render(ch, where)
            bad = np.argwhere((iq != want_iq).any(axis=1))[:, 0]
            print("where %d: pre-pass %d kernel %d; repairs %d rewalked %d; blocks whose IQ differs: %d %r" %
                  (where, pp, kern, synth.info(pkg.INFO_CHAIN_REPAIRS), synth.info(pkg.INFO_CHAIN_FALLBACKS), len(bad), bad[:16].tolist()))
            for b in bad[:3]:
                e = np.argwhere(iq[b] != want_iq[b])[:, 0]
                print("   block %d: %d elements differ, first %r last %r (samples %d .. %d)" % (b, len(e), e[:6].tolist(), e[-3:].tolist(), e[0] // 2, e[-1] // 2))
            act = ch["prn"] > 0
            for f in ("carr_phase", "code_phase", "iword", "ibit", "icode", "dataBit", "codeCA"):
                d = np.argwhere((st[f] != want_st[f]) & act)
                if len(d):
                    print("   end state %s differs in %d block-channels, first %r: got %r want %r" %
                          (f, len(d), d[0].tolist(), st[f][tuple(d[0])], want_st[f][tuple(d[0])]))
        if a.alone:
            for i in range(nch):
                c1 = ch[:, i:i + 1].copy()
                w_iq, w_st, _ = oracle.fill_blocks(c1, 1.0 / a.fs, a.nsamp, chain=a.chain, fixed=False)
                iq, st, pp, kern = render(c1, a.where)
                bad = np.argwhere((iq != w_iq.reshape(nb, -1)).any(axis=1))[:, 0]
                dcp = np.argwhere((st["carr_phase"] != w_st["carr_phase"]) & (c1["prn"] > 0))
                if len(bad) or len(dcp):
                    b0 = int(bad[0]) if len(bad) else int(dcp[0][0])
                    print("channel %d alone (pre-pass %d kernel %d): %d blocks differ %r; carr end differs in %d; at block %d: prn %d f_carr %r "
                          "carr_phase %r code_phase %r f_code %r" % (i, pp, kern, len(bad), bad[:10].tolist(), len(dcp), b0, c1["prn"][b0, 0],
                                                                     c1["f_carr"][b0, 0], c1["carr_phase"][b0, 0], c1["code_phase"][b0, 0], c1["f_code"][b0, 0]))


if __name__ == "__main__":
    main()
