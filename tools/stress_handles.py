#!/usr/bin/env python3
"""Several producer threads, each with handles of its own on ONE GPU, created and destroyed over and over: the drop-in call
(gpsbb_fill_block: 12 ch, 2.6 MS/s), a chained batch on the breakpoint kernel (16 ch, 25 MS/s, device-side chain) and a short
chained stream — every result against what one quiet handle produced first.
usage: tools/stress_handles.py [threads] [rounds]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def work(pkg, cases, ref, rounds, tid, out):
    bad = 0
    for r in range(rounds):
        with pkg.Synth(0) as s:
            ch1, d1, n1 = cases["fill"]
            iq, st = s.fill_block(ch1, d1, n1)
            if not (iq == ref["fill"]).all():
                bad += 1
                out.append((tid, r, "fill"))
            chb, db, nb = cases["batch"]
            b = s.batch(chb, db, nb, flags=pkg.CHAIN_CARRIER)
            b.run()
            s.sync()
            iqb, _ = b.read()
            b.close()
            if not (iqb == ref["batch"]).all():
                bad += 1
                out.append((tid, r, "batch", [k for k in range(iqb.shape[0]) if not (iqb[k] == ref["batch"][k]).all()][:8]))
            chs, ds, ns, bps = cases["stream"]
            st = s.stream(chs.shape[1], ds, ns, bps, depth=2, flags=pkg.CHAIN_CARRIER)
            got = []
            npush = chs.shape[0] // bps
            pushed = popped = 0
            while popped < npush:
                while pushed < npush and st.pending < 2:
                    st.push(chs[pushed * bps:(pushed + 1) * bps])
                    pushed += 1
                iqs, _ = st.pop()
                got.append(iqs)
                popped += 1
            st.close()
            got = np.concatenate(got)
            if not (got == ref["stream"]).all():
                bad += 1
                out.append((tid, r, "stream", [k for k in range(got.shape[0]) if not (got[k] == ref["stream"][k]).all()][:8]))
    return bad


def main():
    nthr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    pkg = load_package()
    cases = {
        "fill": (pkg.synth_descriptors(1, nch=12, seed=11)[0], 1 / 2.6e6, 300000),
        "batch": (pkg.synth_descriptors(24, nch=16, seed=12), 1 / 25e6, 200000),
        "stream": (pkg.synth_descriptors(48, nch=12, seed=13), 1 / 2.6e6, 100000, 8),
    }
    ref = {}
    out = []
    one = {"fill": None, "batch": None, "stream": None}

    class Grab(list):
        pass
    # the reference: one quiet handle
    with pkg.Synth(0) as s:
        ref["fill"], _ = s.fill_block(*cases["fill"])
        b = s.batch(cases["batch"][0], cases["batch"][1], cases["batch"][2], flags=pkg.CHAIN_CARRIER)
        b.run()
        s.sync()
        ref["batch"], _ = b.read()
        b.close()
        chs, ds, ns, bps = cases["stream"]
        st = s.stream(chs.shape[1], ds, ns, bps, depth=2, flags=pkg.CHAIN_CARRIER)
        got = []
        for k in range(chs.shape[0] // bps):
            st.push(chs[k * bps:(k + 1) * bps])
            got.append(st.pop()[0])
        st.close()
        ref["stream"] = np.concatenate(got)
    th = [threading.Thread(target=work, args=(pkg, cases, ref, rounds, t, out)) for t in range(nthr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for o in out[:20]:
        print("differs:", o)
    print("%d differing results in %d threads x %d rounds x 3 cases" % (len(out), nthr, rounds))


if __name__ == "__main__":
    main()
