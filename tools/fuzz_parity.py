#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: N random workloads (sample rate, block length, channels, Doppler from
mHz to the contract's limit, inactive channels, float / fixed-point carrier, chained or independent blocks,
seeding on the device or on the host), every one bit-exact against the CPU oracle or the run stops.

    python tools/fuzz_parity.py [--cases 300] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def stream_soak(a):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    import oracle_binding as ob
    pkg = load_package()
    oracle = ob.Oracle()
    rng = np.random.default_rng(a.seed)
    fallbacks = ties = 0
    prepass = {}
    with pkg.Synth(0) as synth:
        rng_dig = np.random.default_rng(a.seed ^ 0xD16E57)
        digested = 0
        for case in range(a.cases):
            fs = float(rng.choice([16.368e6, 20e6, 25e6, 30e6, 2.0 ** 25, 50e6]))
            low_rate = rng.random() < (1.0 if getattr(a, "low_rate", False) else 0.25)  # a rate below the breakpoint kernel's: k_synth_pd
            if low_rate:
                fs = float(rng.choice([1e6, 2.6e6, 4.092e6, 10e6]))
            if a.ties:                                 # f_carr * delt is then exact: the steps really have few bits
                fs, low_rate = 2.0 ** 25, False
            nch = int(rng.integers(1, 17))
            bps = int(rng.choice([4, 16, 33, 64, 100] if a.nsamp_max <= 1000000 else [2, 3, 5]))
            pushes = int(rng.integers(2, 6 if a.nsamp_max <= 1000000 else 4))
            nsamp = int(rng.integers(2000, max(2001, min(a.nsamp_max, int(a.budget / (bps * pushes * nch))))))
            nb = bps * pushes
            ch = pkg.synth_descriptors(nb, nch=nch, seed=int(rng.integers(1, 2 ** 31)))
            # Doppler: a slow drift per channel (what a real stream looks like) or independent per block
            f0 = rng.uniform(-1.0, 1.0, size=nch) * (min(fs / 2100.0, 2e4) if not low_rate else 6e3) * 10.0 ** rng.uniform(-4, 0, size=nch)
            if rng.random() < 0.6:
                drift = rng.uniform(-1e-3, 1e-3, size=nch) * np.abs(f0)
                f = f0[None, :] + drift[None, :] * np.arange(nb)[:, None]
            else:
                f = f0[None, :] * rng.uniform(0.5, 1.0, size=(nb, nch)) * np.where(rng.random((nb, nch)) < 0.1, -1.0, 1.0)
            if rng.random() < 0.3:                     # exact binary steps: ties on coarse grids
                f[:, 0] = np.sign(f[0, 0] if f[0, 0] != 0 else 1.0) * fs * 2.0 ** float(rng.integers(-20, -11))
            if a.ties:                                 # every channel: steps with few mantissa bits (sums tie often), either sign
                for i in range(nch):
                    m = float(rng.integers(1, 1 << int(rng.integers(1, 12))) | 1)
                    f[:, i] = (-1.0 if rng.random() < 0.5 else 1.0) * fs * m * 2.0 ** float(rng.integers(-24, -14))
                    if not (abs(f[0, i]) < (fs / 2100.0 if not low_rate else 6e3)):
                        f[:, i] = fs * 2.0 ** -15
            if rng.random() < 0.2:
                f[:, -1] = 0.0
            if rng.random() < 0.2:                     # no Doppler for a stretch of blocks in the middle of a chain (round 6: the laps take it)
                zb = int(rng.integers(0, nb)); f[zb:int(rng.integers(zb, nb + 1)), int(rng.integers(0, nch))] = 0.0 if rng.random() < 0.7 else -0.0
            if os.environ.get("GPSBB_FUZZ_WHERE") == "3":
                # a campaign aimed at the lap-parallel pre-pass: nothing that sends the case elsewhere (a rate only the per-sample
                # kernel renders; a carrier that does not move at all no longer does: the laps take it since round 6)
                if fs < 2e6:
                    fs = 2.6e6
            ch["f_carr"] = f
            ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
            # satellites come and go: a channel changes PRN or goes idle for a stretch of blocks
            for _ in range(int(rng.integers(0, 4))):
                i = int(rng.integers(0, nch)); b0 = int(rng.integers(0, nb)); b1 = int(rng.integers(b0, nb + 1))
                ch["prn"][b0:b1, i] = 0 if rng.random() < 0.5 else int(rng.integers(1, 33))
            delt = 1.0 / fs
            depth = int(rng.integers(2, 5))
            dev_only = bool(rng.integers(0, 2))
            # pre-pass and chain on the device: lap-parallel (round 5), by the row walks (rounds 1-4), or whatever the library picks
            where = int(os.environ.get("GPSBB_FUZZ_WHERE", rng.choice([3, 1, 0], p=[0.45, 0.4, 0.15])))
            kern = 1 if rng.random() < (0.25 if os.environ.get("GPSBB_FUZZ_WHERE") != "3" else 0.1) else 0  # now and then the per-sample kernel where the other one would do
            if getattr(a, "only", -1) >= 0 and case != a.only:         # --only: the same random numbers drawn, nothing rendered
                continue
            if os.environ.get("GPSBB_FUZZ_VERBOSE"):
                print("case", case, dict(fs=fs, nsamp=nsamp, nch=nch, bps=bps, pushes=pushes, depth=depth, dev_only=dev_only, where=where, kern=kern), flush=True)
            want_iq, want_st, _ = oracle.fill_blocks(ch, delt, nsamp, chain=True, fixed=False)
            synth.set_option(pkg.OPT_SEED_WHERE, where)
            synth.set_option(pkg.OPT_SYNTH_KERNEL, kern)
            st = synth.stream(nch, delt, nsamp, bps, depth=depth,
                              flags=pkg.CHAIN_CARRIER | (pkg.STREAM_DEVICE_ONLY if dev_only else 0))
            got = []
            k = 0
            popped = 0
            # every other push (drawn from a generator of its own: the cases stay what their seeds always made them) is rendered WITH
            # its block digests (GPSBB_PUSH_DIGEST: the synthesis kernel adds them up as it renders, or a digest kernel behind it):
            # compared with the digests of the oracle's bytes — which also puts the IQ of HBM-only rings under the check
            flagged = [bool(rng_dig.random() < 0.5) for _ in range(pushes)]
            want_dig = pkg.block_digest_host(want_iq)
            while popped < pushes:
                while k < pushes and st.pending < depth:
                    st.push(ch[k * bps:(k + 1) * bps], digest=flagged[k]); k += 1
                if flagged[popped]:
                    iq, es, dig = st.pop_digest(copy=True)
                    if not (dig == want_dig[popped * bps:(popped + 1) * bps]).all():
                        np.save("gpurun_out/fuzz_stream_fail_ch.npy", ch)
                        raise SystemExit("STREAM DIGEST MISMATCH %r push %d blocks %r" % (dict(case=case, fs=fs, nsamp=nsamp, nch=nch, bps=bps, pushes=pushes, dev_only=dev_only, where=where, kern=kern), popped,
                                                                                        np.argwhere(dig != want_dig[popped * bps:(popped + 1) * bps])[:, 0].tolist()[:10]))
                    digested += bps
                else:
                    iq, es = st.pop(copy=True)
                got.append((None if dev_only else np.asarray(iq).reshape(bps, -1), es))  # HBM-only ring: end states only
                popped += 1
            st.close()
            prepass[synth.info(pkg.INFO_PREPASS)] = prepass.get(synth.info(pkg.INFO_PREPASS), 0) + 1
            what = dict(case=case, fs=fs, nsamp=nsamp, nch=nch, bps=bps, pushes=pushes, depth=depth, dev_only=dev_only, where=where)
            if os.environ.get("GPSBB_FUZZ_VERBOSE"):
                print(what, "pre-pass", synth.info(pkg.INFO_PREPASS), "kernel", synth.info(pkg.INFO_LAST_KERNEL), "min |f_carr|", float(np.abs(f[ch["prn"] > 0]).min()) if (ch["prn"] > 0).any() else None, flush=True)
            if getattr(a, "also_batch", False):
                # the same blocks as ONE chained batch (no stream to continue: pass B starts from the host's drift model of the
                # carrier, long blocks are cut into segments) and as a batch of independent blocks seeded with the oracle's phases
                bt = synth.batch(ch, delt, nsamp, flags=pkg.CHAIN_CARRIER)
                bt.run(); synth.sync(); biq, bst = bt.read(); bt.close()
                act_all = ch["prn"] > 0
                if not (biq == want_iq).all() or bst["carr_phase"][act_all].tobytes() != want_st["carr_phase"][act_all].tobytes():
                    np.save("gpurun_out/fuzz_stream_fail_ch.npy", ch)
                    raise SystemExit("BATCH (chained) MISMATCH %r" % what)
                ind = ch.copy()
                ind["carr_phase"][1:] = np.where((ch["prn"][1:] > 0) & (ch["prn"][1:] == ch["prn"][:-1]), want_st["carr_phase"][:-1], ch["carr_phase"][1:])
                bt = synth.batch(ind, delt, nsamp)
                bt.run(); synth.sync(); biq, bst = bt.read(); bt.close()
                if not (biq == want_iq).all() or bst["carr_phase"][act_all].tobytes() != want_st["carr_phase"][act_all].tobytes():
                    np.save("gpurun_out/fuzz_stream_fail_ch.npy", ind)
                    raise SystemExit("BATCH (independent blocks) MISMATCH %r" % what)
            for j, (iq, es) in enumerate(got):
                w = want_iq[j * bps:(j + 1) * bps].reshape(bps, -1)
                if iq is not None and not (iq == w).all():
                    bad = np.argwhere(iq != w)[0]
                    np.save("gpurun_out/fuzz_stream_fail_ch.npy", ch)
                    raise SystemExit("STREAM MISMATCH %r push %d first at block %d element %d" % (what, j, bad[0], bad[1]))
                act = ch["prn"][j * bps:(j + 1) * bps] > 0
                wcp = want_st["carr_phase"][j * bps:(j + 1) * bps]
                if es["carr_phase"][act].tobytes() != wcp[act].tobytes():
                    bad = np.argwhere((es["carr_phase"] != wcp) & act)
                    np.save("gpurun_out/fuzz_stream_fail_ch.npy", ch)
                    raise SystemExit("STREAM END STATE MISMATCH %r push %d: %d block-channels, first (block, channel) %r got %r want %r; "
                                     "f_carr there %r prn column %r" %
                                     (what, j, len(bad), bad[0].tolist(), es["carr_phase"][tuple(bad[0])], wcp[tuple(bad[0])],
                                      ch["f_carr"][j * bps + bad[0][0], bad[0][1]], ch["prn"][:, bad[0][1]].tolist()))
        fallbacks = synth.info(pkg.INFO_CHAIN_FALLBACKS)
        ties = synth.info(pkg.INFO_CHAIN_TIES)
        on_dev = synth.info(pkg.INFO_CHAIN_ON_DEVICE)
        repairs = synth.info(pkg.INFO_CHAIN_REPAIRS)
    print("fuzz_parity --stream: %d chained streams bit-exact (seed %d), %d blocks' digests as rendered equal the oracle's bytes'; last push chained on the device: %d; blocks / laps walked "
          "again by the fix-up / the lap repair: %d; links / guesses that did not hold: %d; wrap ties recorded: %d; pre-pass of each case's "
          "last push {1: row walks, 2: host threads, 3: lap-parallel}: %r" % (a.cases, a.seed, digested, on_dev, fallbacks, repairs, ties, prepass))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--shapes", action="store_true", help="a few extreme shapes instead of random small ones")
    ap.add_argument("--ev", action="store_true",
                    help="only workloads the breakpoint kernel k_synth_ev is eligible for: sample rates above 16 MS/s, "
                         "Doppler up to fs/2100 (four table-index changes per run of 16 samples), IEEE carrier")
    ap.add_argument("--stream", action="store_true",
                    help="chained streams instead of batches: several pushes of many short blocks through a ring, the carrier "
                         "chained on the device from push to push (PRN changes, idle channels, steps that tie), against the "
                         "oracle's sequential render of the whole stream")
    ap.add_argument("--ties", action="store_true", help="--stream: every carrier step has only a few mantissa bits (exact ties at wraps and "
                    "binade crossings are common instead of one in thousands)")
    ap.add_argument("--low-rate", action="store_true", help="--stream: every case at 1 .. 10 MS/s (k_synth_pd / the per-sample kernels)")
    ap.add_argument("--also-batch", action="store_true", help="--stream: every case also as one chained batch and as a batch of independent blocks")
    ap.add_argument("--nsamp-max", type=int, default=200000, help="--stream: longest block")
    ap.add_argument("--budget", type=float, default=3e7, help="--stream: channel-samples per case (what the CPU oracle has to walk)")
    ap.add_argument("--only", type=int, default=-1, help="render this case only (the others' random numbers are drawn all the same)")
    a = ap.parse_args()
    if a.stream:
        return stream_soak(a)
    import torch  # noqa: F401  (first: the HIP runtime)
    from __graft_entry__ import load_package
    import oracle_binding as ob
    pkg = load_package()
    oracle = ob.Oracle()
    rng = np.random.default_rng(a.seed)
    lib = pkg.lib()
    # (fs, nsamp, nch, nblocks): one very long block, many tiny blocks, many one-sample blocks, a block one
    # sample longer than a multiple of the tile, a wide batch at the headline geometry
    shapes = [(25e6, 1 << 24, 2, 1), (2.6e6, 777, 12, 3000), (10e6, 1, 16, 5000), (25e6, 1024 * 300 + 1, 16, 3),
              (25e6, 2500000, 16, 6), (1e6, 5000000, 3, 2)]
    used = {}
    prepass = {}
    REG_SAMPLES = 400000
    reg_buf = np.zeros((REG_SAMPLES, 2), np.int16)
    dropin = 0
    with pkg.Synth(0) as synth:
        synth.host_register(reg_buf)
        for case in range(len(shapes) * 2 if a.shapes else a.cases):
            fs = float(rng.choice([1e6, 2.6e6, 3e6, 4.092e6, 10e6, 16e6, 25e6, 30e6, 2.0 ** 25, 50e6]))
            nsamp = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 120000), 1024 * int(rng.integers(1, 60))]))
            nch = int(rng.integers(1, 17))
            nblocks = int(rng.integers(1, 6))
            fixed = bool(rng.integers(0, 4) == 0)
            chain = bool(rng.integers(0, 2))
            mode = int(os.environ.get("GPSBB_FUZZ_WHERE", rng.choice([1, 2, 3])))  # 1: the row walks (k_seed / k_walk), 2: host threads, 3: lap-parallel
            kern = int(rng.integers(0, 2))            # 0: automatic (breakpoint kernel where eligible), 1: per-sample
            if a.shapes:
                fs, nsamp, nch, nblocks = shapes[case // 2]
                fixed, mode = False, (1, 3)[case % 2]
            if a.ev:
                fs = float(rng.choice([16e6, 16.368e6, 20e6, 25e6, 30e6, 2.0 ** 25, 50e6, 61.44e6]))
                nsamp = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 300000), 1024 * int(rng.integers(1, 200))]))
                fixed = False
            ch = pkg.synth_descriptors(nblocks, nch=nch, seed=int(rng.integers(1, 2 ** 31)))
            scale = 10.0 ** rng.uniform(-3, np.log10((1.0 / 2100.0 if a.ev else 0.124) * fs), size=(nblocks, nch))
            ch["f_carr"] = np.where(rng.random((nblocks, nch)) < 0.5, -1.0, 1.0) * scale
            if rng.random() < 0.3:                     # exact binary steps: phases land on boundaries
                ch["f_carr"] = np.sign(ch["f_carr"]) * fs * 2.0 ** rng.integers(-20, -11 if a.ev else -4, size=(nblocks, nch))
            if a.ev and rng.random() < 0.2:            # a channel that does not move at all, one that barely does
                ch["f_carr"][:, 0] = 0.0
                if nch > 1:
                    ch["f_carr"][:, 1] = 1e-9 * fs
            ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
            if rng.random() < 0.2:
                ch["code_phase"] = np.floor(ch["code_phase"])        # starts on chip boundaries
            if rng.random() < 0.2:
                ch["carr_phase"] = np.floor(ch["carr_phase"] * 512.0) / 512.0
            ch["prn"][rng.random((nblocks, nch)) < 0.15] = 0
            if fixed:
                ch["carr_phase"] = np.floor(ch["carr_phase"] * 2.0 ** 32)
            flags = (pkg.FIXED_CARRIER if fixed else 0) | (pkg.CHAIN_CARRIER if chain else 0)
            if a.only >= 0 and case != a.only:
                continue
            want_iq, want_st, _ = oracle.fill_blocks(ch, 1.0 / fs, nsamp, chain=chain, fixed=fixed)
            synth.set_option(pkg.OPT_SEED_WHERE, mode)
            synth.set_option(pkg.OPT_SYNTH_KERNEL, kern)
            b = synth.batch(ch, 1.0 / fs, nsamp, flags=flags)
            b.run()
            try:
                synth.sync()
            except Exception:
                np.save("gpurun_out/fuzz_fail_ch.npy", ch)
                print("FAILED in sync:", dict(case=case, fs=fs, nsamp=nsamp, nch=nch, nblocks=nblocks, fixed=fixed, chain=chain,
                                              mode=mode, kern=kern), "f_carr", ch["f_carr"].tolist(), "prn", ch["prn"].tolist(),
                      "carr_phase", ch["carr_phase"].tolist())
                raise
            iq, st = b.read()
            b.close()
            used[synth.info(pkg.INFO_LAST_KERNEL)] = used.get(synth.info(pkg.INFO_LAST_KERNEL), 0) + 1
            prepass[synth.info(pkg.INFO_PREPASS)] = prepass.get(synth.info(pkg.INFO_PREPASS), 0) + 1
            what = dict(case=case, fs=fs, nsamp=nsamp, nch=nch, nblocks=nblocks, fixed=fixed, chain=chain, mode=mode, kern=kern)
            if not (iq == want_iq).all():
                bad = np.argwhere(iq != want_iq)[0]
                np.save("gpurun_out/fuzz_fail_ch.npy", ch)
                nbad = np.argwhere((iq != want_iq).reshape(len(iq), -1).any(axis=1))[:, 0]
                raise SystemExit("MISMATCH %r first at block %d sample %d; %d blocks differ: %r; pre-pass %d kernel %d; f_carr of the block %r prn %r code_phase %r carr_phase %r" %
                                 (what, bad[0], bad[1], len(nbad), nbad[:20].tolist(), synth.info(pkg.INFO_PREPASS), synth.info(pkg.INFO_LAST_KERNEL),
                                  ch["f_carr"][bad[0]].tolist(), ch["prn"][bad[0]].tolist(), ch["code_phase"][bad[0]].tolist(), ch["carr_phase"][bad[0]].tolist()))
            act = ch["prn"] > 0
            for f in ("carr_phase", "code_phase", "iword", "ibit", "icode", "dataBit", "codeCA"):
                if st[f][act].tobytes() != want_st[f][act].tobytes():
                    raise SystemExit("END STATE MISMATCH %r field %s" % (what, f))
            # ... and the case's first block through the drop-in call: copied into a fresh buffer, then rendered into a registered one
            # (gpsbb_host_register) at an offset that changes from case to case
            if nsamp <= REG_SAMPLES - 4096:
                f0 = pkg.FIXED_CARRIER if fixed else 0
                iq1, st1 = synth.fill_block(ch[0], 1.0 / fs, nsamp, flags=f0)
                at = int(rng.integers(0, 4096))
                reg_buf[:] = 0x5a5a
                iq2, st2 = synth.fill_block(ch[0], 1.0 / fs, nsamp, flags=f0, out=reg_buf[at:at + nsamp])
                a0 = ch["prn"][0] > 0
                for nm, q, s_ in (("copied", iq1, st1), ("registered", iq2[:nsamp], st2)):
                    if not (q == want_iq[0]).all():
                        np.save("gpurun_out/fuzz_fail_ch.npy", ch)
                        raise SystemExit("MISMATCH of the drop-in call (%s) %r" % (nm, what))
                    for f in ("carr_phase", "code_phase", "iword", "ibit", "icode", "dataBit", "codeCA"):
                        if s_[f][a0].tobytes() != want_st[f][0][a0].tobytes():
                            raise SystemExit("END STATE MISMATCH of the drop-in call (%s) %r field %s" % (nm, what, f))
                if not ((reg_buf[:at] == 0x5a5a).all() and (reg_buf[at + nsamp:] == 0x5a5a).all()):
                    raise SystemExit("the drop-in call wrote outside its block %r" % (what,))
                dropin += 1
        exact_runs = synth.info(pkg.INFO_EXACT_RUNS)
        repairs, rewalked = synth.info(pkg.INFO_CHAIN_REPAIRS), synth.info(pkg.INFO_CHAIN_FALLBACKS)
        synth.host_unregister(reg_buf)
    print("fuzz_parity: %d cases bit-exact (seed %d), %d of them also as the drop-in call (copied / into a registered buffer); synthesis kernel used {1: per-sample, 2: breakpoint}: %r; "
          "lane-runs recomputed exactly by the breakpoint kernel: %d; pre-pass {1: row walks, 2: host threads, 3: lap-parallel}: %r; "
          "links that did not hold: %d, laps / blocks walked again: %d" %
          (len(shapes) * 2 if a.shapes else a.cases, a.seed, dropin, used, exact_runs, prepass, repairs, rewalked))


if __name__ == "__main__":
    main()
