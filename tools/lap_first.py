"""first contact of the lap-parallel pre-pass with the GPU: a few workloads against the oracle, with the pre-pass forced"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import __graft_entry__ as g
import oracle_binding as ob
pkg = g.load_package()
orc = ob.Oracle()
fails = 0
with pkg.Synth(0) as s:
    s.set_option(pkg.OPT_SEED_WHERE, 3)
    for (fs, nsamp, nch, nb, seed, chain) in [(25e6, 100000, 16, 2, 0x5EED, True), (25e6, 100000, 16, 6, 7, False), (2.6e6, 300000, 12, 4, 9, True),
                                             (25e6, 2500000, 16, 3, 11, True), (4.092e6, 65537, 1, 3, 20, True), (3e6, 300000, 12, 5, 19, False)]:
        ch = pkg.synth_descriptors(nb, nch=nch, seed=seed)
        want_iq, want_st, _ = orc.fill_blocks(ch, 1.0 / fs, nsamp, chain=chain)
        b = s.batch(ch, 1.0 / fs, nsamp, flags=pkg.CHAIN_CARRIER if chain else 0)
        b.run(); s.sync()
        iq, st = b.read(); b.close()
        pre = s.info(pkg.INFO_PREPASS)
        nd = int((iq != want_iq).any(axis=-1).sum())
        cp = st["carr_phase"].tobytes() == want_st["carr_phase"].tobytes()
        cd = st["code_phase"].tobytes() == want_st["code_phase"].tobytes()
        rest = all(st[f].tobytes() == want_st[f].tobytes() for f in ("iword", "ibit", "icode", "dataBit", "codeCA"))
        print("fs %.3g nsamp %d nch %d nb %d chain %d: prepass %d  differing samples %d  carr_end %s code_end %s nav %s repairs %d rewalked %d hz %s" % (
            fs, nsamp, nch, nb, chain, pre, nd, cp, cd, rest, s.info(pkg.INFO_CHAIN_REPAIRS), s.info(pkg.INFO_CHAIN_FALLBACKS), s.hazards()), flush=True)
        if nd or not (cp and cd and rest) or pre != 3:
            fails += 1
            if nd:
                bad = np.argwhere((iq != want_iq).any(axis=-1))
                print("   first differing (block, sample):", bad[:5].tolist(), " last:", bad[-3:].tolist())
print("FAILS", fails)
sys.exit(1 if fails else 0)
