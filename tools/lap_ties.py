"""Steps whose sums tie on the coarsest grid (one block-channel in 2^12 of a real stream): a chained batch in which three
channels have nothing but such steps, through the lap-parallel pre-pass, against the oracle; how many links broke."""
import sys, os, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
import oracle_binding as ob
pkg = g.load_package()
fs, nsamp, nb, nch = 25e6, 400000, 24, 6
delt = 1.0 / fs

def f_with_step_bits(f0, grid_exp, low_half, span=1 << 19):
    """an f_carr near f0 whose step fl(f * delt) has, below 2^grid_exp, no bits (low_half False) or exactly the one below it (True)"""
    f = np.float64(f0)
    ulp = np.spacing(f)
    cand = f + ulp * np.arange(span, dtype=np.float64)
    st = cand * np.float64(delt)
    bits = np.abs(st).view(np.uint64)
    ex = ((bits >> np.uint64(52)) & np.uint64(0x7ff)).astype(np.int64) - 1023
    mant = (bits & np.uint64((1 << 52) - 1)) | np.uint64(1 << 52)
    dt = (grid_exp - (ex - 52)).astype(np.int64)          # bits of the mantissa below the grid
    assert (dt > 0).all() and (dt < 53).all()
    low = mant & ((np.uint64(1) << dt.astype(np.uint64)) - np.uint64(1))
    want = (np.uint64(1) << (dt - 1).astype(np.uint64)) if low_half else np.zeros_like(low)
    hit = np.nonzero(low == want)[0]
    assert hit.size, "no such step within %d ulps of %r" % (span, f0)
    return cand[hit[0]]

ch = pkg.synth_descriptors(nb, nch=nch, seed=4242)
rng = np.random.default_rng(5)
kinds = ["rising, low bits = half of 2^-52", "rising, multiple of 2^-52", "falling, low bits = half of 2^-53", "falling, multiple of 2^-53", "code step: low bits = half of 2^-43", "ordinary"]
for b in range(nb):
    for i, (f0, top, half) in enumerate([(7750.0, -52, True), (4250.0, -52, False), (-5750.0, -53, True), (-2250.0, -53, False)]):
        ch["f_carr"][b, i] = f_with_step_bits(f0 * (1 + 0.01 * rng.uniform(-1, 1)), top, half)
ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
for b in range(nb):  # ... and a code step that ties on the grid of [512, 1024), 2^-43
    ch["f_code"][b, 4] = f_with_step_bits(1.023e6 + 2.0 * rng.uniform(-1, 1), -43, True)
ch["prn"] = np.arange(1, nch + 1)[None, :]
want_iq, want_st, _ = ob.Oracle().fill_blocks(ch, delt, nsamp, chain=True)
with pkg.Synth(0) as s:
    for where in (3, 1):
        s.set_option(pkg.OPT_SEED_WHERE, where)
        r0, w0 = s.info(pkg.INFO_CHAIN_REPAIRS), s.info(pkg.INFO_CHAIN_FALLBACKS)
        b = s.batch(ch, delt, nsamp, flags=pkg.CHAIN_CARRIER)
        b.run(); s.sync()
        iq, st = b.read(); b.close()
        print("where %d prepass %d: differing samples %d, end phases equal %s, links broken %d, laps walked again %d" % (
            where, s.info(pkg.INFO_PREPASS), int((iq != want_iq).any(axis=-1).sum()),
            st["carr_phase"].tobytes() == want_st["carr_phase"].tobytes(), s.info(pkg.INFO_CHAIN_REPAIRS) - r0, s.info(pkg.INFO_CHAIN_FALLBACKS) - w0))
print("channels:", kinds)
