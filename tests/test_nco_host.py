"""Exact NCO jump-ahead (csrc/gpsbb_nco.h, the code the device pre-pass runs) against brute-force
stepping of the reference's recurrences (plutogpssim.c:2709-2712, 2741-2746), on the host.

Python floats are IEEE doubles and CPython never contracts a*b+c, so the loops below are the reference's
arithmetic."""
import ctypes as C
import math
import random
import struct

import numpy as np
import pytest


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def brute_carr(x, s, n, record=None, wraps=None):
    w = 0
    for _ in range(n):
        if record is not None:
            record.append(x)
        if wraps is not None:
            wraps.append(w)
        x = x + s
        if x >= 1.0:
            x -= 1.0
            w += 1
        elif x < 0.0:
            x += 1.0
            w += 1
    return x


def brute_code(x, s, n, record=None):
    w = 0
    for _ in range(n):
        if record is not None:
            record.append((x, w))
        x = x + s
        if x >= 1023.0:
            x -= 1023.0
            w += 1
    return x, w


FS = [1.0e6, 2.6e6, 3.0e6, 4.092e6, 25.0e6]


def carr_cases(rng, n):
    out = []
    for _ in range(n):
        fs = rng.choice(FS)
        f = rng.choice([-1, 1]) * 10.0 ** rng.uniform(-2, 4.3)
        out.append((rng.random(), f * (1.0 / fs)))
    return out


def test_carr_jump_random(pkg):
    L = pkg.exp_lib()
    rng = random.Random(1234)
    for x0, s in carr_cases(rng, 60):
        n = rng.choice([1, 7, 1000, 30000, 300000])
        want = brute_carr(x0, s, n)
        got = L.gpsbb_test_carr_jump(x0, s, n)
        assert bits(got) == bits(want), (x0, s, n)


def test_carr_jump_edge_cases(pkg):
    L = pkg.exp_lib()
    cases = [
        (0.0, 1e-4), (0.0, -1e-4), (1.0, 1e-4), (1.0, -1e-4), (1.0, 0.0), (0.5, 0.0), (0.25, -0.0),
        (0.5, 2.0 ** -10), (0.5, -(2.0 ** -10)),              # step with a 1-bit mantissa: ties everywhere
        (0.3, 2.0 ** -10 + 2.0 ** -62), (0.3, -(2.0 ** -10 + 2.0 ** -62)),
        (0.7, 3 * 2.0 ** -12), (0.1, 0.125), (0.9, -0.125),   # largest steps of the contract
        (0.5, 2.0 ** -60), (0.5, -(2.0 ** -60)), (0.5, 1e-300), (0.75, -1e-300), (0.5, 5e-324),
        (2.0 ** -30, -(2.0 ** -31)), (2.0 ** -53, 2.0 ** -53), (1.0 - 2.0 ** -53, 2.0 ** -54),
        (0.5 + 2.0 ** -53, 2.0 ** -54 + 2.0 ** -80),
    ]
    for x0, s in cases:
        for n in (1, 2, 3, 50, 5000, 70000):
            want = brute_carr(x0, s, n)
            got = L.gpsbb_test_carr_jump(x0, s, n)
            assert bits(got) == bits(want), (x0, s, n)


def test_code_jump_random(pkg):
    L = pkg.exp_lib()
    rng = random.Random(99)
    for _ in range(40):
        fs = rng.choice(FS)
        fc = 1.023e6 + rng.uniform(-6000, 6000) / 1540.0
        s = fc * (1.0 / fs)
        x0 = rng.random() * 1023.0
        n = rng.choice([1, 13, 5000, 300000])
        want, w = brute_code(x0, s, n)
        wr = C.c_longlong()
        got = L.gpsbb_test_code_jump(x0, s, n, C.byref(wr))
        assert bits(got) == bits(want) and wr.value == w, (x0, s, n)


def test_code_jump_edges(pkg):
    L = pkg.exp_lib()
    for x0, s in [(0.0, 1.023), (1022.999999, 1.5), (1022.5, 0.5), (512.0, 0.25), (0.0, 0.04092), (1023.0 - 2.0 ** -43, 0.3)]:
        for n in (1, 2, 100, 4000, 100000):
            want, w = brute_code(x0, s, n)
            wr = C.c_longlong()
            got = L.gpsbb_test_code_jump(x0, s, n, C.byref(wr))
            assert bits(got) == bits(want) and wr.value == w, (x0, s, n)


def rows_for(pkg, kind, x0, s, nav0, nsamp):
    L = pkg.exp_lib()
    cap = int(L.gpsbb_test_row_bound(kind, abs(s), nsamp))
    rows = np.zeros(cap, pkg.ROW_DTYPE)
    xe, ne = C.c_double(), C.c_uint()
    cnt = L.gpsbb_test_build_rows(kind, x0, s, nav0, nsamp, rows.ctypes.data, cap, C.byref(xe), C.byref(ne))
    assert cnt <= cap, "row bound violated: %d > %d (kind=%d s=%r)" % (cnt, cap, kind, s)
    return rows[:cnt], xe.value, ne.value


def check_rows_cover(rows, traj_bits, nsamp):
    """Every sample's state must be rows[r].xb + (n - n0)*inc for the row that holds n."""
    n0 = rows["n0"].astype(np.int64)
    assert n0[0] == 0 and (np.diff(n0) > 0).all()
    n = np.arange(nsamp, dtype=np.int64)
    r = np.searchsorted(n0, n, side="right") - 1
    with np.errstate(over="ignore"):
        got = rows["xb"][r] + ((n - n0[r]) * rows["inc"][r]).astype(np.uint64)
    bad = np.nonzero(got != traj_bits)[0]
    assert bad.size == 0, "first mismatch at sample %d" % bad[0]


def test_rows_carrier(pkg):
    rng = random.Random(5)
    cases = carr_cases(rng, 25) + [(0.5, 2.0 ** -10), (0.0, -1e-4), (1.0, 3e-4), (0.5, 0.0), (0.3, -0.125), (0.3, 0.125)]
    for x0, s in cases:
        nsamp = rng.choice([1, 100, 4096, 50000])
        rec = []
        xe_want = brute_carr(x0, s, nsamp, rec)
        rows, xe, _ = rows_for(pkg, 1, x0, s, 0, nsamp)
        assert bits(xe) == bits(xe_want)
        check_rows_cover(rows, np.array([bits(v) for v in rec], np.uint64), nsamp)


def test_rows_code_and_nav(pkg):
    rng = random.Random(6)
    for _ in range(20):
        fs = rng.choice(FS)
        s = (1.023e6 + rng.uniform(-6000, 6000) / 1540.0) * (1.0 / fs)
        x0 = rng.random() * 1023.0
        nsamp = rng.choice([1, 3000, 60000])
        icode, ibit, iword = rng.randrange(20), rng.randrange(30), rng.randrange(9, 59)
        rec = []
        xe_want, w = brute_code(x0, s, nsamp, rec)
        rows, xe, nav_end = rows_for(pkg, 0, x0, s, icode | (ibit << 5) | (iword << 10), nsamp)
        assert bits(xe) == bits(xe_want)
        check_rows_cover(rows, np.array([bits(v[0]) for v in rec], np.uint64), nsamp)
        # nav counters per row == counters after that many wraps (plutogpssim.c:2714-2733)
        def adv(k):
            t = icode + k
            c, t = t % 20, t // 20
            t += ibit
            return c | ((t % 30) << 5) | ((iword + t // 30) << 10)
        wraps_at = np.array([v[1] for v in rec])
        for r in rows:
            assert r["nav"] == adv(int(wraps_at[r["n0"]]))
        assert nav_end == adv(w)


def rows_f64_for(pkg, kind, x0, s, nav0, nsamp):
    L = pkg.exp_lib()
    cap = int(L.gpsbb_test_row_bound(kind, abs(s), nsamp))
    rows = np.zeros(cap, pkg.ROW_DTYPE)
    xe, ne = C.c_double(), C.c_uint()
    L.gpsbb_test_build_rows_f64.restype = C.c_int
    L.gpsbb_test_build_rows_f64.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint, C.c_int, C.c_void_p, C.c_int,
                                            C.POINTER(C.c_double), C.POINTER(C.c_uint)]
    cnt = L.gpsbb_test_build_rows_f64(kind, x0, s, nav0, nsamp, rows.ctypes.data, cap, C.byref(xe), C.byref(ne))
    assert cnt <= cap, "row bound violated: %d > %d (kind=%d s=%r)" % (cnt, cap, kind, s)
    return rows[:cnt], xe.value, ne.value


def check_f64_rows(pkg, kind, x0, s, nav0, nsamp, traj, nav_at=None, wraps_at=None):
    """The device pre-pass's builder (double arithmetic, rows {n0, nav, x, S}): same end state as the integer
    builder, row count within the bound, and, sample by sample against the brute-force trajectory,
    x + (n - n0)*S in exact rational arithmetic is the state at n (that is what one FMA returns)."""
    from fractions import Fraction
    _, xe_i, nav_i = rows_for(pkg, kind, x0, s, nav0, nsamp)
    rf, xe_f, nav_f = rows_f64_for(pkg, kind, x0, s, nav0, nsamp)
    assert bits(xe_f) == bits(xe_i) and nav_f == nav_i
    n0 = rf["n0"].astype(np.int64)
    assert n0[0] == 0 and (np.diff(n0) > 0).all()
    xs = rf["xb"].view(np.float64)
    Ss = rf["inc"].view(np.float64)
    assert (rf["xb"] == np.array([bits(traj[int(k)]) for k in n0], np.uint64)).all()
    if nav_at is not None:
        for r in rf:
            assert (r["nav"] & 0x3fffffff) == nav_at(int(r["n0"]))
    if wraps_at is not None:
        # bit 30: the step that led to the row's first sample wrapped; every wrap starts a row
        for r in rf:
            k = int(r["n0"])
            want = (k > 0 and wraps_at[k] != wraps_at[k - 1]) or (kind == 1 and traj[k] >= 1.0)
            assert bool(r["nav"] & 0x40000000) == want, (kind, x0, s, k)
        starts = set(int(k) for k in n0)
        for k in range(1, nsamp):
            if wraps_at[k] != wraps_at[k - 1]:
                assert k in starts, (kind, x0, s, k)
    rng = random.Random(len(rf) * 7919 + nsamp)
    picks = set(int(v) - 1 for v in n0[1:]) | {nsamp - 1}
    picks |= set(rng.randrange(nsamp) for _ in range(300))
    for n in sorted(picks):
        r = int(np.searchsorted(n0, n, side="right")) - 1
        got = Fraction(float(xs[r])) + (n - int(n0[r])) * Fraction(float(Ss[r]))
        assert got == Fraction(traj[n]), (kind, x0, s, n, r)


def test_rows_f64_builder_carrier(pkg):
    rng = random.Random(15)
    cases = carr_cases(rng, 40) + [
        (0.5, 2.0 ** -10), (0.0, -1e-4), (1.0, 3e-4), (0.5, 0.0), (0.3, -0.125), (0.3, 0.125), (0.5, 2.0 ** -60),
        (0.5, 1e-300), (0.75, -1e-300), (2.0 ** -30, -(2.0 ** -31)), (0.5 + 2.0 ** -53, 2.0 ** -54 + 2.0 ** -80),
        (0.3, 2.0 ** -10 + 2.0 ** -62), (0.3, -(2.0 ** -10 + 2.0 ** -62)), (0.7, 3 * 2.0 ** -12),
        (0.25, 2.0 ** -13), (0.25, -(2.0 ** -13)), (1e-290, 1e-295), (0.6, 2.0 ** -3 - 2.0 ** -56)]
    for x0, s in cases:
        nsamp = rng.choice([1, 100, 4096, 50000])
        rec, wr = [], []
        brute_carr(x0, s, nsamp, rec, wr)
        check_f64_rows(pkg, 1, x0, s, 0, nsamp, rec, None, wr)


def test_rows_f64_builder_code(pkg):
    rng = random.Random(16)
    cases = []
    for _ in range(25):
        fs = rng.choice(FS + [2.0 ** 25])
        cases.append((rng.random() * 1023.0, (1.023e6 + rng.uniform(-6000, 6000) / 1540.0) * (1.0 / fs)))
    cases += [(0.0, 1.023), (1022.999999, 1.5), (1022.5, 0.5), (512.0, 0.25), (0.0, 0.04092), (1023.0 - 2.0 ** -43, 0.3),
              (5.0, 2.0 ** -5), (1022.96875, 2.0 ** -5), (0.0, 2.0 ** -4), (700.0, 1.0), (3.0, 1e-3)]
    for x0, s in cases:
        nsamp = rng.choice([1, 3000, 60000])
        rec = []
        brute_code(x0, s, nsamp, rec)

        def nav_at(n, rec=rec):
            t = 3 + rec[n][1]
            c, t = t % 20, t // 20 + 7
            return c | ((t % 30) << 5) | ((20 + t // 30) << 10)
        check_f64_rows(pkg, 0, x0, s, 3 | (7 << 5) | (20 << 10), nsamp, [v[0] for v in rec], nav_at, [v[1] for v in rec])


def test_chain_carrier_host_matches_oracle(pkg, oracle):
    ch = pkg.synth_descriptors(6, nch=5, seed=77)
    ch["prn"][3:, 2] = 9          # channel re-allocated to another PRN at block 3: phase restarts
    ch["prn"][:, 4] = 0           # inactive channel
    delt, nsamp = 1 / 2.6e6, 20000
    seeds = pkg.chain_carrier_host(ch, delt, nsamp)
    _, st, _ = oracle.fill_blocks(ch, delt, nsamp, chain=True, want_iq=False)
    for b in range(1, 6):
        for i in range(4):
            if ch["prn"][b, i] == ch["prn"][b - 1, i]:
                assert bits(seeds[b, i]) == bits(st["carr_phase"][b - 1, i])
            else:
                assert bits(seeds[b, i]) == bits(ch["carr_phase"][b, i])
    assert (seeds[0, :4] == ch["carr_phase"][0, :4]).all()


def test_drift_model_predicts_the_carrier_to_1e13(pkg):
    """The host's drift model of the recurrence (CarrDrift: where pass B of the device-side chain starts a segment when
    no walk predicts it) against the exact jump-ahead: within 2e-13 cycles over a segment's length for Dopplers of either
    sign at both sample rates, and an order of magnitude closer than x0 + n*s.  Only a prediction — the chain's
    exactness never rests on it — but a prediction this good is what keeps the fix-up from walking segments."""
    L = pkg.exp_lib()
    rng = np.random.default_rng(7)
    for fs, n in ((25e6, 625000), (2.6e6, 75000), (3e6, 300000)):
        err, naive = [], []
        for _ in range(200):
            s = float(rng.uniform(-5000.0, 5000.0)) / fs
            x0 = float(rng.random())
            want = L.gpsbb_test_carr_jump(x0, s, n)
            d = L.gpsbb_test_carr_predict(x0, s, n) - want
            err.append(d - round(d))
            v = x0 + n * s
            d = (v - np.floor(v)) - want
            naive.append(d - round(d))
        assert np.abs(err).max() < 2e-13, (fs, np.abs(err).max())
        assert np.std(err) * 10 < np.std(naive), (fs, np.std(err), np.std(naive))
    # steps the model does not cover fall back to plain arithmetic: still a number in [0, 1)
    for s in (0.0, 1e-30, 0.3, -0.3):
        v = L.gpsbb_test_carr_predict(0.25, s, 1000)
        assert 0.0 <= v < 1.0


def test_fixed_point_carrier_tile_index_is_the_accumulator(pkg):
    """What the model kernels start a tile from with GPSBB_FIXED_CARRIER: (phase mod 2^25) / 2^16 of the reference's 32-bit
    accumulator (c:2699, 2748) after tile * 1024 steps — the index with its fraction, exactly; the mirrored value used for a
    falling phase, 512 - 2^-16 - y, is the index of the bitwise complement."""
    L = pkg.exp_lib()
    rng = np.random.default_rng(11)
    for _ in range(2000):
        ph0 = int(rng.integers(0, 2 ** 32))
        step = int(rng.integers(-2 ** 20, 2 ** 20))
        t = int(rng.integers(0, 3000))
        ph = ph0
        ph = (ph0 + t * 1024 * step) % 2 ** 32
        y = L.gpsbb_test_fixed_tile_index(ph0, step, t)
        assert y == (ph % 2 ** 25) / 65536.0
        assert int(y) == (ph >> 16) & 0x1ff
        ym = (512.0 - 2.0 ** -16) - y
        assert int(ym) == 511 - ((ph >> 16) & 0x1ff) and ym == ((2 ** 25 - 1 - ph % 2 ** 25)) / 65536.0

