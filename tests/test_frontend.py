"""The from-scratch host front end (pluto-gps-sim_amd/host/gpsfe.c) against descriptor dumps of the
reference's own code: every field of every channel of every kept block must be bit-identical.  carr_phase
(which the reference carries in the sample loop) is reconstructed with gpsbb_chain_carrier_host.  CPU only."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from conftest import GOLDEN

SITE = (30.286502, 120.032669, 100.0)
FIELDS = ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "dwrd", "carr_phase")


@pytest.fixture(scope="module")
def fe_pkg(pkg):
    pkg.build_frontend()
    return pkg


def chained(pkg, ch, fs, nsamp):
    out = ch.copy()
    out["carr_phase"] = pkg.chain_carrier_host(ch, 1.0 / fs, nsamp)
    out["carr_phase"][ch["prn"] <= 0] = 0.0
    return out


def assert_desc_equal(got, want, what):
    for f in FIELDS:
        a, b = np.ascontiguousarray(got[f]), np.ascontiguousarray(want[f])
        assert a.tobytes() == b.tobytes(), "%s: field %s differs" % (what, f)


# golden file, nav file, motion file, MAX_CHAN, further FrontEnd options
SCENARIOS = [
    ("static_F", "synth3540.14n", None, 12, {}),
    ("motion_F", "synth3540.14n", "circle_motion.csv", 12, {}),
    ("dense_S", "dense3540.14n", None, 16, {}),
    ("rinex3_F", "synth3540_v3.rnx", None, 12, {"rinex3": True}),                    # readRinex3, c:1241-1610
    ("toverwrite_F", "synth3540.14n", None, 12, {"start": (2014, 12, 21, 10, 0, 0.0), "time_overwrite": True}),  # c:2523-2553
    ("motion_ref_F", "synth3540.14n", "circle.csv", 12, {}),                         # the reference's own circle.csv
    ("swap_S", "dense3540.14n", None, 16, {"start": (2014, 12, 20, 1, 20, 0.0)}),    # a slot changes hands at block 1500
]


@pytest.mark.parametrize("name,nav,motion,max_chan,kw", SCENARIOS)
def test_descriptors_match_the_reference_dumps(fe_pkg, name, nav, motion, max_chan, kw):
    pkg = fe_pkg
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    blocks = [int(b) for b in z["blocks"]]
    want = z["desc"].view(pkg.CHAN_DTYPE).reshape(len(blocks), -1)
    fe = pkg.FrontEnd(os.path.join(GOLDEN, nav), llh=SITE, motion=os.path.join(GOLDEN, motion) if motion else None,
                      max_chan=max_chan, **kw)
    ch = chained(pkg, fe.generate(max(blocks) + 1), fs, nsamp)
    fe.close()
    for k, b in enumerate(blocks):
        assert_desc_equal(ch[b], want[k], "%s block %d" % (name, b))


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("extra,kw", [
    ((), {}),
    (("-i",), {"iono": False}),
    (("-t", "2014/12/20,01:30:17"), {"start": (2014, 12, 20, 1, 30, 17.0)}),
    (("-t", "2014/12/21,10:00:00", "-T"), {"start": (2014, 12, 21, 10, 0, 0.0), "time_overwrite": True}),
])
def test_every_block_against_a_live_reference_run(fe_pkg, extra, kw):
    """All blocks of a 65 s run (two nav-frame refreshes) against the reference slices run right now,
    for the option combinations that change the front end's arithmetic."""
    pkg = fe_pkg
    nav = os.path.join(GOLDEN, "synth3540.14n")
    nblocks, nsamp, fs = 650, 2000, 2600000
    _, want, _ = ob.run_ref_sim(nav, nblocks, nsamp, fs, llh=tuple(str(v) for v in SITE), max_chan=12, opt="",
                                extra=extra)
    fe = pkg.FrontEnd(nav, llh=SITE, max_chan=12, **kw)
    ch = chained(pkg, fe.generate(nblocks), fs, nsamp)
    fe.close()
    assert_desc_equal(ch, want, "live run %r" % (extra,))


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_ecef_position_and_ephemeris_rollover(fe_pkg):
    """-c ECEF input, and a start 20 s before the 02:00 set becomes current so the roll-over at c:2776-2790
    (ieph++ and fresh subframes) happens inside the run."""
    pkg = fe_pkg
    nav = os.path.join(GOLDEN, "synth3540.14n")
    xyz = (-2758918.636, 4772301.120, 3197889.437)
    nblocks, nsamp, fs = 700, 1000, 2600000
    extra = ("-c", "%r,%r,%r" % xyz, "-t", "2014/12/20,00:59:40")
    _, want, _ = ob.run_ref_sim(nav, nblocks, nsamp, fs, max_chan=12, extra=extra)
    fe = pkg.FrontEnd(nav, ecef=xyz, start=(2014, 12, 20, 0, 59, 40.0), max_chan=12)
    ch = chained(pkg, fe.generate(nblocks), fs, nsamp)
    fe.close()
    assert_desc_equal(ch, want, "rollover run")
    assert (want["dwrd"][0] != want["dwrd"][-1]).any()


def test_front_end_errors(fe_pkg):
    pkg = fe_pkg
    with pytest.raises(RuntimeError):
        pkg.FrontEnd("/nonexistent.14n", llh=SITE)
    with pytest.raises(RuntimeError):  # start outside the window (c:2555-2564)
        pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=SITE, start=(2015, 1, 1, 0, 0, 0.0))
    with pytest.raises(RuntimeError):
        pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=SITE, motion="/nonexistent.csv")


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_rinex3_reader_against_the_reference(fe_pkg):
    """readRinex3 (plutogpssim.c:1241-1610): shifted columns, iono/UTC under other labels, non-GPS records
    skipped.  Whole run against the reference slices reading the same RINEX 3 file."""
    pkg = fe_pkg
    nav = os.path.join(GOLDEN, "synth3540_v3.rnx")
    nblocks, nsamp, fs = 320, 1000, 2600000
    _, want, _ = ob.run_ref_sim(nav, nblocks, nsamp, fs, llh=tuple(str(v) for v in SITE), max_chan=12, extra=("-3",))
    fe = pkg.FrontEnd(nav, llh=SITE, max_chan=12, rinex3=True)
    ch = chained(pkg, fe.generate(nblocks), fs, nsamp)
    fe.close()
    assert_desc_equal(ch, want, "RINEX 3 run")
    assert (want["prn"][0] > 0).sum() == 12


def test_rinex3_and_rinex2_files_give_the_same_orbits(fe_pkg):
    """The two fixture files carry the same ephemerides; everything except the nav words that hold the
    (differently rounded) UTC terms must agree."""
    pkg = fe_pkg
    a = pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=SITE, max_chan=12).generate(5)
    b = pkg.FrontEnd(os.path.join(GOLDEN, "synth3540_v3.rnx"), llh=SITE, max_chan=12, rinex3=True).generate(5)
    for f in ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "carr_phase"):
        assert a[f].tobytes() == b[f].tobytes(), f
    with pytest.raises(RuntimeError):   # a v2 file through the v3 reader is rejected (c:1279-1282)
        pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=SITE, rinex3=True)


def test_fixed_carrier_descriptors(fe_pkg):
    """Front end in the fixed-point carrier variant: the accumulator's initial value (c:1966-1967) and its
    chain (start + nsamp*step mod 2^32) against the dumps of the reference built without FLOAT_CARR_PHASE."""
    pkg = fe_pkg
    z = np.load(os.path.join(GOLDEN, "static_F_fixed.npz"))
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    blocks = [int(b) for b in z["blocks"]]
    want = z["desc"].view(pkg.CHAN_DTYPE).reshape(len(blocks), -1)
    fe = pkg.FrontEnd(os.path.join(GOLDEN, "synth3540.14n"), llh=SITE, max_chan=12, fixed_carrier=True)
    ch = fe.generate(max(blocks) + 1)
    fe.close()
    # chain the 32-bit accumulator on the host the way the loop does (c:2675, 2748)
    step = np.round(512.0 * 65536.0 * ch["f_carr"] * (1.0 / fs)).astype(np.int64)
    ph = ch["carr_phase"].astype(np.int64)
    for b in range(1, ch.shape[0]):
        same = (ch["prn"][b] == ch["prn"][b - 1]) & (ch["prn"][b] > 0)
        ph[b][same] = (ph[b - 1][same] + nsamp * step[b - 1][same]) % (1 << 32)
    ch["carr_phase"] = ph.astype(np.float64)
    for k, b in enumerate(blocks):
        assert_desc_equal(ch[b], want[k], "fixed block %d" % b)


def test_feed_back_keeps_a_newly_allocated_channels_own_phase(fe_pkg):
    """The drop-in loop (gpsfe_next_block / render / gpsfe_feed_back) across the 30 s maintenance of block 1499, which
    frees channel 10 (PRN 11 has set) and gives it to PRN 18 in the same pass: the new satellite starts from
    allocateChannel's phase (c:1956-1964), not from the end phase of the block PRN 11 was rendered in.  The render is
    replaced by the exact carrier jump-ahead (the end phase is all feed_back looks at); descriptors against the
    reference's dump around the hand-over."""
    pkg = fe_pkg
    z = np.load(os.path.join(GOLDEN, "swap_S.npz"))
    fs, nsamp = float(z["fs"]), int(z["nsamp"])
    blocks = [int(b) for b in z["blocks"]]
    want = z["desc"].view(pkg.CHAN_DTYPE).reshape(len(blocks), -1)
    assert want["prn"][1][10] == 11 and want["prn"][2][10] == 18
    fe = pkg.FrontEnd(os.path.join(GOLDEN, "dense3540.14n"), llh=SITE, max_chan=16, start=(2014, 12, 20, 1, 20, 0.0))
    lib = pkg.exp_lib()
    for b in range(max(blocks) + 1):
        ch = fe.next_block()
        if b in blocks:
            assert_desc_equal(ch, want[blocks.index(b)], "feed-back loop, block %d" % b)
        st = np.zeros(16, pkg.STATE_DTYPE)
        for i in range(16):
            if ch["prn"][i] > 0:
                s = float(np.float64(ch["f_carr"][i]) * np.float64(1.0 / fs))
                st["carr_phase"][i] = lib.gpsbb_test_carr_jump(float(ch["carr_phase"][i]), s, nsamp)
                st["dataBit"][i] = 1
        fe.feed_back(st)
    fe.close()


@pytest.mark.parametrize("motion,max_chan,nav", [(None, 12, "synth3540.14n"), ("circle.csv", 12, "synth3540.14n"), (None, 16, "dense3540.14n")])
def test_generate_on_threads_equals_the_sequential_loop(pkg, motion, max_chan, nav):
    """gpsfe_generate spreads the blocks between two 30 s maintenances (c:2764-2798) over threads (ranges first, then every
    block's descriptor from the ranges at its ends): the same bytes as one gpsfe_next_block per block, across
    maintenances, ephemeris roll-overs, the wrap of the motion file, and when the two calls are mixed."""
    pkg.build_frontend()
    kw = dict(llh=SITE, max_chan=max_chan, motion=os.path.join(GOLDEN, motion) if motion else None)
    n = 3700  # 370 s: twelve maintenances; circle.csv (3000 points) wraps
    seq = pkg.FrontEnd(os.path.join(GOLDEN, nav), **kw)
    want = np.stack([seq.next_block() for _ in range(n)])
    seq.close()
    for threads in (1, 2, 7, 0):
        fe = pkg.FrontEnd(os.path.join(GOLDEN, nav), **kw)
        fe.set_threads(threads)
        got = fe.generate(n)
        assert got.tobytes() == want.tobytes(), threads
        fe.close()
    fe = pkg.FrontEnd(os.path.join(GOLDEN, nav), **kw)
    fe.set_threads(5)
    parts = [fe.generate(100), np.stack([fe.next_block() for _ in range(3)]), fe.generate(299), fe.generate(1), fe.generate(n - 403)]
    assert np.concatenate(parts).tobytes() == want.tobytes()
    fe.close()
