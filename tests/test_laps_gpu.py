"""The lap-parallel exact NCO pre-pass (pluto-gps-sim_amd/csrc/gpsbb_laps.hip.h, round 5): what replaces the inline recurrences
plutogpssim.c:2709-2712 / 2741-2746 for the model kernels.  tests/test_parity_gpu.py runs every parity case through it (mode
"laps+auto"); here: that it is what runs by default, the steps its links have to treat specially, and its repair path forced."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def f_with_step_bits(f0, delt, grid_exp, low_half, span=1 << 19):
    """an f near f0 whose step fl(f * delt) has, below 2^grid_exp, no bits (low_half False) or exactly the one below it (True)"""
    f = np.float64(f0)
    cand = f + np.spacing(f) * np.arange(span, dtype=np.float64)
    bits = np.abs(cand * np.float64(delt)).view(np.uint64)
    ex = ((bits >> np.uint64(52)) & np.uint64(0x7ff)).astype(np.int64) - 1023
    mant = (bits & np.uint64((1 << 52) - 1)) | np.uint64(1 << 52)
    dt = (grid_exp - (ex - 52)).astype(np.int64)  # bits of the mantissa below the grid
    assert (dt > 0).all() and (dt < 53).all()
    low = mant & ((np.uint64(1) << dt.astype(np.uint64)) - np.uint64(1))
    want = (np.uint64(1) << (dt - 1).astype(np.uint64)) if low_half else np.zeros_like(low)
    hit = np.nonzero(low == want)[0]
    assert hit.size
    return cand[hit[0]]


@pytest.fixture()
def fresh(pkg, synth):
    for opt in (pkg.OPT_SEED_WHERE, pkg.OPT_SYNTH_KERNEL, pkg.OPT_CHAIN_WHERE):
        synth.set_option(opt, 0)
    yield synth
    for opt in (pkg.OPT_SEED_WHERE, pkg.OPT_SYNTH_KERNEL, pkg.OPT_CHAIN_WHERE):
        synth.set_option(opt, 0)


def test_it_is_what_runs_by_default(pkg, fresh, oracle):
    """GPSBB_INFO_PREPASS: 3 = lap-parallel for the model kernels at any batch size (one block included: the drop-in call),
    1 = the row walks where the per-sample kernel renders (its tables are rows) or where the host asks for them."""
    s = fresh
    ch = pkg.synth_descriptors(3, nch=12, seed=5)
    s.fill_block(ch[0], 1 / 2.6e6, 30000)
    assert s.info(pkg.INFO_LAST_KERNEL) == 2 and s.info(pkg.INFO_PREPASS) == 3
    b = s.batch(ch, 1 / 25e6, 50000, flags=pkg.CHAIN_CARRIER)
    b.run(); s.sync(); b.close()
    assert s.info(pkg.INFO_PREPASS) == 3 and s.info(pkg.INFO_CHAIN_ON_DEVICE) == 1
    s.set_option(pkg.OPT_SYNTH_KERNEL, 1)
    s.fill_block(ch[0], 1 / 2.6e6, 30000)
    assert s.info(pkg.INFO_LAST_KERNEL) == 1 and s.info(pkg.INFO_PREPASS) in (1, 2)
    s.set_option(pkg.OPT_SYNTH_KERNEL, 0)
    s.set_option(pkg.OPT_SEED_WHERE, 1)
    s.fill_block(ch[0], 1 / 2.6e6, 30000)
    assert s.info(pkg.INFO_PREPASS) == 1
    # a carrier step below 2^-50 that is not zero is outside what the laps' turn covers: the row walks take the batch; a carrier
    # that does not move at all stands still on the laps (round 6: test_a_carrier_without_doppler_stays_on_the_laps)
    s.set_option(pkg.OPT_SEED_WHERE, 3)
    for f, want_prepass in ((1e-9, 1), (0.0, 3)):
        ch["f_carr"][:, 2] = f
        want_iq, _, _ = oracle.fill_blocks(ch, 1 / 25e6, 20000)
        b = s.batch(ch, 1 / 25e6, 20000)
        b.run(); s.sync()
        assert s.info(pkg.INFO_PREPASS) == want_prepass and (b.read()[0] == want_iq).all(), f
        b.close()


def test_steps_that_tie_on_the_coarsest_grid(pkg, fresh, oracle):
    """One block-channel in 2^12 of a real stream has a step whose sums on the coarsest grid of its chain are exact ties: a rising
    carrier's wrap sums (grid of [1, 2): 2^-52) every other wrap, every sum of a falling carrier in [0.5, 1) (2^-53), every sum of a
    code phase in [512, 1024) (2^-43).  What such a sum does to the offset between two trajectories depends on the offset
    modulo 4: part of the links (LapMap).  Here every block of five channels has such a step; bit-exact, and the links hold
    (they would not if the maps were plain translations: about every other lap then needs the repair)."""
    s = fresh
    fs, nsamp, nb, nch = 25e6, 400000, 24, 6
    delt = 1.0 / fs
    ch = pkg.synth_descriptors(nb, nch=nch, seed=4242)
    rng = np.random.default_rng(5)
    for b in range(nb):
        for i, (f0, top, half) in enumerate([(7750.0, -52, True), (4250.0, -52, False), (-5750.0, -53, True), (-2250.0, -53, False)]):
            ch["f_carr"][b, i] = f_with_step_bits(f0 * (1 + 0.01 * rng.uniform(-1, 1)), delt, top, half)
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    for b in range(nb):
        ch["f_code"][b, 4] = f_with_step_bits(1.023e6 + 2.0 * rng.uniform(-1, 1), delt, -43, True)
    ch["prn"] = np.arange(1, nch + 1)[None, :]
    want_iq, want_st, _ = oracle.fill_blocks(ch, delt, nsamp, chain=True)
    r0 = s.info(pkg.INFO_CHAIN_REPAIRS)
    b = s.batch(ch, delt, nsamp, flags=pkg.CHAIN_CARRIER)
    b.run(); s.sync()
    iq, st = b.read(); b.close()
    assert s.info(pkg.INFO_PREPASS) == 3
    assert (iq == want_iq).all()
    for f in ("carr_phase", "code_phase", "iword", "ibit", "icode", "dataBit", "codeCA"):
        assert st[f].tobytes() == want_st[f].tobytes(), f
    assert s.info(pkg.INFO_CHAIN_REPAIRS) - r0 <= 2


def test_laps_across_blocks_and_turning_carriers(pkg, fresh, oracle):
    """Laps are a chain's, not a block's: a slow carrier's lap spans many blocks (the step changes inside it), a Doppler that
    changes sign turns the phase round inside a lap (the lap then ends with the other kind of wrap, on the other grid), a channel
    is re-allocated or idle for a while (its chain ends and another starts from the descriptor's phase), the last step of a
    block wraps.  Chained batch and stream, against the oracle."""
    s = fresh
    fs, nsamp, nb, nch = 25e6, 70000, 60, 10
    delt = 1.0 / fs
    ch = pkg.synth_descriptors(nb, nch=nch, seed=99)
    rng = np.random.default_rng(12)
    t = np.arange(nb)[:, None]
    f = np.zeros((nb, nch))
    f[:, 0] = 3.0 + 0.01 * t[:, 0]                      # one lap in ~120 blocks
    f[:, 1] = -40.0                                      # a falling lap over nine blocks
    f[:, 2] = 900.0 * np.sin((t[:, 0] + 0.37) / 4.0)     # through zero again and again (never exactly zero: that is the row walks')
    f[:, 3] = np.where(t[:, 0] % 2 == 0, 2500.0, -2500.0)  # turns round at every block edge
    f[:, 4] = fs / nsamp * 3                             # three laps per block, exactly: wraps at block edges
    f[:, 5] = -fs / nsamp * 2
    f[:, 6:] = rng.uniform(-5000, 5000, (1, nch - 6)) + rng.uniform(-1, 1, (nb, nch - 6))
    ch["f_carr"] = f
    ch["f_code"] = 1.023e6 + f / 1540.0
    ch["prn"] = np.arange(1, nch + 1)[None, :]
    ch["prn"][20:, 7] = 30
    ch["prn"][33:37, 8] = 0
    ch["carr_phase"][:, 4] = 0.0
    ch["carr_phase"][:, 5] = 1.0 - 2.0 ** -53
    want_iq, want_st, hz = oracle.fill_blocks(ch, delt, nsamp, chain=True)
    s.hazards(reset=True)
    s.set_option(pkg.OPT_SEED_WHERE, 3)
    b = s.batch(ch, delt, nsamp, flags=pkg.CHAIN_CARRIER)
    b.run(); s.sync()
    iq, st = b.read(); b.close()
    assert s.info(pkg.INFO_PREPASS) == 3
    act = ch["prn"] > 0
    assert (iq == want_iq).all()
    assert st["carr_phase"][act].tobytes() == want_st["carr_phase"][act].tobytes()
    assert s.hazards(reset=True) == {"itable_512": int(hz["itable_512"]), "dwrd_oob": int(hz["dwrd_oob"])}
    bps = 6
    stq = s.stream(nch, delt, nsamp, bps, depth=3, flags=pkg.CHAIN_CARRIER)
    got, gst = [], []
    for k in range(nb // bps):
        if stq.pending == 3:
            a, e = stq.pop(); got.append(a.copy()), gst.append(e.copy())
        stq.push(ch[k * bps:(k + 1) * bps])
    while stq.pending:
        a, e = stq.pop(); got.append(a.copy()), gst.append(e.copy())
    stq.close()
    assert (np.concatenate(got).reshape(want_iq.shape) == want_iq).all()
    assert np.concatenate(gst)["carr_phase"][act].tobytes() == want_st["carr_phase"][act].tobytes()


@pytest.mark.parametrize("args,jitter,at_least", [(["--stream", "--cases", "10", "--seed", "31", "--budget", "2e7", "--also-batch"], 4000000000, 20),
                                                 (["--stream", "--cases", "8", "--seed", "32", "--budget", "2e7", "--low-rate"], 4000000000, 20),
                                                 (["--ev", "--cases", "120", "--seed", "33"], 1000000000, 20),
                                                 # (slow carriers, short blocks: few laps, so few links to break)
                                                 (["--stream", "--ties", "--cases", "8", "--seed", "34", "--budget", "2e7"], 4000000000, 3)])
def test_the_repair_path_made_common(pkg, args, jitter, at_least):
    """k_lap_repair has work about once in 10^8 laps.  The experiments build pushes every lap's reference state off the model by a
    pseudo-random number of grid steps (GPSBB_LAP_JITTER: up to 4e9 of them — 5e-7 cycles, 5e-4 chips — so that reference laps
    cross binade edges at other samples than the true ones: every tenth link then does not hold): the fuzz campaigns, forced onto
    the lap-parallel pre-pass, have to stay bit-exact — chained streams, batches, low rates, steps with few mantissa bits — and
    the repair must really have run."""
    env = dict(os.environ, GPSBB_PY_LIB="exp", GPSBB_LAP_JITTER=str(jitter), GPSBB_FUZZ_WHERE="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "bit-exact" in r.stdout
    m = re.search(r"(?:links / guesses|links) that did not hold: (\d+)", r.stdout)
    assert m and int(m.group(1)) >= at_least, r.stdout[-600:]
    assert "3:" in r.stdout.split("lap-parallel}:")[1]  # the lap-parallel pre-pass did take cases


def test_a_link_that_breaks_inside_a_group_the_repair_walks_again(pkg):
    """Round 5's soak, seed 3700 case 31 (tests/golden/lap_repair_second_break.npz: 64 blocks, a 141 Hz carrier, reference states
    pushed off by up to 2e9 grid steps): link 8 breaks, the repair walks laps 9 .. 56 again as one group, link 40 breaks INSIDE
    that group — the group's lanes beyond it have then written walks from starts that were not the truth over what pass 2 had left
    there, and pass 2's start of lap 41 happened to BE the truth: "as pass 2 had it, so everything from here on stands" left 18
    blocks' end phases one grid step off (the IQ did not notice: 2^-52 of a cycle).  Laps a failed group may have written into are
    done again whatever they start from."""
    env = dict(os.environ, GPSBB_PY_LIB="exp", GPSBB_LAP_JITTER="2000000000")
    for extra in ({}, {"GPSBB_LAP_UNIT_CARR": "9", "GPSBB_LAP_UNIT_CODE": "5"}, {"GPSBB_LAP_UNIT_CARR": "1", "GPSBB_LAP_UNIT_CODE": "1"}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_fail.py"), os.path.join(ROOT, "tests", "golden", "lap_repair_second_break.npz"),
                            "25e6", "155681", "--chain"], env=dict(env, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        first = r.stdout.split("where 1:")[0]
        assert "where 3: pre-pass 3" in first and "blocks whose IQ differs: 0 " in first and "end state" not in first, r.stdout[-1500:]
        assert int(re.search(r"repairs (\d+)", first).group(1)) >= 2


def test_one_sample_blocks(pkg, fresh, oracle):
    """Blocks of ONE sample that do not continue each other (round 5's soak, the "shapes" flavour): a carrier phase of exactly 0
    that falls made the model count a wrap before the block's first step — a second lap, whose first sample could only be the
    NEXT block's: it wrote that block's first tile state.  Never more laps than samples they could start at."""
    s = fresh
    rng = np.random.default_rng(77)
    ch = pkg.synth_descriptors(600, nch=16, seed=78)
    ch["f_carr"] = rng.choice([-1.0, 1.0], size=ch.shape) * 10.0 ** rng.uniform(-2, 5.4, size=ch.shape)
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    ch["carr_phase"] = np.floor(ch["carr_phase"] * 512.0) / 512.0
    ch["carr_phase"][rng.random(ch.shape) < 0.3] = 0.0
    ch["code_phase"][rng.random(ch.shape) < 0.1] = 0.0
    for fs, nsamp in ((10e6, 1), (25e6, 1), (25e6, 2), (2.6e6, 1)):
        for chain in (False, True):
            want_iq, want_st, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=chain)
            b = s.batch(ch, 1 / fs, nsamp, flags=pkg.CHAIN_CARRIER if chain else 0)
            b.run(); s.sync()
            iq, st = b.read(); b.close()
            assert s.info(pkg.INFO_PREPASS) == 3
            assert (iq == want_iq).all(), (fs, nsamp, chain)
            act = ch["prn"] > 0
            for f in ("carr_phase", "code_phase", "icode"):
                assert st[f][act].tobytes() == want_st[f][act].tobytes(), (fs, nsamp, chain, f)


def test_a_small_chained_stream_at_a_rate_the_laps_decline(pkg, fresh, oracle):
    """gpsbb_stream_push decides where the carrier is chained before the kernel plan exists (a small push goes to the device where
    the lap-parallel pre-pass will take it); a 1 MS/s stream is rendered by the per-sample kernel, whose pre-pass is not the laps:
    round 5's soak found the push promised the device while batch_setup chose the host threads for its size (GPSBB_E_INTERNAL);
    round 5 kept such a push on the device — by the row walks, milliseconds per one-block push.  Round 6: the promise is only
    made where the kernel plan (ev_plan) takes the blocks, so this stream stays with the host threads (0.3 ms), consistently."""
    s = fresh
    nch, bps, pushes, nsamp, fs = 3, 4, 3, 20000, 1e6
    ch = pkg.synth_descriptors(bps * pushes, nch=nch, seed=91)
    want_iq, want_st, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=True)
    st = s.stream(nch, 1 / fs, nsamp, bps, depth=2, flags=pkg.CHAIN_CARRIER)
    got = []
    for k in range(pushes):
        st.push(ch[k * bps:(k + 1) * bps])
        iq, es = st.pop(copy=True)
        got.append(np.asarray(iq).reshape(bps, -1))
    st.close()
    assert s.info(pkg.INFO_LAST_KERNEL) == 1 and s.info(pkg.INFO_PREPASS) == 2 and s.info(pkg.INFO_CHAIN_ON_DEVICE) == 0
    assert (np.concatenate(got).reshape(want_iq.shape) == want_iq).all()


@pytest.mark.parametrize("where", [3, 1])
def test_the_null_stream_owns_nothing(pkg, where):
    """Round 4's race — a fresh stream's carry cleared by a null-stream memset that the library's non-blocking streams do not
    wait for, landing after the first push had written the exact end phase: one run in a hundred under load — made
    deterministic: the experiments build parks a kernel on the null stream for 60 ms before every zeroing
    (GPSBB_X_PARK_NULL_MS), so that anything still ordered behind the null stream lands 60 ms late, every time.  With the bug
    restored (GPSBB_X_NULL_MEMSET: the zeroing as a null-stream memset nobody waits for) the second push chains from a wiped
    phase in every run; the product's zeroing (a stream of the handle, waited for) does not care what the null stream is
    doing.  For the chain of either pre-pass (the lap-parallel one reads the carry in its plan kernel, the row walks in
    k_chain_prefix / k_chain_fix_par).  DESIGN.md 4.1 has the table of who owns which buffer."""
    import json
    env = dict(os.environ, GPSBB_PY_LIB="exp", GPSBB_X_PARK_NULL_MS="60")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "order_guard.py"), str(where)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    good = json.loads(r.stdout.strip().splitlines()[-1])
    assert good["equal"] and good["park_ms"] == 60 and all(x["prepass"] == where for x in good["runs"]), good
    r = subprocess.run(cmd, env=dict(env, GPSBB_X_NULL_MEMSET="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bad = json.loads(r.stdout.strip().splitlines()[-1])
    assert bad["null_memset"] and all(x["blocks_that_differ"] > 0 for x in bad["runs"]), bad   # every run, not one in a hundred


def test_the_prepasses_leave_the_same_tables(pkg):
    """The IQ cannot tell a tile state that is off by one grid step (it changes no sample), so the tables themselves are compared:
    every tile state, every tile's data bits and every end-of-block state of 11 batches (both geometries, chained and independent,
    Dopplers over seven decades and through zero, sign changes, re-allocated and idle channels) as left by the lap-parallel
    pre-pass, by the same with its reference states pushed 1000 and 4e9 grid steps off the model (links that break, the repair
    kernel at work) and by the row walks of rounds 1-4 — bit-identical (gpsbb_test_table_digest, experiments build)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "table_check.py"), "--cases", "8", "--seed", "5"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "tables bit-identical in every mode (11 workloads x 4 modes)" in r.stdout


def test_a_carrier_without_doppler_stays_on_the_laps(pkg, fresh, oracle):
    """A carrier step of exactly zero (f_carr = 0.0 / -0.0: a bench-top scenario; c:2741 adds nothing) used to send the WHOLE batch
    to the row walks (the lap walk's turn did not cover it).  Round 6: the phase stands still, the lap-parallel pre-pass takes the
    batch — one such channel among 15 ordinary ones, one whose Doppler is zero in some blocks only (its lap passes through them),
    one standing on a phase of exactly 0.0 and one on exactly 1.0 (the table index 512 at every sample: the hazard counter counts
    them like the oracle) — as a chained batch, as independent blocks, as a chained stream across pushes, and as the drop-in call."""
    s = fresh
    fs, nsamp, nch, nb = 25e6, 150000, 16, 12
    ch = pkg.synth_descriptors(nb, nch=nch, seed=606)
    ch["f_carr"][:, 3] = 0.0
    ch["f_carr"][:, 7] = -0.0
    ch["f_carr"][4:9, 5] = 0.0            # no Doppler for a while, in the middle of a chain
    ch["f_carr"][:, 9] = 0.0
    ch["carr_phase"][:, 9] = 0.0
    ch["f_carr"][:, 11] = 0.0
    ch["carr_phase"][:, 11] = 1.0
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    for flags in (pkg.CHAIN_CARRIER, 0):
        want_iq, want_st, want_hz = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=bool(flags))
        s.hazards(reset=True)
        b = s.batch(ch, 1 / fs, nsamp, flags=flags)
        b.run()
        s.sync()
        iq, st = b.read()
        b.close()
        assert s.info(pkg.INFO_PREPASS) == 3 and s.info(pkg.INFO_LAST_KERNEL) == 2
        assert (iq == want_iq).all()
        for f in ("carr_phase", "code_phase"):
            assert st[f].tobytes() == want_st[f].tobytes(), f
        assert s.hazards(reset=True)["itable_512"] == int(want_hz["itable_512"]) > 0
    want_iq, want_st, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=True)
    st_ = s.stream(nch, 1 / fs, nsamp, 4, depth=2, flags=pkg.CHAIN_CARRIER)
    got = []
    for k in range(3):
        st_.push(ch[4 * k:4 * k + 4])
        iq, es = st_.pop(copy=True)
        got.append(np.asarray(iq).reshape(4, nsamp, 2))
        assert es["carr_phase"].tobytes() == want_st["carr_phase"][4 * k:4 * k + 4].tobytes()
    st_.close()
    assert s.info(pkg.INFO_PREPASS) == 3 and (np.concatenate(got) == want_iq).all()
    one_iq, one_st, _ = oracle.fill_blocks(ch[:1], 1 / fs, nsamp)
    iq, st1 = s.fill_block(ch[0], 1 / fs, nsamp)
    assert s.info(pkg.INFO_PREPASS) == 3 and (iq == one_iq[0]).all() and st1["carr_phase"].tobytes() == one_st["carr_phase"][0].tobytes()
    s.hazards(reset=True)   # (the channel on phase 1.0 counted: the session's handle goes on clean)
