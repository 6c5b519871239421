"""The CPU oracle against (a) the committed golden vectors, which were produced by the reference's own
statements (tests/golden/make_golden.py), and (b) — when oracle/_ref is present — the verbatim reference
loop itself on fresh random descriptors.  CPU only."""
import hashlib
import os

import numpy as np
import pytest

import oracle_binding as ob
from conftest import GOLDEN

FIXTURES = ["static_F", "motion_F", "dense_S", "loop_M2"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z, float(z["fs"]), int(z["nsamp"])


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_golden_blocks(oracle, name):
    z, fs, nsamp = load(name)
    desc = z["desc"].view(ob.CHAN_DTYPE).reshape(z["desc"].shape[0], -1)
    iq, st, hz = oracle.fill_blocks(desc, 1.0 / fs, nsamp)
    assert hz["itable_512"] == 0 and hz["dwrd_oob"] == 0
    want_st = z["end_state"].view(ob.STATE_DTYPE).reshape(st.shape)
    for k in range(desc.shape[0]):
        assert (iq[k, :z["iq_prefix"].shape[1]] == z["iq_prefix"][k]).all(), (name, k)
        assert sha(iq[k]) == str(z["iq_sha256"][k]), (name, k)
        for f in ("carr_phase", "code_phase", "iword", "ibit", "icode", "dataBit", "codeCA"):
            assert (st[f][k] == want_st[f][k]).all(), (name, k, f)


def test_golden_carrier_chain_is_continuous():
    """Consecutive blocks of the real run: block b+1 starts where block b ended (plutogpssim.c:2741-2746)."""
    z, _, _ = load("static_F")
    desc = z["desc"].view(ob.CHAN_DTYPE).reshape(z["desc"].shape[0], -1)
    st = z["end_state"].view(ob.STATE_DTYPE).reshape(desc.shape)
    blocks = list(z["blocks"])
    for a, b in zip(range(len(blocks) - 1), range(1, len(blocks))):
        if blocks[b] == blocks[a] + 1:
            same = desc["prn"][a] == desc["prn"][b]
            assert (desc["carr_phase"][b][same] == st["carr_phase"][a][same]).all()


def test_tables_and_codes_are_stable(oracle):
    s, c = oracle.tables()
    assert s[0] == 1 and c[0] == 512 and c[384] == 0 and s[128] == 512 and s.sum() == 512 and c.sum() == 511
    # first ten chips of PRN 1 are 1100100000 (octal 1440, ICD-GPS-200 table 3-I)
    assert "".join(map(str, oracle.codegen(1)[:10])) == "1100100000"
    assert "".join(map(str, oracle.codegen(32)[:10])) == "1111001010"  # octal 1712


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
class TestAgainstReferenceSlices:
    def test_tables_and_codegen(self, oracle):
        r = ob.RefLoop()
        for a, b in zip(oracle.tables(), r.tables()):
            assert (a == b).all()
        for prn in range(1, 33):
            assert (oracle.codegen(prn) == r.codegen(prn)).all()

    @pytest.mark.parametrize("fs,nsamp,nch,seed", [(25e6, 60000, 16, 1), (2.6e6, 40000, 12, 2), (1e6, 5000, 3, 3),
                                                    (3e6, 1, 16, 4), (4.092e6, 30001, 7, 5)])
    def test_loop_random(self, oracle, pkg, fs, nsamp, nch, seed):
        r = ob.RefLoop()
        ch = pkg.synth_descriptors(2, nch=nch, seed=seed)
        for b in range(2):
            want_iq, want_st = r.fill(ch[b], 1.0 / fs, nsamp)
            iq, st, hz = oracle.fill_blocks(ch[b], 1.0 / fs, nsamp)
            assert (iq[0] == want_iq).all()
            assert st[0].tobytes() == want_st.tobytes()

    def test_O0_equals_O2(self, pkg):
        ch = pkg.synth_descriptors(1, nch=16, seed=9)[0]
        a = ob.RefLoop().fill(ch, 1 / 25e6, 50000)
        b = ob.RefLoop("_O2").fill(ch, 1 / 25e6, 50000)
        assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()

    def test_scenario_runner_O0_equals_O2(self):
        nav = os.path.join(GOLDEN, "dense3540.14n")
        a = ob.run_ref_sim(nav, 3, 30000, 2600000, llh=("30.286502", "120.032669", "100"), max_chan=16)
        b = ob.run_ref_sim(nav, 3, 30000, 2600000, llh=("30.286502", "120.032669", "100"), max_chan=16, opt="_O2")
        assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()


def test_oracle_rejects_out_of_contract(oracle, pkg):
    ch = pkg.synth_descriptors(1, nch=2, seed=3)
    bad = ch.copy()
    bad["code_phase"][0, 0] = 1023.0
    with pytest.raises(ValueError):
        oracle.fill_blocks(bad, 1 / 2.6e6, 10)
    bad = ch.copy()
    bad["prn"][0, 1] = 33
    with pytest.raises(ValueError):
        oracle.fill_blocks(bad, 1 / 2.6e6, 10)


def test_oracle_hazards_are_defined_and_counted(oracle, pkg):
    ch = pkg.synth_descriptors(1, nch=1, seed=3)
    ch["carr_phase"][0, 0] = 1.0           # only reachable through the latent rounding case
    ch["f_carr"][0, 0] = -100.0
    _, _, hz = oracle.fill_blocks(ch, 1 / 2.6e6, 10)
    assert hz["itable_512"] == 1
    ch = pkg.synth_descriptors(1, nch=1, seed=4)
    ch["iword"][0, 0], ch["ibit"][0, 0], ch["icode"][0, 0] = 59, 29, 19
    _, st, hz = oracle.fill_blocks(ch, 1 / 1e6, 100000)   # 0.1 s: 100 code periods -> 5 bit fetches in word 60
    assert hz["dwrd_oob"] == 5 and st["iword"][0, 0] == 60


# ---- the reference's fixed-point carrier variant (`#ifndef FLOAT_CARR_PHASE`) ----------------------------

def test_fixed_carrier_oracle_reproduces_golden_blocks(oracle):
    z, fs, nsamp = load("static_F_fixed")
    desc = z["desc"].view(ob.CHAN_DTYPE).reshape(z["desc"].shape[0], -1)
    iq, st, _ = oracle.fill_blocks(desc, 1.0 / fs, nsamp, fixed=True)
    want_st = z["end_state"].view(ob.STATE_DTYPE).reshape(st.shape)
    for k in range(desc.shape[0]):
        assert sha(iq[k]) == str(z["iq_sha256"][k]), k
        assert st[k].tobytes() == want_st[k].tobytes()
    # and it is a different signal from the floating-point variant
    zf, _, _ = load("static_F")
    assert str(z["iq_sha256"][0]) != str(zf["iq_sha256"][0])


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
def test_fixed_carrier_oracle_against_the_reference_build(oracle, pkg):
    r = ob.RefLoop("_fixed")
    ch = pkg.synth_descriptors(3, nch=16, seed=77)
    ch["carr_phase"] = np.floor(ch["carr_phase"] * 2.0 ** 32)
    ch["f_carr"][0, :4] = [0.0, 124999.0, -124999.0, 1e-3]
    ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    for b in range(3):
        want_iq, want_st = r.fill(ch[b], 1.0 / 1e6, 40001)
        iq, st, _ = oracle.fill_blocks(ch[b], 1.0 / 1e6, 40001, fixed=True)
        assert (iq[0] == want_iq).all() and st[0].tobytes() == want_st.tobytes()
