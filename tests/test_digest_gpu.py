"""gpsbb_device_digest (include/gpsbb.h): one 64-bit number per block of IQ in device memory — what bench.py compares instead
of the bytes when it cross-checks every block of the timed mode, and what the node driver's slots are checked by."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nsamp,nb", [(1, 3), (255, 4), (1024, 2), (100003, 5), (2500000, 2)])
def test_device_digest_equals_the_host_formula_and_sees_one_bit(pkg, synth, nsamp, nb):
    ch = pkg.synth_descriptors(nb, nch=5, seed=nsamp)
    b = synth.batch(ch, 1 / 25e6, nsamp)
    b.run()
    synth.sync()
    iq, _ = b.read()
    dptr = b.device_iq()
    got = synth.device_digest(dptr, nb, nsamp)
    want = pkg.block_digest_host(iq)
    assert got.dtype == np.uint64 and (got == want).all()
    assert len(set(got.tolist())) == nb  # different blocks, different numbers
    # one bit of one sample: another number; two samples swapped: another number (the sum is over position-dependent terms)
    other = iq.copy()
    other[nb - 1, nsamp // 2, 1] ^= 1
    assert (pkg.block_digest_host(other)[:nb - 1] == want[:nb - 1]).all() and pkg.block_digest_host(other)[nb - 1] != want[nb - 1]
    if nsamp > 2 and (iq[0, 0] != iq[0, 1]).any():
        other = iq.copy()
        other[0, [0, 1]] = other[0, [1, 0]]
        assert pkg.block_digest_host(other)[0] != want[0]
    b.close()


@pytest.mark.parametrize("fs,nsamp,nch,bps,device_only", [(25e6, 100003, 16, 5, True), (25e6, 4096 * 3, 7, 6, False), (2.6e6, 30000, 12, 4, True),
                                                        (1e6, 5000, 3, 3, False)])
def test_a_push_rendered_with_its_digests(pkg, synth, oracle, fs, nsamp, nch, bps, device_only):
    """GPSBB_PUSH_DIGEST: the synthesis kernel adds every block's digest up as it renders (k_synth_ev_digest, on the samples it has
    in registers: whole tiles, the ragged last tile of a block, blocks that do not start on 16 bytes) — or, for the kernels without
    such a variant (k_synth_pd at 2.6 MS/s, k_synth at 1 MS/s), a digest kernel runs behind them.  What pop_digest hands out is
    gpsbb_device_digest's number of the oracle's bytes, push after push of a chained stream; a push without the flag has no
    digests to hand out (GPSBB_E_STATE, nothing popped); pushes with and without the flag mix in one ring."""
    pushes = 4
    ch = pkg.synth_descriptors(pushes * bps, nch=nch, seed=int(nsamp) + nch)
    want_iq, want_st, _ = oracle.fill_blocks(ch, 1 / fs, nsamp, chain=True)
    want = pkg.block_digest_host(want_iq)
    st = synth.stream(nch, 1 / fs, nsamp, bps, depth=3, flags=pkg.CHAIN_CARRIER | (pkg.STREAM_DEVICE_ONLY if device_only else 0))
    flagged = [True, True, False, True]
    done = 0
    for k in range(pushes):
        st.push(ch[k * bps:(k + 1) * bps], digest=flagged[k])
        if st.pending == 3 or k == pushes - 1:
            while st.pending and (st.pending == 3 or k == pushes - 1):
                if flagged[done]:
                    iq, es, dig = st.pop_digest(copy=True)
                    assert (dig == want[done * bps:(done + 1) * bps]).all(), (done, dig, want[done * bps:(done + 1) * bps])
                else:
                    with pytest.raises(pkg.GpsbbError):
                        st.pop_digest()
                    iq, es = st.pop(copy=True)
                got = synth.device_read(iq, (bps, nsamp, 2)) if device_only else iq
                assert (got == want_iq[done * bps:(done + 1) * bps]).all(), done
                assert es["carr_phase"].tobytes() == want_st["carr_phase"][done * bps:(done + 1) * bps].tobytes()
                done += 1
    assert done == pushes
    st.close()
