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
