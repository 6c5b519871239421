"""The TX hand-off surface (pluto-gps-sim_amd/host/gpsbb_tx.c): the reference's one-buffer mutex/condvar
protocol (plutogpssim.c:2146-2158, 2689, 2757-2759) with a pluggable sink.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

PUSH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int16), C.c_size_t)


@pytest.fixture(scope="module")
def tx(pkg):
    pkg.build_frontend()
    L = pkg.fe_lib()
    L.gpsbb_tx_create.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, PUSH_FN, C.c_void_p]
    L.gpsbb_tx_begin.argtypes = [C.c_void_p]
    L.gpsbb_tx_begin.restype = C.POINTER(C.c_int16)
    L.gpsbb_tx_end.argtypes = [C.c_void_p]
    L.gpsbb_tx_destroy.argtypes = [C.c_void_p]
    L.gpsbb_tx_destroy.restype = None
    L.gpsbb_tx_delivered.argtypes = [C.c_void_p]
    L.gpsbb_tx_delivered.restype = C.c_ulong
    return L


def test_every_block_is_delivered_once_and_in_order(tx):
    nsamp, nblocks = 1000, 50
    got = []

    @PUSH_FN
    def push(user, iq, n):
        got.append(np.ctypeslib.as_array(iq, (2 * n,)).copy())
        return 0

    h = C.c_void_p()
    assert tx.gpsbb_tx_create(C.byref(h), nsamp, push, None) == 0
    for b in range(nblocks):
        buf = tx.gpsbb_tx_begin(h)                      # generator holds the mutex while it fills (c:2689)
        np.ctypeslib.as_array(buf, (2 * nsamp,))[:] = b
        assert tx.gpsbb_tx_end(h) == 0                  # signal + wait until the TX thread has copied it
    tx.gpsbb_tx_destroy(h)
    assert len(got) == nblocks
    for b, blk in enumerate(got):
        assert (blk == b).all()


def test_sink_error_stops_the_generator(tx):
    """A negative return of the sink plays iio_buffer_push failing (c:2153-2157): the generator sees `exit`."""
    calls = []

    @PUSH_FN
    def push(user, iq, n):
        calls.append(1)
        return -1 if len(calls) == 3 else 0

    h = C.c_void_p()
    assert tx.gpsbb_tx_create(C.byref(h), 16, push, None) == 0
    stopped_at = None
    for b in range(10):
        tx.gpsbb_tx_begin(h)
        if tx.gpsbb_tx_end(h) == 1:
            stopped_at = b
            break
    tx.gpsbb_tx_destroy(h)
    assert stopped_at is not None and stopped_at <= 4 and len(calls) == 3


def test_file_sink_writes_gps_sdr_sim_format(tx, tmp_path):
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    path = str(tmp_path / "iq.bin")
    f = libc.fopen(path.encode(), b"wb")
    push = C.cast(tx.gpsbb_tx_push_to_file, PUSH_FN)
    h = C.c_void_p()
    assert tx.gpsbb_tx_create(C.byref(h), 64, push, f) == 0
    want = []
    for b in range(5):
        buf = tx.gpsbb_tx_begin(h)
        blk = (np.arange(128) + 1000 * b).astype(np.int16)
        np.ctypeslib.as_array(buf, (128,))[:] = blk
        want.append(blk)
        tx.gpsbb_tx_end(h)
    tx.gpsbb_tx_destroy(h)
    libc.fclose(f)
    assert (np.fromfile(path, np.int16) == np.concatenate(want)).all()


@pytest.mark.gpu
def test_paced_consumer_soak_of_the_drop_in_call(pkg, tmp_path):
    """The drop-in shape under a consumer that takes blocks at a fixed rate, as the radio does (plutogpssim.c:2146-2158:
    iio_buffer_push blocks until the device has room; libiio queues 4 kernel buffers): gpsbb-sim's main loop (front end ->
    gpsbb_fill_block -> mutex/condvar hand-off, c:2655-2806) for 2000 blocks = 200 s of signal with the consumer paced at
    5 ms per 100 ms block (20 x real time; the 10 000-block runs at 100 x are in profiles/r04_paced_soak_1ms.json — here the
    period leaves room for a shared test box's scheduling noise).  No under-run, the call's p99 far below the period, and the kept
    blocks — before, at and after the 30 s nav refresh — are the golden vectors of the reference's own code."""
    import hashlib
    import json
    import subprocess
    import numpy as np
    from conftest import GOLDEN
    pkg.build_frontend()
    exe = os.path.join(os.path.dirname(pkg.LIB_PATH), "gpsbb-sim")
    z = np.load(os.path.join(GOLDEN, "static_F.npz"))
    nsamp = int(z["nsamp"])
    keep = [int(b) for b in z["blocks"] if int(b) < 2000]
    out, stats = str(tmp_path / "kept.bin"), str(tmp_path / "stats.json")
    subprocess.run([exe, "-e", os.path.join(GOLDEN, "synth3540.14n"), "-l", "30.286502,120.032669,100", "-s", "2600000",
                    "-d", "200", "-P", "5000", "-S", stats, "-k", ",".join(map(str, keep)), "-o", out], check=True,
                   stderr=subprocess.DEVNULL, timeout=600)
    st = json.load(open(stats))
    assert st["blocks"] == 2000 and st["delivered"] == 2000 and st["device_queue_blocks"] == 4
    # wall-clock properties of a shared test box are not the library's: the hard asserts are the bytes below; the timing is held
    # to what only a broken pipeline misses (the tight figures are measurements: profiles/r04_paced_soak_*.json), and to the
    # tight ones where GPSBB_TEST_TIMING=1 says the box is quiet
    if os.environ.get("GPSBB_TEST_TIMING"):
        assert st["underruns"] == 0, st
        assert st["fill_block_ms"]["p99"] < 5.0 and st["fill_block_ms"]["p50"] < 1.0, st
    else:
        assert st["underruns"] <= 20 and st["fill_block_ms"]["p50"] < 5.0, st
    iq = np.fromfile(out, np.int16).reshape(len(keep), nsamp, 2)
    for k in range(len(keep)):
        assert hashlib.sha256(iq[k].tobytes()).hexdigest() == str(z["iq_sha256"][k]), keep[k]
