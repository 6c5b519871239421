"""The C-ABI library builds, loads and exports every symbol include/gpsbb.h declares; host-side helpers
agree with the oracle.  No compute call needs a GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gpsbb.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpsbb_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_are_exported(pkg):
    L = pkg.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libgpsbb.so does not export %s" % n
    assert sorted(pkg.API_SYMBOLS) == names


def test_node_header_symbols_are_exported(pkg):
    """include/gpsbb_node.h (the N-GPU driver) lives in libgpsbb.so as well; the shard plan is pure arithmetic."""
    txt = open(os.path.join(ROOT, "include", "gpsbb_node.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(gpsbb_node_[a-z_0-9]+)\s*\(", txt)) - {"gpsbb_node_sink_fn"})
    assert names == sorted(pkg.NODE_API_SYMBOLS)
    for n in names:
        assert hasattr(pkg.lib(), n)
    # contiguous shards on whole pushes, nothing lost, nothing twice (BASELINE configs[4]: 36 000 blocks over 8 GPUs)
    for nblocks, nshards, bps in [(36000, 8, 400), (36000, 8, 32), (7, 3, 2), (5, 8, 1), (100, 1, 16), (33, 4, 16)]:
        first = pkg.node_plan(nblocks, nshards, bps)
        assert first[0] == 0 and first[-1] == nblocks and all(a <= b for a, b in zip(first, first[1:]))
        assert all(f % bps == 0 for f in first[:-1])
    # 90 slots of 400 blocks over 8 shards: shard g starts at floor(90 g / 8) * 400
    assert pkg.node_plan(36000, 8, 400) == [0, 4400, 8800, 13200, 18000, 22400, 26800, 31200, 36000]
    assert pkg.lib().gpsbb_node_create(None, None) == -1


def test_struct_sizes_match_header(pkg):
    assert pkg.CHAN_DTYPE.itemsize == 296 and pkg.STATE_DTYPE.itemsize == 40


def test_strerror_and_version(pkg):
    L = pkg.lib()
    assert L.gpsbb_version() == 1
    assert L.gpsbb_strerror(0) == b"ok"
    assert b"contract" in L.gpsbb_strerror(-2)


def test_codegen_and_tables_match_oracle(pkg, oracle):
    for prn in range(1, 33):
        assert (pkg.codegen(prn) == oracle.codegen(prn)).all()
    s, c = pkg.sincos_tables()
    os_, oc = oracle.tables()
    assert (s == os_).all() and (c == oc).all()
    with pytest.raises(pkg.GpsbbError):
        pkg.codegen(0)


def test_bad_arguments_are_rejected_without_a_device(pkg):
    L = pkg.lib()
    assert L.gpsbb_create(None, 0) == -1
    assert L.gpsbb_fill_block(None, None, 0, 0.0, 0, None, None) == -1
    assert L.gpsbb_chain_carrier_host(None, 1, 1, 1e-6, 10, None, 0) == -1
    ch = pkg.synth_descriptors(1, nch=2, seed=1)
    ch["gain"][0, 0] = float("nan")
    seed = np.zeros((1, 2))
    assert L.gpsbb_chain_carrier_host(ch.ctypes.data, 1, 2, 1e-6, 10, seed.ctypes.data, 1) == -2


def test_no_cpu_fallback(pkg):
    """Without a GPU the library must refuse, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert pkg.lib().gpsbb_create(C.byref(h), 0) == -6


def test_host_library_exports_its_headers(pkg):
    """libgpsfe.so exports every function include/gpsfe.h and include/gpsbb_tx.h declare."""
    pkg.build_frontend()
    L = pkg.fe_lib()
    for hdr in ("gpsfe.h", "gpsbb_tx.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names = sorted(set(re.findall(r"\b(gps(?:fe|bb_tx)_[a-z_0-9]+)\s*\(", txt)))
        assert len(names) >= 6, hdr
        for n in names:
            assert hasattr(L, n), "libgpsfe.so does not export %s" % n
