"""ctypes bindings for the CPU oracle and (when built) the verbatim reference slices.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CHAN_DTYPE = np.dtype([("prn", "<i4"), ("iword", "<i4"), ("ibit", "<i4"), ("icode", "<i4"),
                       ("f_carr", "<f8"), ("f_code", "<f8"), ("carr_phase", "<f8"),
                       ("code_phase", "<f8"), ("gain", "<f8"), ("dwrd", "<u4", (60,))])
STATE_DTYPE = np.dtype([("carr_phase", "<f8"), ("code_phase", "<f8"), ("iword", "<i4"),
                        ("ibit", "<i4"), ("icode", "<i4"), ("dataBit", "<i4"), ("codeCA", "<i4"),
                        ("_pad", "<i4")])
HAZ_DTYPE = np.dtype([("itable_512", "<u8"), ("dwrd_oob", "<u8")])
assert CHAN_DTYPE.itemsize == 296 and STATE_DTYPE.itemsize == 40


def build(force=False):
    """Compile the oracle (and the reference slices if /root/reference exists)."""
    so = os.path.join(HERE, "libgpsbb_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", HERE, "libgpsbb_oracle.so", "libgpsbb_oracle_O0.so"])
    return so


class Oracle:
    def __init__(self, variant=""):
        build()
        self.lib = C.CDLL(os.path.join(HERE, "libgpsbb_oracle%s.so" % variant))
        L = self.lib
        L.gpsbb_oracle_fill_blocks.restype = C.c_int
        L.gpsbb_oracle_fill_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gpsbb_oracle_fill_blocks_fixed.restype = C.c_int
        L.gpsbb_oracle_fill_blocks_fixed.argtypes = L.gpsbb_oracle_fill_blocks.argtypes
        L.gpsbb_oracle_tables.argtypes = [C.c_void_p, C.c_void_p]
        L.gpsbb_oracle_codegen.argtypes = [C.c_void_p, C.c_int]

    def tables(self):
        s = np.zeros(512, np.int32)
        c = np.zeros(512, np.int32)
        self.lib.gpsbb_oracle_tables(s.ctypes.data, c.ctypes.data)
        return s, c

    def codegen(self, prn):
        ca = np.full(1023, -1, np.int32)
        self.lib.gpsbb_oracle_codegen(ca.ctypes.data, prn)
        return ca

    def fill_blocks(self, ch, delt, nsamp, chain=False, want_iq=True, fixed=False):
        """ch: CHAN_DTYPE array [nblocks, nch] -> (iq int16 [nblocks, nsamp, 2], end_state, hazards).
        fixed=True: the reference's fixed-point carrier variant (carr_phase = 32-bit accumulator value)."""
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        if ch.ndim == 1:
            ch = ch[None, :]
        nb, nch = ch.shape
        iq = np.zeros((nb, nsamp, 2), np.int16) if want_iq else None
        st = np.zeros((nb, nch), STATE_DTYPE)
        hz = np.zeros(1, HAZ_DTYPE)
        fn = self.lib.gpsbb_oracle_fill_blocks_fixed if fixed else self.lib.gpsbb_oracle_fill_blocks
        rc = fn(ch.ctypes.data, nb, nch, delt, nsamp, int(chain),
                                               iq.ctypes.data if want_iq else None, st.ctypes.data,
                                               hz.ctypes.data)
        if rc != 0:
            raise ValueError("oracle rejected the descriptors (rc=%d)" % rc)
        return iq, st, hz[0]


REF_DIR = os.path.join(HERE, "_ref")


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libplutoref.so")) and \
        os.path.exists(os.path.join(REF_DIR, "libplutoref_fixed.so"))


class RefLoop:
    """The reference's own sample loop (plutogpssim.c:2690-2756 compiled verbatim), via ctypes."""

    def __init__(self, variant=""):
        self.lib = C.CDLL(os.path.join(REF_DIR, "libplutoref%s.so" % variant))
        L = self.lib
        L.ref_loop_fill.restype = C.c_int
        L.ref_loop_fill.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_tables.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]
        L.ref_codegen.argtypes = [C.c_void_p, C.c_int]

    def tables(self):
        ps, pc = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
        self.lib.ref_tables(C.byref(ps), C.byref(pc))
        return (np.ctypeslib.as_array(ps, (512,)).astype(np.int32),
                np.ctypeslib.as_array(pc, (512,)).astype(np.int32))

    def codegen(self, prn):
        ca = np.full(1023, -1, np.int32)
        self.lib.ref_codegen(ca.ctypes.data, prn)
        return ca

    def fill(self, ch, delt, nsamp):
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        nch = ch.shape[0]
        iq = np.zeros((nsamp, 2), np.int16)
        st = np.zeros(nch, STATE_DTYPE)
        rc = self.lib.ref_loop_fill(ch.ctypes.data, nch, delt, nsamp, iq.ctypes.data, st.ctypes.data)
        if rc != 0:
            raise ValueError("ref_loop_fill rc=%d" % rc)
        return iq, st


def run_ref_sim(nav, nblocks, nsamp, fs, llh=None, motion=None, max_chan=12, opt="", extra=()):
    """Run the scenario runner built from the reference's main() slices; returns (iq, desc, state)."""
    import tempfile
    exe = os.path.join(REF_DIR, "ref_sim%d%s" % (max_chan, opt))
    with tempfile.TemporaryDirectory() as td:
        iqp, dp, sp = (os.path.join(td, n) for n in ("iq.bin", "desc.bin", "state.bin"))
        cmd = [exe, "-e", nav, "-s", str(int(fs)), "-n", str(nsamp), "-b", str(nblocks), "-o", iqp,
               "-d", dp, "-S", sp]
        if motion:
            cmd += ["-u", motion]
        elif llh is not None:
            cmd += ["-l", "%s,%s,%s" % tuple(llh)]
        cmd += list(extra)
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        iq = np.fromfile(iqp, np.int16).reshape(nblocks, nsamp, 2)
        desc = np.fromfile(dp, CHAN_DTYPE).reshape(nblocks, max_chan)
        st = np.fromfile(sp, STATE_DTYPE).reshape(nblocks, max_chan)
    return iq, desc, st
