"""pluto-gps-sim_amd — MI355X-native GPS L1 C/A baseband IQ synthesis (libgpsbb) for Python callers.

This is only a ctypes veneer over the C ABI in include/gpsbb.h: the product is libgpsbb.so
(hand-written HIP for gfx950 in csrc/).  There is no Python or CPU implementation of the fill here; every
fill call goes through the shared library and raises when the library or a GPU is missing.

The directory name contains '-' (it mirrors the reference's repo name), so import it by path:

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("pluto_gps_sim_amd", ".../pluto-gps-sim_amd/__init__.py")
    mod = importlib.util.module_from_spec(spec); sys.modules[spec.name] = mod; spec.loader.exec_module(mod)

(`__graft_entry__.load_package()` and tests/conftest.py do exactly that.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# The product is libgpsbb.so: it reads no environment variable and exports include/gpsbb.h, nothing else.  The experiments
# build of the same sources (libgpsbb_exp.so: measurement knobs from the environment, gpsbb_test_* hooks) is what the
# tuning scripts under tools/ use when they say so (GPSBB_PY_LIB=exp, read HERE, in the Python veneer) and what
# exp_lib() hands to the NCO unit tests.
_which = os.environ.get("GPSBB_PY_LIB")  # "exp", the tag of a variant a tuning script built (libgpsbb_<tag>.so), or the PATH of a library
# (the deliberately wrong builds of the tests live in the tests' temporary directories, never beside the product)
LIB_PATH = _which if _which and os.sep in _which else os.path.join(HERE, "libgpsbb_%s.so" % _which if _which else "libgpsbb.so")
EXP_LIB_PATH = os.path.join(HERE, "libgpsbb_exp.so")

MAX_CHAN = 16
N_DWRD = 60

CHAN_DTYPE = np.dtype([("prn", "<i4"), ("iword", "<i4"), ("ibit", "<i4"), ("icode", "<i4"),
                       ("f_carr", "<f8"), ("f_code", "<f8"), ("carr_phase", "<f8"),
                       ("code_phase", "<f8"), ("gain", "<f8"), ("dwrd", "<u4", (N_DWRD,))])
STATE_DTYPE = np.dtype([("carr_phase", "<f8"), ("code_phase", "<f8"), ("iword", "<i4"),
                        ("ibit", "<i4"), ("icode", "<i4"), ("dataBit", "<i4"), ("codeCA", "<i4"),
                        ("_pad", "<i4")])
# The reference's channel_t (plutogpssim.h:152-174, FLOAT_CARR_PHASE build, LP64) as a numpy record: what a caller that kept the
# reference's structures hands to gpsbb_fill_block_ref.  tests/test_ref_layout.py checks every offset against offsetof() on the
# real header (in the build container, where /root/reference exists).
REF_CHANNEL_DTYPE = np.dtype([("prn", "<i4"), ("ca", "<i4", (1023,)), ("f_carr", "<f8"), ("f_code", "<f8"),
                              ("carr_phase", "<f8"), ("code_phase", "<f8"), ("g0_week", "<i4"), ("_p0", "<i4"),
                              ("g0_sec", "<f8"), ("sbf", "<u8", (50,)), ("dwrd", "<u8", (60,)), ("iword", "<i4"),
                              ("ibit", "<i4"), ("icode", "<i4"), ("dataBit", "<i4"), ("codeCA", "<i4"), ("_p1", "<i4"),
                              ("azel", "<f8", (2,)), ("rho0", "<f8", (8,))])
# ... and the same struct of a reference built WITHOUT FLOAT_CARR_PHASE (h:12 removed): h:160-161 put a 32-bit accumulator and its
# step where the double was (same size, so nothing else moves): what gpsbb_fill_block_ref_fixed is handed
REF_CHANNEL_FIXED_DTYPE = np.dtype([("prn", "<i4"), ("ca", "<i4", (1023,)), ("f_carr", "<f8"), ("f_code", "<f8"),
                                    ("carr_phase", "<u4"), ("carr_phasestep", "<i4"), ("code_phase", "<f8"), ("g0_week", "<i4"),
                                    ("_p0", "<i4"), ("g0_sec", "<f8"), ("sbf", "<u8", (50,)), ("dwrd", "<u8", (60,)),
                                    ("iword", "<i4"), ("ibit", "<i4"), ("icode", "<i4"), ("dataBit", "<i4"), ("codeCA", "<i4"),
                                    ("_p1", "<i4"), ("azel", "<f8", (2,)), ("rho0", "<f8", (8,))])


class RefLayout(C.Structure):
    """gpsbb_ref_layout_t (include/gpsbb.h): where the fields gpsbb_fill_block_ref reads and updates sit in the caller's channel_t"""
    _fields_ = [(n, C.c_size_t) for n in ("stride", "off_prn", "off_f_carr", "off_f_code", "off_carr_phase", "off_code_phase",
                                          "off_dwrd", "sizeof_dwrd_elem", "off_iword", "off_ibit", "off_icode", "off_dataBit",
                                          "off_codeCA")]


def ref_layout(dtype=None):
    dtype = REF_CHANNEL_DTYPE if dtype is None else dtype
    off = lambda n: dtype.fields[n][1]
    return RefLayout(dtype.itemsize, off("prn"), off("f_carr"), off("f_code"), off("carr_phase"), off("code_phase"), off("dwrd"), 8,
                     off("iword"), off("ibit"), off("icode"), off("dataBit"), off("codeCA"))


def ref_channels(d):
    """one block's descriptors (CHAN_DTYPE[nch]) as the reference's channel_t[] and gain[] (plutogpssim.c:2241)"""
    chan = np.zeros(d.shape[0], REF_CHANNEL_DTYPE)
    for f in ("prn", "f_carr", "f_code", "carr_phase", "code_phase", "iword", "ibit", "icode"):
        chan[f] = d[f]
    chan["dwrd"] = d["dwrd"]
    return chan, np.ascontiguousarray(d["gain"])


ROW_DTYPE = np.dtype([("n0", "<i4"), ("nav", "<u4"), ("xb", "<u8"), ("inc", "<i8")])
assert CHAN_DTYPE.itemsize == 296 and STATE_DTYPE.itemsize == 40 and ROW_DTYPE.itemsize == 24

CHAIN_CARRIER = 1
FIXED_CARRIER = 2
STREAM_DEVICE_ONLY = 4
OPT_SEED_WHERE, OPT_SYNTH_KERNEL, OPT_SKIP_SEED, OPT_CHAIN_WHERE = 1, 2, 3, 4
INFO_LAST_KERNEL, INFO_EXACT_RUNS, INFO_CHAIN_ON_DEVICE, INFO_CHAIN_FALLBACKS, INFO_CHAIN_TIES, INFO_CHAIN_REPAIRS = 1, 2, 3, 4, 5, 6
INFO_STREAMS, INFO_HW_QUEUES, INFO_TILES_RENDERED, INFO_PREPASS = 7, 8, 9, 10
NODE_INDEXED, NODE_CONCURRENT, NODE_DEVICE_ONLY, NODE_NO_AFFINITY, NODE_FIXED_CARRIER, NODE_INTERLEAVED, NODE_DIGESTS = 1, 2, 4, 8, 16, 32, 64
PUSH_NEW_CHAIN = 1
PUSH_DIGEST = 2
NODE_MAX_SHARDS = 64

ERRORS = {0: "GPSBB_OK", -1: "GPSBB_E_BADARG", -2: "GPSBB_E_BADCHAN", -3: "GPSBB_E_HIP", -4: "GPSBB_E_NOMEM",
          -5: "GPSBB_E_INTERNAL", -6: "GPSBB_E_NODEVICE", -7: "GPSBB_E_STATE"}

# every symbol include/gpsbb.h declares
API_SYMBOLS = [
    "gpsbb_create", "gpsbb_destroy", "gpsbb_strerror", "gpsbb_last_hip_error", "gpsbb_version",
    "gpsbb_fill_block", "gpsbb_fill_block_ex", "gpsbb_fill_block_ref", "gpsbb_fill_block_ref_fixed", "gpsbb_batch_create", "gpsbb_batch_destroy",
    "gpsbb_batch_iq_bytes", "gpsbb_batch_run", "gpsbb_sync", "gpsbb_batch_read", "gpsbb_batch_device_iq",
    "gpsbb_get_hazards", "gpsbb_device_read", "gpsbb_device_digest", "gpsbb_slot_digest", "gpsbb_batch_last_timing", "gpsbb_batch_timing_stats", "gpsbb_fill_ceiling", "gpsbb_stream_create",
    "gpsbb_stream_destroy", "gpsbb_stream_push", "gpsbb_stream_pop", "gpsbb_stream_pending", "gpsbb_stream_timing_stats",
    "gpsbb_codegen", "gpsbb_sincos_tables", "gpsbb_chain_carrier_host", "gpsbb_chain_carrier", "gpsbb_set_option",
    "gpsbb_get_info", "gpsbb_stream_reset", "gpsbb_device_affinity", "gpsbb_stream_push_ex", "gpsbb_stream_pop_digest", "gpsbb_host_register", "gpsbb_host_unregister",
]
# ... and include/gpsbb_node.h
NODE_API_SYMBOLS = ["gpsbb_node_create", "gpsbb_node_run", "gpsbb_node_run_digest", "gpsbb_node_slot_digests", "gpsbb_node_destroy", "gpsbb_node_plan", "gpsbb_node_begin", "gpsbb_node_feed", "gpsbb_node_end"]


class GpsbbError(RuntimeError):
    def __init__(self, rc, what=""):
        self.rc = rc
        super().__init__("%s failed: %s (%d)" % (what, ERRORS.get(rc, "?"), rc))


def build(force=False):
    """Compile csrc/ for gfx950 with hipcc into libgpsbb.so next to this file (in-tree, so it travels)."""
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(EXP_LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, "csrc", f)) > min(os.path.getmtime(LIB_PATH), os.path.getmtime(EXP_LIB_PATH))
            for f in os.listdir(os.path.join(HERE, "csrc"))):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc")])
    return LIB_PATH


_lib = None


def lib():
    """The loaded libgpsbb.so (never a fallback: raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgpsbb.so is not built (run __graft_entry__.build() or make -C %s/csrc)" % HERE)
        L = C.CDLL(LIB_PATH)
        vp, i, d, u = C.c_void_p, C.c_int, C.c_double, C.c_uint
        L.gpsbb_create.argtypes = [C.POINTER(vp), i]
        L.gpsbb_destroy.argtypes = [vp]
        L.gpsbb_destroy.restype = None
        L.gpsbb_strerror.argtypes = [i]
        L.gpsbb_strerror.restype = C.c_char_p
        L.gpsbb_last_hip_error.argtypes = [vp]
        L.gpsbb_set_option.argtypes = [vp, i, C.c_long]
        L.gpsbb_get_info.argtypes = [vp, i, C.POINTER(C.c_uint64)]
        L.gpsbb_fill_block.argtypes = [vp, vp, i, d, i, vp, vp]
        L.gpsbb_fill_block_ex.argtypes = [vp, vp, i, d, i, u, vp, vp]
        L.gpsbb_fill_block_ref.argtypes = [vp, vp, vp, i, vp, d, i, vp]
        L.gpsbb_fill_block_ref_fixed.argtypes = [vp, vp, vp, C.c_size_t, i, vp, d, i, vp]
        L.gpsbb_batch_create.argtypes = [vp, vp, i, i, d, i, u, C.POINTER(vp)]
        L.gpsbb_batch_destroy.argtypes = [vp]
        L.gpsbb_batch_destroy.restype = None
        L.gpsbb_batch_iq_bytes.argtypes = [vp]
        L.gpsbb_batch_iq_bytes.restype = C.c_size_t
        L.gpsbb_batch_run.argtypes = [vp, vp]
        L.gpsbb_sync.argtypes = [vp]
        L.gpsbb_batch_read.argtypes = [vp, vp, vp]
        L.gpsbb_batch_device_iq.argtypes = [vp]
        L.gpsbb_batch_device_iq.restype = vp
        L.gpsbb_get_hazards.argtypes = [vp, vp, i]
        L.gpsbb_device_read.argtypes = [vp, vp, vp, C.c_size_t]
        L.gpsbb_device_digest.argtypes = [vp, vp, C.c_long, C.c_int, vp]
        L.gpsbb_batch_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.gpsbb_batch_timing_stats.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                               C.POINTER(C.c_float), i]
        L.gpsbb_fill_ceiling.argtypes = [vp, vp, C.c_size_t, i, C.POINTER(C.c_float)]
        L.gpsbb_stream_create.argtypes = [vp, i, d, i, i, i, u, C.POINTER(vp)]
        L.gpsbb_stream_destroy.argtypes = [vp]
        L.gpsbb_stream_destroy.restype = None
        L.gpsbb_stream_push.argtypes = [vp, vp]
        L.gpsbb_stream_pop.argtypes = [vp, C.POINTER(vp), vp]
        L.gpsbb_stream_pending.argtypes = [vp]
        L.gpsbb_stream_timing_stats.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_float), C.POINTER(C.c_float), i]
        L.gpsbb_codegen.argtypes = [i, vp]
        L.gpsbb_sincos_tables.argtypes = [vp, vp]
        L.gpsbb_chain_carrier_host.argtypes = [vp, i, i, d, i, vp, i]
        L.gpsbb_chain_carrier.argtypes = [vp, vp, i, i, d, i, vp, vp]
        L.gpsbb_stream_reset.argtypes = [vp]
        L.gpsbb_stream_push_ex.argtypes = [vp, vp, u]
        if hasattr(L, "gpsbb_stream_pop_digest"):  # (an older build loaded for an A/B: tools/ab_lib.sh)
            L.gpsbb_stream_pop_digest.argtypes = [vp, C.POINTER(vp), vp, vp]
        L.gpsbb_device_affinity.argtypes = [i, C.POINTER(i), C.c_char_p, C.c_size_t]
        L.gpsbb_node_create.argtypes = [C.POINTER(vp), vp]
        L.gpsbb_node_destroy.argtypes = [vp]
        L.gpsbb_node_destroy.restype = None
        L.gpsbb_node_run.argtypes = [vp, vp, C.c_long, vp, vp, vp]
        L.gpsbb_node_run_digest.argtypes = [vp, vp, C.c_long, vp, vp]
        L.gpsbb_node_slot_digests.argtypes = [vp, i, vp, i]
        L.gpsbb_slot_digest.argtypes = [vp, vp, C.c_long, C.c_int, vp]
        L.gpsbb_host_register.argtypes = [vp, vp, C.c_size_t]
        L.gpsbb_host_unregister.argtypes = [vp, vp]
        L.gpsbb_node_begin.argtypes = [vp, vp, vp]
        L.gpsbb_node_feed.argtypes = [vp, vp, C.c_long]
        L.gpsbb_node_end.argtypes = [vp, vp]
        L.gpsbb_node_plan.argtypes = [C.c_long, i, i, C.POINTER(C.c_long)]
        _lib = L
    return _lib


_exp_lib = None


def exp_lib():
    """The experiments build (libgpsbb_exp.so) with the gpsbb_test_* hooks of csrc/gpsbb_testhooks.h: the shared NCO code
    (gpsbb_nco.h) compiled for the host, which tests/test_nco_host.py checks against brute-force stepping without a GPU."""
    global _exp_lib
    if _exp_lib is None:
        if not os.path.exists(EXP_LIB_PATH):
            raise RuntimeError("libgpsbb_exp.so is not built (make -C %s/csrc)" % HERE)
        L = C.CDLL(EXP_LIB_PATH)
        i, d, u, vp = C.c_int, C.c_double, C.c_uint, C.c_void_p
        L.gpsbb_test_carr_jump.argtypes = [d, d, C.c_longlong]
        L.gpsbb_test_carr_jump.restype = d
        L.gpsbb_test_code_jump.argtypes = [d, d, C.c_longlong, C.POINTER(C.c_longlong)]
        L.gpsbb_test_code_jump.restype = d
        L.gpsbb_test_build_rows.argtypes = [i, d, d, u, i, vp, i, C.POINTER(d), C.POINTER(u)]
        L.gpsbb_test_row_bound.argtypes = [i, d, i]
        L.gpsbb_test_row_bound.restype = C.c_ulonglong
        L.gpsbb_test_carr_predict.argtypes = [d, d, i]
        L.gpsbb_test_carr_predict.restype = d
        L.gpsbb_test_fixed_tile_index.argtypes = [C.c_uint32, C.c_int32, i]
        L.gpsbb_test_fixed_tile_index.restype = d
        _exp_lib = L
    return _exp_lib


def _chk(rc, what):
    if rc != 0:
        raise GpsbbError(rc, what)


def _as_chan(ch):
    ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
    if ch.ndim == 1:
        ch = ch[None, :]
    if ch.ndim != 2:
        raise ValueError("descriptors must be [nblocks, nch]")
    return ch


# ---- host helpers ------------------------------------------------------------------------------------

def codegen(prn):
    ca = np.zeros(1023, np.uint8)
    _chk(lib().gpsbb_codegen(prn, ca.ctypes.data), "gpsbb_codegen")
    return ca


def sincos_tables():
    s = np.zeros(512, np.int32)
    c = np.zeros(512, np.int32)
    _chk(lib().gpsbb_sincos_tables(s.ctypes.data, c.ctypes.data), "gpsbb_sincos_tables")
    return s, c


def chain_carrier_host(ch, delt, nsamp, nthreads=0):
    ch = _as_chan(ch)
    nb, nch = ch.shape
    seed = np.zeros((nb, nch), np.float64)
    _chk(lib().gpsbb_chain_carrier_host(ch.ctypes.data, nb, nch, delt, nsamp, seed.ctypes.data, nthreads),
         "gpsbb_chain_carrier_host")
    return seed


# ---- device objects ----------------------------------------------------------------------------------

class Synth:
    """One libgpsbb handle: one GPU, one producer thread (gpsbb_create / gpsbb_destroy)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _chk(lib().gpsbb_create(C.byref(self._h), device), "gpsbb_create")

    def close(self):
        if self._h:
            lib().gpsbb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def fill_block(self, ch, delt, nsamp, flags=0, out=None):
        """gpsbb_fill_block(_ex): ch = CHAN_DTYPE[nch] -> (int16 [nsamp,2], STATE_DTYPE[nch]); out: the caller's iq_buff"""
        ch = np.ascontiguousarray(ch, dtype=CHAN_DTYPE)
        iq = np.empty((nsamp, 2), np.int16) if out is None else out
        assert iq.dtype == np.int16 and iq.flags.c_contiguous and iq.size >= 2 * nsamp
        st = np.zeros(ch.shape[0], STATE_DTYPE)
        _chk(lib().gpsbb_fill_block_ex(self._h, ch.ctypes.data, ch.shape[0], delt, nsamp, flags, iq.ctypes.data,
                                       st.ctypes.data), "gpsbb_fill_block_ex")
        return iq, st

    def fill_block_ref(self, chan, gain, delt, nsamp, iq, layout=None):
        """gpsbb_fill_block_ref: the reference's own channel_t[] (REF_CHANNEL_DTYPE, updated in place like the loop does,
        c:2709-2746) and gain[]; iq: int16 [nsamp, 2], the caller's iq_buff (c:84)"""
        layout = layout or ref_layout(chan.dtype)
        _chk(lib().gpsbb_fill_block_ref(self._h, chan.ctypes.data, C.byref(layout), chan.shape[0], gain.ctypes.data, delt, nsamp,
                                        iq.ctypes.data), "gpsbb_fill_block_ref")

    def host_register(self, arr):
        """gpsbb_host_register: fills into `arr` (the caller's iq_buff, kept alive by the caller) are rendered straight into it"""
        _chk(lib().gpsbb_host_register(self._h, arr.ctypes.data, arr.nbytes), "gpsbb_host_register")

    def host_unregister(self, arr):
        _chk(lib().gpsbb_host_unregister(self._h, arr.ctypes.data), "gpsbb_host_unregister")

    def batch(self, ch, delt, nsamp, flags=0):
        return Batch(self, ch, delt, nsamp, flags)

    def stream(self, nch, delt, nsamp, blocks_per_slot, depth=3, flags=0):
        return Stream(self, nch, delt, nsamp, blocks_per_slot, depth, flags)

    def sync(self):
        _chk(lib().gpsbb_sync(self._h), "gpsbb_sync")

    def device_read(self, d_ptr, shape, dtype=np.int16):
        """gpsbb_device_read: a numpy array filled from device memory the library handed out"""
        out = np.empty(shape, dtype)
        _chk(lib().gpsbb_device_read(self._h, out.ctypes.data, d_ptr, out.nbytes), "gpsbb_device_read")
        return out

    def slot_digest(self, d_ptr, nblocks, nsamp):
        """gpsbb_slot_digest: digests of a slot known to be complete (what pop returned), without draining the handle"""
        out = np.zeros(nblocks, np.uint64)
        _chk(lib().gpsbb_slot_digest(self._h, C.c_void_p(int(d_ptr)), nblocks, nsamp, out.ctypes.data), "gpsbb_slot_digest")
        return out

    def device_digest(self, d_ptr, nblocks, nsamp):
        """gpsbb_device_digest: one uint64 per block of IQ in device memory (see block_digest_host)"""
        out = np.empty(nblocks, np.uint64)
        for k0 in range(0, nblocks, 65535):
            n = min(65535, nblocks - k0)
            _chk(lib().gpsbb_device_digest(self._h, d_ptr + k0 * nsamp * 4, n, nsamp, out[k0:].ctypes.data), "gpsbb_device_digest")
        return out

    def hazards(self, reset=False):
        v = np.zeros(2, np.uint64)
        _chk(lib().gpsbb_get_hazards(self._h, v.ctypes.data, int(reset)), "gpsbb_get_hazards")
        return {"itable_512": int(v[0]), "dwrd_oob": int(v[1])}

    def set_option(self, option, value):
        """gpsbb_set_option: OPT_SEED_WHERE / OPT_SYNTH_KERNEL / OPT_SKIP_SEED (per handle)"""
        _chk(lib().gpsbb_set_option(self._h, option, value), "gpsbb_set_option")

    def info(self, what):
        """gpsbb_get_info: INFO_LAST_KERNEL / INFO_EXACT_RUNS"""
        v = C.c_uint64()
        _chk(lib().gpsbb_get_info(self._h, what, C.byref(v)), "gpsbb_get_info")
        return int(v.value)

    def chain_carrier(self, ch, delt, nsamp, want_seeds=True):
        """gpsbb_chain_carrier: the exact carrier chain on the device, nothing rendered -> (start phase of every block
        [nblocks, nch] or None, phase after the last block [nch])"""
        ch = _as_chan(ch)
        nb, nch = ch.shape
        seed = np.zeros((nb, nch), np.float64) if want_seeds else None
        end = np.zeros(nch, np.float64)
        _chk(lib().gpsbb_chain_carrier(self._h, ch.ctypes.data, nb, nch, delt, nsamp,
                                       seed.ctypes.data if want_seeds else None, end.ctypes.data), "gpsbb_chain_carrier")
        return seed, end

    def shard_seed(self, ch, b0, delt, nsamp):
        """The exact carr_phase block b0 of the stream `ch` starts from (per channel): the end of the device-side chain
        over the blocks before it for a channel that keeps its prn, the descriptor's own phase otherwise."""
        ch = _as_chan(ch)
        if b0 == 0:
            return ch["carr_phase"][0].copy()
        _, end = self.chain_carrier(ch[:b0], delt, nsamp, want_seeds=False)
        cont = (ch["prn"][b0] > 0) & (ch["prn"][b0] == ch["prn"][b0 - 1])
        return np.where(cont, end, ch["carr_phase"][b0])

    def fill_ceiling(self, d_ptr, nbytes, iters=10):
        ms = C.c_float()
        _chk(lib().gpsbb_fill_ceiling(self._h, d_ptr, nbytes, iters, C.byref(ms)), "gpsbb_fill_ceiling")
        return ms.value


class Batch:
    """Block descriptors resident in HBM (gpsbb_batch_*)."""

    def __init__(self, synth, ch, delt, nsamp, flags=0):
        ch = _as_chan(ch)
        self.synth = synth
        self.nblocks, self.nch = ch.shape
        self.nsamp = nsamp
        self._b = C.c_void_p()
        _chk(lib().gpsbb_batch_create(synth._h, ch.ctypes.data, self.nblocks, self.nch, delt, nsamp, flags,
                                      C.byref(self._b)), "gpsbb_batch_create")

    def close(self):
        if self._b:
            lib().gpsbb_batch_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def iq_bytes(self):
        return lib().gpsbb_batch_iq_bytes(self._b)

    def run(self, d_iq=None):
        """Enqueue seeding pre-pass + synthesis; d_iq = device pointer (int) or None for the internal buffer."""
        _chk(lib().gpsbb_batch_run(self._b, d_iq), "gpsbb_batch_run")

    def read(self, want_iq=True):
        iq = np.empty((self.nblocks, self.nsamp, 2), np.int16) if want_iq else None
        st = np.zeros((self.nblocks, self.nch), STATE_DTYPE)
        _chk(lib().gpsbb_batch_read(self._b, iq.ctypes.data if want_iq else None, st.ctypes.data),
             "gpsbb_batch_read")
        return iq, st

    def timing(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _chk(lib().gpsbb_batch_last_timing(self._b, C.byref(a), C.byref(b), C.byref(c)), "gpsbb_batch_last_timing")
        return {"ms_seed": a.value, "ms_synth": b.value, "ms_total": c.value}

    def timing_stats(self, reset=True):
        n, a, b, c = C.c_int(), C.c_float(), C.c_float(), C.c_float()
        _chk(lib().gpsbb_batch_timing_stats(self._b, C.byref(n), C.byref(a), C.byref(b), C.byref(c), int(reset)),
             "gpsbb_batch_timing_stats")
        return {"runs": n.value, "ms_seed_sum": a.value, "ms_synth_sum": b.value, "ms_total_sum": c.value}

    def device_iq(self):
        return lib().gpsbb_batch_device_iq(self._b)


class Stream:
    """Time-sharded streaming with pinned host gather (gpsbb_stream_*)."""

    def __init__(self, synth, nch, delt, nsamp, blocks_per_slot, depth=3, flags=0):
        self.synth = synth
        self.nch, self.nsamp, self.bps = nch, nsamp, blocks_per_slot
        self.device_only = bool(flags & STREAM_DEVICE_ONLY)
        self._s = C.c_void_p()
        _chk(lib().gpsbb_stream_create(synth._h, nch, delt, nsamp, blocks_per_slot, depth, flags,
                                       C.byref(self._s)), "gpsbb_stream_create")

    def close(self):
        if self._s:
            lib().gpsbb_stream_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, ch, new_chain=False, digest=False):
        """gpsbb_stream_push(_ex); digest: GPSBB_PUSH_DIGEST — the push is rendered with its blocks' digests (pop_digest)"""
        ch = _as_chan(ch)
        if ch.shape != (self.bps, self.nch):
            raise ValueError("push expects [blocks_per_slot, nch] descriptors")
        flags = (PUSH_NEW_CHAIN if new_chain else 0) | (PUSH_DIGEST if digest else 0)
        if flags:
            _chk(lib().gpsbb_stream_push_ex(self._s, ch.ctypes.data, flags), "gpsbb_stream_push_ex")
        else:
            _chk(lib().gpsbb_stream_push(self._s, ch.ctypes.data), "gpsbb_stream_push")

    def pop_digest(self, copy=True):
        """gpsbb_stream_pop_digest: pop() plus the popped push's block digests (uint64 [blocks_per_slot])"""
        p = C.c_void_p()
        st = np.zeros((self.bps, self.nch), STATE_DTYPE)
        dig = np.zeros(self.bps, np.uint64)
        _chk(lib().gpsbb_stream_pop_digest(self._s, C.byref(p), st.ctypes.data, dig.ctypes.data), "gpsbb_stream_pop_digest")
        if self.device_only:
            return p.value, st, dig
        n = self.bps * self.nsamp * 2
        view = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), (n,)).reshape(self.bps, self.nsamp, 2)
        return (view.copy() if copy else view), st, dig

    def pop(self, copy=True):
        """(IQ [blocks_per_slot, nsamp, 2] int16 in the slot's pinned host buffer, end states); a stream created with
        STREAM_DEVICE_ONLY returns the slot's device pointer (an int) instead of the array."""
        p = C.c_void_p()
        st = np.zeros((self.bps, self.nch), STATE_DTYPE)
        _chk(lib().gpsbb_stream_pop(self._s, C.byref(p), st.ctypes.data), "gpsbb_stream_pop")
        if self.device_only:
            return p.value, st
        n = self.bps * self.nsamp * 2
        view = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), (n,)).reshape(self.bps, self.nsamp, 2)
        return (view.copy() if copy else view), st

    @property
    def pending(self):
        return lib().gpsbb_stream_pending(self._s)

    def reset(self):
        """gpsbb_stream_reset: a new stream on the same ring (every slot popped); the next push is block 0 again"""
        _chk(lib().gpsbb_stream_reset(self._s), "gpsbb_stream_reset")

    def timing_stats(self, reset=True):
        n, a, b = C.c_int(), C.c_float(), C.c_float()
        _chk(lib().gpsbb_stream_timing_stats(self._s, C.byref(n), C.byref(a), C.byref(b), int(reset)),
             "gpsbb_stream_timing_stats")
        return {"runs": n.value, "ms_seed_sum": a.value, "ms_synth_sum": b.value}


# ---- synthetic descriptors (BASELINE / SURVEY section 8d, workload M2) -------------------------------

# ---- include/gpsbb_node.h: one process, N handles, one sink ------------------------------------------

class _NodeConfig(C.Structure):
    _fields_ = [("nshards", C.c_int), ("devices", C.POINTER(C.c_int)), ("nch", C.c_int), ("delt", C.c_double), ("nsamp", C.c_int),
                ("blocks_per_slot", C.c_int), ("depth", C.c_int), ("flags", C.c_uint)]


class _NodeShardStats(C.Structure):
    _fields_ = [("first_block", C.c_long), ("nblocks", C.c_long), ("device", C.c_int), ("numa_node", C.c_int), ("cpus_bound", C.c_int),
                ("seed_seconds", C.c_double), ("busy_seconds", C.c_double), ("wait_seconds", C.c_double)]


class _NodeStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("blocks", C.c_long), ("nshards", C.c_int), ("shard", _NodeShardStats * NODE_MAX_SHARDS)]


NODE_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int)


def node_plan(nblocks, nshards, blocks_per_slot):
    first = (C.c_long * (nshards + 1))()
    _chk(lib().gpsbb_node_plan(nblocks, nshards, blocks_per_slot, first), "gpsbb_node_plan")
    return list(first)


def device_affinity(device):
    """gpsbb_device_affinity -> (numa node, "cpu list")"""
    node = C.c_int(-1)
    buf = C.create_string_buffer(512)
    _chk(lib().gpsbb_device_affinity(device, C.byref(node), buf, 512), "gpsbb_device_affinity")
    return node.value, buf.value.decode()


class Node:
    """gpsbb_node_*: nshards producer threads (one handle + one ring each, bound next to their GPU), one sink."""

    def __init__(self, nshards, nch, delt, nsamp, blocks_per_slot, depth=3, flags=0, devices=None):
        self.nshards, self.nch, self.nsamp = nshards, nch, nsamp
        dev = (C.c_int * nshards)(*(devices if devices is not None else range(nshards)))
        cfg = _NodeConfig(nshards, dev, nch, delt, nsamp, blocks_per_slot, depth, flags)
        self._n = C.c_void_p()
        _chk(lib().gpsbb_node_create(C.byref(self._n), C.byref(cfg)), "gpsbb_node_create")

    def close(self):
        if self._n:
            lib().gpsbb_node_destroy(self._n)
            self._n = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run(self, ch, sink, expect_stop=False):
        """sink(iq_ptr, first_block, nblocks, shard) -> int (< 0 stops); returns the run's statistics as a dict.
        sink may also be a C function pointer (int) for a sink that lives in native code."""
        ch = _as_chan(ch)
        if ch.ndim != 2 or ch.shape[1] != self.nch:
            raise ValueError("descriptors of shape (nblocks, %d) wanted, got %r" % (self.nch, ch.shape))  # (C would read out of bounds)
        raised = []
        if callable(sink):
            def guarded(user, iq, first, nb, shard):
                # an exception must not vanish inside ctypes (which would turn it into "return 0": the run goes on as if the
                # sink were fine): it stops the run and comes back out of Node.run
                try:
                    return int(sink(iq, first, nb, shard) or 0)
                except BaseException as e:  # noqa: BLE001
                    raised.append(e)
                    return -1
            cb = NODE_SINK(guarded)
            fn = C.cast(cb, C.c_void_p)
        else:
            cb, fn = None, C.c_void_p(sink)
        st = _NodeStats()
        rc = lib().gpsbb_node_run(self._n, ch.ctypes.data, ch.shape[0], fn, None, C.byref(st))
        del cb
        if raised:
            raise raised[0]
        if rc != 0 and not (expect_stop and rc == -7):
            raise GpsbbError(rc, "gpsbb_node_run")
        return {"rc": rc, "seconds": st.seconds, "blocks": st.blocks,
                "shards": [{k: getattr(st.shard[g], k) for k, _ in _NodeShardStats._fields_} for g in range(st.nshards)]}

    def run_digest(self, ch):
        """gpsbb_node_run_digest: the driver's own sink — one 64-bit digest per block, taken on the GPU that rendered it by the
        shard's producer thread (no Python in the data path); returns (statistics, uint64 [nblocks])"""
        ch = _as_chan(ch)
        if ch.ndim != 2 or ch.shape[1] != self.nch:
            raise ValueError("descriptors of shape (nblocks, %d) wanted, got %r" % (self.nch, ch.shape))
        digs = np.zeros(ch.shape[0], np.uint64)
        st = _NodeStats()
        _chk(lib().gpsbb_node_run_digest(self._n, ch.ctypes.data, ch.shape[0], digs.ctypes.data, C.byref(st)), "gpsbb_node_run_digest")
        return {"rc": 0, "seconds": st.seconds, "blocks": st.blocks,
                "shards": [{k: getattr(st.shard[g], k) for k, _ in _NodeShardStats._fields_} for g in range(st.nshards)]}, digs

    def slot_digests(self, shard, nblocks):
        """gpsbb_node_slot_digests, from inside a sink of a NODE_DIGESTS node: the digests the blocks just handed over were rendered with"""
        out = np.zeros(nblocks, np.uint64)
        _chk(lib().gpsbb_node_slot_digests(self._n, shard, out.ctypes.data, nblocks), "gpsbb_node_slot_digests")
        return out

    def begin(self, sink):
        """gpsbb_node_begin: an incremental run; feed() the stream as it comes, end() when it is over"""
        self._raised = []

        def guarded(user, iq, first, nb, shard):
            try:
                return int(sink(iq, first, nb, shard) or 0)
            except BaseException as e:  # noqa: BLE001
                self._raised.append(e)
                return -1
        self._cb = NODE_SINK(guarded)
        _chk(lib().gpsbb_node_begin(self._n, C.cast(self._cb, C.c_void_p), None), "gpsbb_node_begin")

    def feed(self, ch):
        ch = _as_chan(ch)
        if ch.ndim != 2 or ch.shape[1] != self.nch:
            raise ValueError("descriptors of shape (nblocks, %d) wanted, got %r" % (self.nch, ch.shape))
        _chk(lib().gpsbb_node_feed(self._n, ch.ctypes.data, ch.shape[0]), "gpsbb_node_feed")

    def end(self, expect_stop=False):
        st = _NodeStats()
        rc = lib().gpsbb_node_end(self._n, C.byref(st))
        self._cb = None
        if self._raised:
            raise self._raised[0]
        if rc != 0 and not (expect_stop and rc == -7):
            raise GpsbbError(rc, "gpsbb_node_end")
        return {"rc": rc, "seconds": st.seconds, "blocks": st.blocks,
                "shards": [{k: getattr(st.shard[g], k) for k, _ in _NodeShardStats._fields_} for g in range(st.nshards)]}


def block_digest_host(iq):
    """gpsbb_device_digest's number for blocks in host memory: iq int16 [..., nsamp, 2] -> uint64 [...]"""
    a = np.ascontiguousarray(iq, np.int16)
    w = a.view(np.uint32).reshape(a.shape[:-2] + (a.shape[-2],)).astype(np.uint64)
    with np.errstate(over="ignore"):
        m = (np.arange(a.shape[-2], dtype=np.uint64) * np.uint64(0x9E3779BA) + np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
        return (w * m).sum(axis=-1, dtype=np.uint64)


class SplitMix64:
    """Counter-mode splitmix64: draw k (k = 1, 2, ...) is mix(seed + k*0x9E3779B97F4A7C15) — the sequence
    the scalar generator produces, evaluated vectorised."""

    def __init__(self, seed):
        self.seed = np.uint64(seed)
        self.k = 0

    def take_at(self, start, n):
        """draws start+1 .. start+n of the sequence (counter mode: any slice costs what it holds)"""
        with np.errstate(over="ignore"):
            idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
            z = self.seed + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def take(self, n):
        z = self.take_at(self.k, n)
        self.k += n
        return z

    def u01(self, shape):
        n = int(np.prod(shape))
        return ((self.take(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)

    def u32(self, shape):
        n = int(np.prod(shape))
        return (self.take(n) >> np.uint64(32)).reshape(shape)

    def rows(self, nrows, width, first, count):
        """rows [first, first+count) of the (nrows, width) table the next nrows*width draws fill; the generator moves on
        past the whole table, so that what follows does not depend on which rows were asked for"""
        z = self.take_at(self.k + first * width, count * width).reshape(count, width)
        self.k += nrows * width
        return z


def synth_descriptors(nblocks, nch=16, seed=0x5EED, max_doppler=5000.0, first=0, count=None, fields=None):
    """Seeded descriptor-level constellation: PRN 1..nch, f_carr ~ U(-max,max) Hz, f_code = 1.023e6 +
    f_carr/1540, code_phase ~ U[0,1023), carr_phase ~ U[0,1), gain ~ U(0.30,0.80), random 30-bit nav
    words, iword in [9,58], ibit in 0..29, icode in 0..19.
    first / count: only blocks [first, first+count) of the nblocks-block table — the same rows the whole table has, at
    the cost of those rows (a rank of a time-sharded run builds its own shard only).  fields: only these (the others stay
    zero): what a carrier-chain-only call reads is ("prn", "f_carr", "carr_phase")."""
    count = nblocks - first if count is None else count
    g = SplitMix64(seed)
    ch = np.zeros((count, nch), CHAN_DTYPE)

    def want(f):
        return fields is None or f in fields

    def u01(width=nch):
        return (g.rows(nblocks, width, first, count) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def skip(width=nch):
        g.k += nblocks * width

    ch["prn"] = np.arange(1, nch + 1, dtype=np.int32)[None, :]
    if want("f_carr") or want("f_code"):
        ch["f_carr"] = (u01() * 2.0 - 1.0) * max_doppler
        ch["f_code"] = 1.023e6 + ch["f_carr"] / 1540.0
    else:
        skip()
    if want("code_phase"):
        ch["code_phase"] = u01() * 1023.0
    else:
        skip()
    if want("carr_phase"):
        ch["carr_phase"] = u01()
    else:
        skip()
    if fields is not None and not (set(fields) - {"prn", "f_carr", "f_code", "code_phase", "carr_phase"}):
        return ch
    ch["gain"] = 0.30 + 0.50 * u01()
    ch["iword"] = 9 + ((g.rows(nblocks, nch, first, count) >> np.uint64(32)) % np.uint64(50)).astype(np.int32)
    ch["ibit"] = ((g.rows(nblocks, nch, first, count) >> np.uint64(32)) % np.uint64(30)).astype(np.int32)
    ch["icode"] = ((g.rows(nblocks, nch, first, count) >> np.uint64(32)) % np.uint64(20)).astype(np.int32)
    ch["dwrd"] = ((g.rows(nblocks, nch * N_DWRD, first, count) >> np.uint64(32)) & np.uint64(0x3FFFFFFF)).astype(np.uint32).reshape(count, nch, N_DWRD)
    return ch


def grazing_descriptors(nblocks, nch, fs, nsamp, offsets, seed=1, max_doppler=5000.0, fixed=False, samples=None, tol=0.25):
    """Adversarial descriptors for the model kernels (k_synth_ev / k_synth_pd): every (block, channel) is aimed so that
    ONE of its NCOs lands within `k` units of 2^-32 of an integer AT a chosen sample n — the reference's own state there,
    not the linear model's: the phase is refined with the exact jump-ahead (gpsbb_nco.h through the experiments build's
    host hooks, CPU only) until 512*carr_phase(n) resp. code_phase(n) is m + k*2^-32 to `tol` units (k may be fractional).  That is
    where floor() of an in-tile model (c:2697 table index, c:2737 chip) can disagree with the reference and where an index
    or chip change falls (almost) exactly on a sample; k runs over `offsets` (both signs: either side of the integer, inside
    and outside the kernels' danger band).  Channel i of block b aims its carrier when (b + i) is even, its code NCO
    otherwise; with `fixed` (the 32-bit accumulator is exact) always the code.  Returns (descriptors, targets)."""
    from fractions import Fraction
    L = exp_lib()
    rng = np.random.default_rng(seed)
    ch = synth_descriptors(nblocks, nch=nch, seed=seed, max_doppler=max_doppler)
    delt = 1.0 / fs
    unit = Fraction(1, 1 << 32)
    targets = []
    wr = C.c_longlong(0)
    for b in range(nblocks):
        for i in range(nch):
            k = Fraction(float(offsets[(b * nch + i) % len(offsets)]))
            n = int(samples[(b * nch + i) % len(samples)]) if samples is not None else int(rng.integers(1, nsamp))
            n = min(n, nsamp - 1)
            carrier = (b + i) % 2 == 0 and not fixed
            if carrier:
                s = float(ch["f_carr"][b, i]) * delt            # the individually rounded product of c:2741
                want_frac = unit * k                            # 512*phase(n) = integer + k units
                cp = float(ch["carr_phase"][b, i])
                for _ in range(12):
                    got = Fraction(L.gpsbb_test_carr_jump(cp, s, n)) * 512
                    m = round(got)
                    err = (got - m) - want_frac                 # in table-index units
                    if abs(err) <= unit * Fraction(tol):
                        break
                    cp = float((Fraction(cp) - err / 512) % 1)
                ch["carr_phase"][b, i] = cp
                got = Fraction(L.gpsbb_test_carr_jump(cp, s, n)) * 512
                targets.append((b, i, "carr", n, float(k), float((got - round(got)) / unit)))
            else:
                s = float(ch["f_code"][b, i]) * delt            # c:2709
                x = float(ch["code_phase"][b, i])
                for _ in range(12):
                    got = Fraction(L.gpsbb_test_code_jump(x, s, n, C.byref(wr)))
                    m = round(got)
                    err = (got - m) - unit * k
                    if abs(err) <= unit * Fraction(tol):
                        break
                    x = float((Fraction(x) - err) % 1023)
                ch["code_phase"][b, i] = x
                got = Fraction(L.gpsbb_test_code_jump(x, s, n, C.byref(wr)))
                targets.append((b, i, "code", n, float(k), float((got - round(got)) / unit)))
    if fixed:
        ch["carr_phase"] = np.floor(ch["carr_phase"] * 2.0 ** 32)
    return ch, targets


# ---- time sharding (one process per GPU, no data-path collective) ------------------------------------

def shard_blocks(nblocks, rank, world):
    """Contiguous block range [b0, b1) of rank `rank` out of `world` (BASELINE configs[4]: GPU g gets blocks
    [g*B/G, (g+1)*B/G))."""
    return (nblocks * rank) // world, (nblocks * (rank + 1)) // world


def shard_descriptors(ch, rank, world, delt, nsamp, nthreads=0):
    """Descriptors of this rank's time shard with the exact carrier phase seeded at every block start, so
    the shard can be synthesised without the preceding blocks (gpsbb_chain_carrier_host)."""
    ch = _as_chan(ch).copy()
    ch["carr_phase"] = chain_carrier_host(ch, delt, nsamp, nthreads)
    b0, b1 = shard_blocks(ch.shape[0], rank, world)
    return ch[b0:b1]


# ---- host front end (include/gpsfe.h): RINEX-2 + position/motion -> per-block descriptors -------------

FE_LIB_PATH = os.path.join(HERE, "libgpsfe.so")
_fe_lib = None


class _FeConfig(C.Structure):
    _fields_ = [("navfile", C.c_char_p), ("rinex3", C.c_int), ("motion_file", C.c_char_p), ("use_ecef", C.c_int),
                ("pos", C.c_double * 3), ("have_start", C.c_int), ("y", C.c_int), ("m", C.c_int),
                ("d", C.c_int), ("hh", C.c_int), ("mm", C.c_int), ("sec", C.c_double),
                ("time_overwrite", C.c_int), ("iono_disable", C.c_int), ("max_chan", C.c_int),
                ("fixed_carrier", C.c_int)]


def build_frontend(force=False):
    hostdir = os.path.join(HERE, "host")
    sim = os.path.join(HERE, "gpsbb-sim")
    newest = max(os.path.getmtime(os.path.join(hostdir, f)) for f in os.listdir(hostdir))
    if force or not os.path.exists(FE_LIB_PATH) or not os.path.exists(sim) or newest > os.path.getmtime(FE_LIB_PATH):
        subprocess.check_call(["make", "-C", hostdir])
    return FE_LIB_PATH


def fe_lib():
    global _fe_lib
    if _fe_lib is None:
        if not os.path.exists(FE_LIB_PATH):
            raise RuntimeError("libgpsfe.so is not built (make -C %s/host)" % HERE)
        L = C.CDLL(FE_LIB_PATH)
        L.gpsfe_open.argtypes = [C.POINTER(_FeConfig), C.POINTER(C.c_void_p)]
        L.gpsfe_close.argtypes = [C.c_void_p]
        L.gpsfe_close.restype = None
        L.gpsfe_strerror.argtypes = [C.c_int]
        L.gpsfe_strerror.restype = C.c_char_p
        L.gpsfe_max_chan.argtypes = [C.c_void_p]
        L.gpsfe_next_block.argtypes = [C.c_void_p, C.c_void_p]
        L.gpsfe_feed_back.argtypes = [C.c_void_p, C.c_void_p]
        L.gpsfe_generate.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gpsfe_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.gpsfe_time.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.gpsfe_channel_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)] + [C.POINTER(C.c_double)] * 4
        _fe_lib = L
    return _fe_lib


class FrontEnd:
    """gpsfe_open / gpsfe_generate: the scenario the reference's main() runs, as descriptor blocks."""

    def __init__(self, navfile, llh=None, ecef=None, motion=None, start=None, time_overwrite=False,
                 iono=True, max_chan=12, rinex3=False, fixed_carrier=False):
        cfg = _FeConfig()
        cfg.navfile = os.fsencode(navfile)
        cfg.rinex3 = int(rinex3)
        cfg.motion_file = os.fsencode(motion) if motion else None
        if ecef is not None:
            cfg.use_ecef = 1
            cfg.pos = (C.c_double * 3)(*ecef)
        else:
            cfg.pos = (C.c_double * 3)(*(llh if llh is not None else (35.681298, 139.766247, 10.0)))
        if start is not None:  # (y, m, d, hh, mm, sec)
            cfg.have_start = 1
            cfg.y, cfg.m, cfg.d, cfg.hh, cfg.mm = [int(v) for v in start[:5]]
            cfg.sec = float(start[5])
        cfg.time_overwrite = int(time_overwrite)
        cfg.iono_disable = int(not iono)
        cfg.max_chan = max_chan
        cfg.fixed_carrier = int(fixed_carrier)
        self.max_chan = max_chan
        self._fe = C.c_void_p()
        rc = fe_lib().gpsfe_open(C.byref(cfg), C.byref(self._fe))
        if rc != 0:
            raise RuntimeError("gpsfe_open: %s (%d)" % (fe_lib().gpsfe_strerror(rc).decode(), rc))

    def close(self):
        if self._fe:
            fe_lib().gpsfe_close(self._fe)
            self._fe = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_threads(self, n):
        """gpsfe_set_threads: 0 = default (cores up to 16), 1 = the sequential loop"""
        rc = fe_lib().gpsfe_set_threads(self._fe, n)
        if rc != 0:
            raise RuntimeError("gpsfe_set_threads: %d" % rc)

    def generate(self, nblocks):
        ch = np.zeros((nblocks, self.max_chan), CHAN_DTYPE)
        rc = fe_lib().gpsfe_generate(self._fe, nblocks, ch.ctypes.data)
        if rc != 0:
            raise RuntimeError("gpsfe_generate: %d" % rc)
        return ch

    def next_block(self):
        ch = np.zeros(self.max_chan, CHAN_DTYPE)
        rc = fe_lib().gpsfe_next_block(self._fe, ch.ctypes.data)
        if rc != 0:
            raise RuntimeError("gpsfe_next_block: %d" % rc)
        return ch

    def feed_back(self, end_state):
        st = np.ascontiguousarray(end_state, dtype=STATE_DTYPE)
        rc = fe_lib().gpsfe_feed_back(self._fe, st.ctypes.data)
        if rc != 0:
            raise RuntimeError("gpsfe_feed_back: %d" % rc)

    def time(self):
        w, s = C.c_int(), C.c_double()
        fe_lib().gpsfe_time(self._fe, C.byref(w), C.byref(s))
        return w.value, s.value
