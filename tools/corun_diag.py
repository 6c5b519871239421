#!/usr/bin/env python3
"""What a neighbour, or an idle gap, costs the synthesis kernels — measured from INSIDE the kernels, not swept.

Needs the trace build (make -C pluto-gps-sim_amd/csrc trace VARIANT_DIR=build/variants: the experiments build with
-DGPSBB_WG_TRACE): every workgroup of k_synth_ev / k_synth_pd (and of the lap passes) leaves a record of when it entered, had
its tables staged, finished wavefront 0's tiles and left — in ticks of the 100 MHz reference counter AND in shader-clock cycles
— and of the CU it ran on.  From the records of one launch:

    kernel_us       last exit - first entry (what HIP events see, minus the launch overhead)
    mhz             shader cycles / reference ticks over the working workgroups: the clock the chip actually ran at
    cyc_per_tile    wavefront 0's cycles per 1024-sample tile, median over workgroups: what the work costs in ISSUE SLOTS,
                    whatever the clock (it rises when something else issues on the same SIMD, or when LDS / memory waits grow)
    us_per_tile     the same in wall time (= cyc_per_tile / mhz)
    cu_busy         share of (CUs used x kernel time) during which a CU held a working workgroup; the rest is split into
    cu_head / cu_gap / cu_tail   ... before a CU's first workgroup, between two of them, after its last one
    stage_us        entry -> tables staged, median
    lap_wgs         workgroups of the lap passes that ran while the launch was on the chip

Cases per kernel: back to back with the GPU to itself; the same with idle gaps between launches (the host waits, then
launches); beside the lap-parallel pre-pass (resident re-runs); beside the row walks; and, k_synth_ev, the headline's stream
of fresh pushes.    python tools/corun_diag.py [--out gpurun_out/r06_corun] [--quick]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, "build", "variants", "libgpsbb_trace.so")
os.environ.setdefault("GPSBB_PY_LIB", TRACE_LIB)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
import bench

WORDS = 12
CAP = 400000
NTILES = {1: (2500000 + 1023) // 1024, 2: (300000 + 1023) // 1024}  # tiles per block of the two geometries below


def trace_fns():
    lib = pkg.lib()
    lib.gpsbb_test_wg_trace_begin.argtypes = [C.c_uint]
    lib.gpsbb_test_wg_trace_begin.restype = C.c_int
    lib.gpsbb_test_wg_trace_read.argtypes = [C.c_void_p, C.c_uint]
    lib.gpsbb_test_wg_trace_read.restype = C.c_long
    return lib


LIB = trace_fns()


def trace_begin():
    rc = LIB.gpsbb_test_wg_trace_begin(CAP)
    if rc:
        raise SystemExit("wg_trace_begin: %d" % rc)


def trace_read():
    buf = np.zeros((CAP, WORDS), np.uint64)
    n = LIB.gpsbb_test_wg_trace_read(buf.ctypes.data, CAP)
    if n < 0:
        raise SystemExit("wg_trace_read: %d" % n)
    if n > CAP:
        sys.stderr.write("trace overflow: %d records, kept %d\n" % (n, CAP))
    return buf[:min(n, CAP)]


def analyse(rec, kind):
    """per-launch figures of the synthesis kernel `kind` (1 k_synth_ev, 2 k_synth_pd) from the records of one case"""
    k = (rec[:, 10] & 0xff).astype(np.int64)
    worked = (rec[:, 10] & 0x100) != 0
    syn = rec[k == kind]
    lap = rec[(k >= 3) & (k <= 6)]
    if not len(syn):
        return []
    order = np.argsort(syn[:, 0], kind="stable")
    syn = syn[order]
    sw = worked[k == kind][order]
    # launches: a workgroup that enters after everything before it has left starts a new one (one synthesis stream)
    ent, ext = syn[:, 0].astype(np.int64), syn[:, 3].astype(np.int64)
    cuts, hi = [0], ext[0]
    for j in range(1, len(syn)):
        if ent[j] > hi:
            cuts.append(j)
        hi = max(hi, ext[j])
    cuts.append(len(syn))
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        s, w = syn[a:b], sw[a:b]
        if w.sum() < 8:
            continue
        t0, t1 = int(s[:, 0].min()), int(s[:, 3].max())
        dur = t1 - t0
        ws = s[w]
        wall = (ws[:, 3] - ws[:, 0]).astype(np.float64)
        clk = (ws[:, 7] - ws[:, 4]).astype(np.float64)
        tiles = ((ws[:, 9] >> 40) & 0xffffff).astype(np.float64)
        has = tiles > 0
        cyc_tile = ((ws[:, 6] - ws[:, 5]).astype(np.float64)[has] / tiles[has])
        tick_tile = ((ws[:, 2] - ws[:, 1]).astype(np.float64)[has] / tiles[has])
        cu = ((ws[:, 8] >> 32) << 8 | ((ws[:, 8] >> 8) & 0xff)).astype(np.int64)
        busy = head = gap = tail = 0.0
        cus = np.unique(cu)
        for c in cus:
            m = cu == c
            e, x = ws[m, 0].astype(np.int64), ws[m, 3].astype(np.int64)
            o = np.argsort(e)
            e, x = e[o], x[o]
            # a CU holds one synthesis workgroup at a time (its LDS): intervals do not overlap
            busy += float((x - e).sum())
            head += float(e[0] - t0)
            tail += float(t1 - x[-1])
            gap += float(np.maximum(e[1:] - x[:-1], 0).sum())
        tot = float(len(cus) * dur)
        nlap = int(((lap[:, 3].astype(np.int64) > t0) & (lap[:, 0].astype(np.int64) < t1)).sum()) if len(lap) else 0
        lap_cu_time = 0.0
        if len(lap):
            lo = np.maximum(lap[:, 0].astype(np.int64), t0)
            hi2 = np.minimum(lap[:, 3].astype(np.int64), t1)
            lap_cu_time = float(np.maximum(hi2 - lo, 0).sum())
        wgd = np.sort(wall) / 100.0
        last_exit = np.sort(np.array([ws[cu == c, 3].astype(np.int64).max() for c in cus]) - t0) / 100.0
        ntiles_launch = float(int(s[0, 11]) & 0xffffffff) * NTILES[kind]
        out.append({"wg_us_p50": float(wgd[len(wgd) // 2]), "wg_us_p90": float(wgd[int(len(wgd) * 0.9)]), "wg_us_max": float(wgd[-1]),
                    "cu_drained_p50_us": float(last_exit[len(last_exit) // 2]), "cu_drained_p90_us": float(last_exit[int(len(last_exit) * 0.9)]),
                    "wave_us_per_tile": float((ws[:, 3] - ws[:, 1]).astype(np.float64).sum()) / 100.0 * 16.0 / ntiles_launch,
                    "wave_cyc_per_tile": float((ws[:, 7] - ws[:, 5]).astype(np.float64).sum()) * 16.0 / ntiles_launch,
                    "t0": t0, "kernel_us": dur / 100.0, "wgs": int(len(s)), "wgs_worked": int(w.sum()), "cus": int(len(cus)),
                    "mhz": 100.0 * clk.sum() / wall.sum(), "cyc_per_tile": float(np.median(cyc_tile)), "cyc_per_tile_p90": float(np.percentile(cyc_tile, 90)),
                    "us_per_tile": float(np.median(tick_tile)) / 100.0, "stage_us": float(np.median((ws[:, 1] - ws[:, 0]).astype(np.float64))) / 100.0,
                    "cu_busy": busy / tot, "cu_head": head / tot, "cu_gap": gap / tot, "cu_tail": tail / tot,
                    "entry_skew_us": float(np.percentile((ws[:, 0].astype(np.int64) - t0), 50)) / 100.0,
                    "lap_wgs": nlap, "lap_wg_us_per_cu": lap_cu_time / 100.0 / max(len(cus), 1)})
    # the idle time before each launch
    for j in range(1, len(out)):
        out[j]["idle_before_us"] = (out[j]["t0"] - (out[j - 1]["t0"] + out[j - 1]["kernel_us"] * 100.0)) / 100.0
    return out


RAW = {}


def keep_raw(name, rec):
    RAW[name[:40].replace(" ", "_").replace(",", "").replace("'", "")] = rec


def summarise(name, launches, ev_ms=None, skip=2):
    ls = launches[skip:] if len(launches) > skip + 2 else launches
    if not ls:
        return {"case": name, "launches": 0}
    med = lambda f: float(np.median([x[f] for x in ls if f in x])) if any(f in x for x in ls) else None
    r = {"case": name, "launches": len(ls), "event_ms": ev_ms}
    for f in ("wg_us_p50", "wg_us_p90", "wg_us_max", "cu_drained_p50_us", "cu_drained_p90_us", "wave_us_per_tile", "wave_cyc_per_tile", "kernel_us", "mhz", "cyc_per_tile", "cyc_per_tile_p90", "us_per_tile", "stage_us", "cu_busy", "cu_head", "cu_gap", "cu_tail", "cus",
              "wgs_worked", "lap_wgs", "lap_wg_us_per_cu", "idle_before_us"):
        r[f] = med(f)
    return r


def busy_wait(seconds):
    t = time.perf_counter() + seconds
    while time.perf_counter() < t:
        pass


def run_batch_case(s, batch, out, steps, gap_s=None):
    batch.timing_stats(reset=True)
    trace_begin()
    for _ in range(steps):
        batch.run(out.data_ptr())
        if gap_s is not None:
            s.sync()
            busy_wait(gap_s)
    s.sync()
    torch.cuda.synchronize()
    st = batch.timing_stats(reset=True)
    return trace_read(), st["ms_synth_sum"] / max(st["runs"], 1)


def kernel_cases(s, ch, delt, nsamp, flags, kind, quick):
    res = []
    out = torch.empty(ch.shape[0] * nsamp * 2, dtype=torch.int16, device="cuda:0")
    steps = 8 if quick else 16
    for where, name in ((3, "beside the lap passes"), (1, "beside the row walks")):
        s.set_option(pkg.OPT_SEED_WHERE, where)
        batch = s.batch(ch, delt, nsamp, flags=flags)
        for _ in range(4):
            batch.run(out.data_ptr())
        s.sync()
        rec, ms = run_batch_case(s, batch, out, steps)
        keep_raw("k%d %s" % (kind, name), rec)
        res.append(summarise(name + " (resident re-runs)", analyse(rec, kind), ms))
        if where == 3:
            # the same tables, the pre-pass switched off: the kernel with the GPU to itself
            s.set_option(pkg.OPT_SKIP_SEED, 1)
            for _ in range(3):
                batch.run(out.data_ptr())
            s.sync()
            rec, ms = run_batch_case(s, batch, out, steps)
            keep_raw("k%d alone" % kind, rec)
            res.append(summarise("alone, back to back", analyse(rec, kind), ms))
            for gap in ((50e-6, 300e-6, 2e-3, 20e-3) if not quick else (300e-6, 20e-3)):
                rec, ms = run_batch_case(s, batch, out, max(6, steps // 2), gap_s=gap)
                res.append(summarise("alone, host idles %g us between launches" % (gap * 1e6), analyse(rec, kind), ms, skip=1))
            s.set_option(pkg.OPT_SKIP_SEED, 0)
        batch.close()
    s.set_option(pkg.OPT_SEED_WHERE, 0)
    del out
    return res


def stream_case(s, nch, delt, nsamp, PB, npush, depth, kind):
    mine = bench.stream_descriptors(pkg, npush * PB, nch)
    st = s.stream(nch, delt, nsamp, PB, depth=depth, flags=pkg.CHAIN_CARRIER | pkg.STREAM_DEVICE_ONLY)

    def ring(first, count):
        for j in range(count):
            if st.pending >= depth:
                st.pop(copy=False)
            k = (first + j) % npush
            st.push(mine[k * PB:(k + 1) * PB])
        while st.pending:
            st.pop(copy=False)
    ring(0, depth + 2)
    s.sync()
    st.timing_stats(reset=True)
    trace_begin()
    t0 = time.perf_counter()
    ring(depth + 2, npush)
    s.sync()
    dt = time.perf_counter() - t0
    stats = st.timing_stats(reset=True)
    rec = trace_read()
    keep_raw("k%d stream" % kind, rec)
    st.close()
    r = summarise("the headline's stream of fresh chained pushes", analyse(rec, kind), stats["ms_synth_sum"] / max(stats["runs"], 1))
    r["value"] = npush * PB * nsamp / dt
    r["prepass_ms"] = stats["ms_seed_sum"] / max(stats["runs"], 1)
    return r


def show(title, rows):
    print("== " + title)
    print("%-58s %8s %8s %7s %9s %8s %6s %6s %6s %6s %7s %8s | %7s %7s %7s %8s %8s" % ("case", "event_ms", "kern_us", "MHz", "cyc/tile", "us/tile", "busy", "head", "gap", "tail", "lapWGs", "idle_us",
                                                                                  "wg_p50", "wg_p90", "wg_max", "drain50", "drain90"))
    for r in rows:
        if not r.get("launches"):
            print("%-58s (no launches traced)" % r["case"])
            continue
        f = lambda v, p: ("%" + p) % v if v is not None else "-"
        print("%-58s %8s %8.1f %7.1f %9.1f %8.3f %6.3f %6.3f %6.3f %6.3f %7s %8s | %7.1f %7.1f %7.1f %8.1f %8.1f" % (
            r["case"][:58], f(r["event_ms"], ".4f"), r["kernel_us"], r["mhz"], r["wave_cyc_per_tile"], r["wave_us_per_tile"], r["cu_busy"], r["cu_head"], r["cu_gap"],
            r["cu_tail"], f(r["lap_wgs"], ".0f"), f(r["idle_before_us"], ".1f"), r["wg_us_p50"], r["wg_us_p90"], r["wg_us_max"], r["cu_drained_p50_us"], r["cu_drained_p90_us"]))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_corun"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", choices=("ev", "pd"), default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if not os.path.exists(os.environ["GPSBB_PY_LIB"]):
        raise SystemExit("build the trace variant first: make -C pluto-gps-sim_amd/csrc trace VARIANT_DIR=%s" % os.path.dirname(TRACE_LIB))
    result = {}
    with pkg.Synth(0) as s:
        if a.only in (None, "ev"):
            ch = bench.stream_descriptors(pkg, 400, 16)
            rows = kernel_cases(s, ch, 1.0 / 25e6, 2500000, pkg.CHAIN_CARRIER, 1, a.quick)
            rows.append(stream_case(s, 16, 1.0 / 25e6, 2500000, 400, 16 if a.quick else 32, 6, 1))
            result["k_synth_ev: 16 ch, 25 MS/s, 400 blocks of 2.5e6 samples per launch"] = rows
            show("k_synth_ev (1e9 samples per launch)", rows)
        if a.only in (None, "pd"):
            mch = pkg.synth_descriptors(1000, nch=12, seed=0xF00D)
            rows = kernel_cases(s, mch, 1.0 / 2.6e6, 300000, 0, 2, a.quick)
            result["k_synth_pd: 12 ch, 2.6 MS/s, 1000 blocks of 300000 samples per launch"] = rows
            show("k_synth_pd (3e8 samples per launch)", rows)
    json.dump(result, open(os.path.join(a.out, "corun_diag.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(a.out, "corun_raw.npz"), **RAW)


if __name__ == "__main__":
    main()
