#!/usr/bin/env python3
"""Kernel-tuning helper (not the bench contract): one batch resident in HBM, run again and again.
   python tools/kbench.py [--blocks 400] [--steps 10] [--warmup 4] [--chain] [--synth-only] [--per-sample]
Prints one JSON line with value (IQ samples/s), ms_per_step, roofline.ms_per_launch (synthesis kernel, HIP events)
and seed_kernel_ms_per_launch (pre-pass, HIP events)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--blocks", type=int, default=400)
    ap.add_argument("--nch", type=int, default=16)
    ap.add_argument("--fs", type=float, default=25e6)
    ap.add_argument("--nsamp", type=int, default=2500000)
    ap.add_argument("--chain", action="store_true")
    ap.add_argument("--synth-only", action="store_true", help="after warm-up re-run only the synthesis kernel on the tables already built")
    ap.add_argument("--per-sample", action="store_true", help="force the per-sample kernel k_synth")
    ap.add_argument("--smooth", action="store_true", help="bench.py's stream descriptors (slowly varying Doppler) instead of M2's")
    ap.add_argument("--no-cpu", action="store_true", help="ignored (compatibility with older scripts)")
    ap.add_argument("--where", type=int, default=0, help="GPSBB_OPT_SEED_WHERE: 1 the row walks of rounds 1-4, 3 the lap-parallel pre-pass")
    ap.add_argument("--fill-ceiling", action="store_true", help="also run the pure write kernel over the output buffer (counter calibration)")
    ap.add_argument("--read-cal", action="store_true", help="also read the output buffer once with the tile states' access pattern (k_read_pattern, experiments build: FETCH_SIZE calibration); prints the bytes read")
    a = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    if a.smooth:
        import bench
        ch = bench.stream_descriptors(pkg, a.blocks, a.nch)
    else:
        ch = pkg.synth_descriptors(a.blocks, nch=a.nch, seed=0x5EED)
    synth = pkg.Synth(0)
    if a.per_sample:
        synth.set_option(pkg.OPT_SYNTH_KERNEL, 1)
    if a.where:
        synth.set_option(pkg.OPT_SEED_WHERE, a.where)
    batch = synth.batch(ch, 1.0 / a.fs, a.nsamp, flags=pkg.CHAIN_CARRIER if a.chain else 0)
    out = torch.empty(a.blocks * a.nsamp * 2, dtype=torch.int16, device="cuda:0")
    if a.synth_only:
        synth.set_option(pkg.OPT_SKIP_SEED, 1)
        a.warmup = max(a.warmup, 5)
    for _ in range(a.warmup):
        batch.run(out.data_ptr())
    synth.sync()
    torch.cuda.synchronize()
    batch.timing_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        batch.run(out.data_ptr())
    synth.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = batch.timing_stats(reset=True)
    n = a.blocks * a.nsamp
    if a.fill_ceiling:
        synth.fill_ceiling(out.data_ptr(), out.numel() * 2, iters=10)
    read_cal_bytes = None
    if a.read_cal:
        import ctypes as C
        L = pkg.lib()
        L.gpsbb_test_read_pattern.restype = C.c_longlong
        L.gpsbb_test_read_pattern.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for _ in range(3):
            read_cal_bytes = L.gpsbb_test_read_pattern(synth._h, C.c_void_p(out.data_ptr()), out.numel() * 2)
    print(json.dumps({"value": n * a.steps / dt, "ms_per_step": dt / a.steps * 1e3,
                      "roofline": {"ms_per_launch": st["ms_synth_sum"] / max(st["runs"], 1)},
                      "seed_kernel_ms_per_launch": st["ms_seed_sum"] / max(st["runs"], 1),
                      "kernel": synth.info(pkg.INFO_LAST_KERNEL), "chain_on_device": synth.info(pkg.INFO_CHAIN_ON_DEVICE),
                      "chain_fallbacks": synth.info(pkg.INFO_CHAIN_FALLBACKS), "chain_repairs": synth.info(pkg.INFO_CHAIN_REPAIRS),
                      "launches": a.steps + a.warmup, "read_cal_bytes": read_cal_bytes}))
    synth.set_option(pkg.OPT_SKIP_SEED, 0)
    batch.close()
    synth.close()


if __name__ == "__main__":
    main()
