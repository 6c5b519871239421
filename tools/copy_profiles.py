#!/usr/bin/env python3
"""copy the summaries of a tools/gpu_round.sh run from gpurun_out/<tag> into the tracked profiles/ : python tools/copy_profiles.py <tag>"""
import json, os, shutil, sqlite3, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
c = sqlite3.connect(os.path.join(src, "prof", "trace_results.db"))
cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
rows = c.execute("select * from top_kernels").fetchall()
with open(os.path.join(dst, tag + "_stream_kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --no-cpu   (the timed leg alone; bench.py's default steps / repeats)\n")
    try:
        f.write("# the bench line of this very run: " + open(os.path.join(src, "prof_bench_line.json")).read().strip()[:330] + "\n")
    except OSError:
        pass
    f.write("# top_kernels view of the trace database; total_duration and average in microseconds\n")
    f.write(" | ".join(cols) + "\n")
    for r in rows:
        f.write(" | ".join(("%.1f" % x if isinstance(x, float) else str(x)) for x in r) + "\n")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_stream_bench.json"))
shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, tag + "_stream_pmc.json"))
sq = os.path.join(ROOT, "gpurun_out", tag + "_sq.txt")
if os.path.exists(sq):
    shutil.copy(sq, os.path.join(dst, tag + "_sq_counters.txt"))
if os.path.exists(sq):
    # the numbers bench.py's roofline_valu is computed from: vector wave-instructions per sample and the share of bank conflicts,
    # from the SQ passes over ONE 400-block launch (1e9 samples) of the headline geometry
    vals = {}
    for line in open(sq):
        f = line.split()
        if len(f) >= 3 and f[0] == "k_synth_ev":
            vals[f[1]] = float(f[2])
    if "SQ_INSTS_VALU" in vals:
        old = json.load(open(os.path.join(dst, "sq_latest.json")))
        old["k_synth_ev_valu_wave_insts_per_sample"] = vals["SQ_INSTS_VALU"] / (400 * 2500000.0)
        if vals.get("SQ_ACTIVE_INST_LDS"):
            old["lds_bank_conflict_share"] = vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / vals["SQ_ACTIVE_INST_LDS"]
        old["source"] = ("profiles/%s_sq_counters.txt (SQ_INSTS_VALU, SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS of k_synth_ev over ONE 400-block "
                         "launch: tools/kbench.py --blocks 400 --chain --smooth, 1e9 samples), profiles/r03_valu_rates_ubench.txt (4.3 cycles: the "
                         "kernel's mix of f64 / 32-bit integer / v_and), clock 2.38-2.41 GHz measured under load" % tag)
        json.dump(old, open(os.path.join(dst, "sq_latest.json"), "w"), indent=1)
for extra, name in ((tag + "_corun/corun_diag.txt", tag + "_corun_diag.txt"), (tag + "_corun/corun_diag.json", tag + "_corun_diag.json")):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", extra)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", extra), os.path.join(dst, name))
sq = os.path.join(ROOT, "gpurun_out", tag + "_m1_sq.txt")
if os.path.exists(sq):
    shutil.copy(sq, os.path.join(dst, tag + "_m1_sq_counters.txt"))
with_tl = [(n, lab) for n, lab in (("fill_timeline.txt", "copied into a pageable iq_buff"), ("fill_timeline_registered.txt", "rendered into a registered iq_buff (gpsbb_host_register)"))
           if os.path.exists(os.path.join(src, n))]
if with_tl:
    with open(os.path.join(dst, tag + "_fill_timeline.txt"), "w") as f:
        f.write("# tools/fill_timeline.sh: rocprofv3 --kernel-trace --memory-copy-trace -- python tools/fill_timeline.py\n"
                "# 40 gpsbb_fill_block_ref calls of the reference's block (12 ch, 2.6 MS/s, 300 000 samples); one call from the middle of the run,\n"
                "# its kernels and copies on one axis (us; the tracer adds ~15 us to a call)\n")
        for n, lab in with_tl:
            f.write("== " + lab + "\n" + open(os.path.join(src, n)).read())
if os.path.exists(os.path.join(src, "seed_rate.txt")):
    shutil.copy(os.path.join(src, "seed_rate.txt"), os.path.join(dst, tag + "_shard_seed_rate.txt"))
m1db = os.path.join(src, "prof_m1", "trace_results.db")
if os.path.exists(m1db):
    c1 = sqlite3.connect(m1db)
    with open(os.path.join(dst, tag + "_m1_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python tools/m1_rate.py   (12 ch, 2.6 MS/s, 300 000-sample blocks, 1000 blocks per run)\n")
        f.write(" | ".join(cols) + "\n")
        for r in c1.execute("select * from top_kernels").fetchall():
            f.write(" | ".join(("%.1f" % x if isinstance(x, float) else str(x)) for x in r) + "\n")
s = json.load(open(os.path.join(src, "pmc_summary.json")))
d = {k: s[k] for k in ("k_synth_hbm_write_bytes_per_launch", "k_synth_hbm_read_bytes_per_launch", "k_synth_hbm_bytes_per_launch")}
d["kernel"] = "k_synth_ev"
d["source"] = ("profiles/%s_stream_pmc.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes — counter passes serialise kernels: the figures are "
               "k_synth_ev's own —, KiB -> bytes, WRITE_SIZE calibrated on k_fill_ceiling (%.3f), FETCH_SIZE on k_read_pattern (%s))" %
               (tag, s.get("write_size_calibration", 1.0) or 1.0, ("%.3f" % s["fetch_size_calibration"]) if s.get("fetch_size_calibration") else "no pass: x2"))
json.dump(d, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
b = json.load(open(os.path.join(src, "bench.json")))
print("value %.4g  ms/step %.2f  seconds %s" % (b["value"], b["ms_per_step"], b["repeats"]["seconds"]))
print("roofline", json.dumps(b["roofline"])[:700])
for k in ("gather", "resident", "m1", "cpu_baseline", "prepass_ms_per_launch", "device_chain", "fill_block"):
    print(k, json.dumps(b.get(k))[:900 if k == "fill_block" else 420])
for k, v in s["kernels"].items():
    print(k, v.get("calls"), v.get("avg_us"), v.get("WRITE_SIZE_KiB_per_launch"), v.get("FETCH_SIZE_KiB_per_launch"))
