#!/usr/bin/env python3
"""Summarise one GPU round's rocprofv3 output (ROCm 7.2 writes rocpd SQLite .db files) into text/JSON
that can be committed under profiles/.

  python tools/pmc_summary.py gpurun_out/<tag> [profiles/<name-prefix>]

Reads <tag>/prof/*.db (--kernel-trace --stats run), <tag>/pmc_w/*.db (--pmc WRITE_SIZE run) and
<tag>/pmc_r/*.db (--pmc FETCH_SIZE run).  HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md
("HBM"): counters collected in their own passes; WRITE_SIZE / FETCH_SIZE are in KiB (x1024);
WRITE_SIZE is calibrated on k_fill_ceiling, which writes a known byte count; FETCH_SIZE on k_read_pattern, which reads a known byte
count with the synthesis kernel's own pattern for the tile states (8 bytes per lane, a row per lane: tools/kbench.py --read-cal) —
the guide's factor 2 for gfx950 is for wide coalesced reads (16 bytes per lane) and does not hold for this pattern (round 6:
measured, see "fetch_size_calibration"); without a calibration pass the factor 2 is applied and said so.
rocprofv3 --pmc SERIALISES kernels (round 6: none of the lap-pass launches of a counter pass overlaps a k_synth_ev launch in its
own kernel trace, tools/corun_session.sh), so a kernel's counters are its own.
"""
import glob
import json
import os
import sqlite3
import sys


def q(db, sql):
    c = sqlite3.connect(db)
    try:
        return list(c.execute(sql))
    finally:
        c.close()


KERNELS = ("k_synth_ev", "k_synth_ev_dense", "k_synth_ev_digest", "k_synth_pd<true>", "k_synth_pd<false>", "k_synth_pd<true, false>", "k_synth_pd<false, false>",
           "k_synth_pd<true, true>", "k_synth_pd<false, true>", "k_synth", "k_walk<0>", "k_walk<1>", "k_walk<2>", "k_walk<3>", "k_tiles",
           "k_chain_fix", "k_chain_fix_par<128>", "k_chain_fix_par<256>", "k_chain_prefix", "k_seed<true>", "k_seed<false>", "k_fill_ceiling",
           "k_gather_to_host", "k_end_states_to_host", "k_read_pattern", "k_lap_pass1<0>", "k_lap_pass1<1>", "k_lap_pass2<0>", "k_lap_pass2<1>", "k_lap_pass2<0, true>", "k_lap_pass2<1, true>", "k_lap_pass2<0, false>", "k_lap_pass2<1, false>", "k_lap_pass2_2<true>", "k_lap_pass2_2<false>",
           "k_lap_plan<0>", "k_lap_plan<1>", "k_lap_scan<0>", "k_lap_scan<1>", "k_lap_repair<0>", "k_lap_repair<1>", "k_block_digest")


def short(name):
    """the kernel's own name, exactly (rocprofv3 prints "[void ]gpsbb_impl::<name>(<args>)"): k_synth_ev_dense is not k_synth_ev"""
    n = name.split("(")[0].split("::")[-1].strip()
    return n if n in KERNELS else None


def main():
    tag = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else None
    res = {"kernels": {}}
    lines = []
    for db in glob.glob(os.path.join(tag, "prof", "*.db")):
        rows = q(db, "select name,total_calls,total_duration,average,percentage from top_kernels")
        lines.append("%-90s %6s %14s %12s %7s" % ("kernel (rocprofv3 --kernel-trace --stats)", "calls", "total_us", "avg_us", "pct"))
        for n, calls, tot, avg, pct in rows:
            lines.append("%-90s %6d %14.3f %12.3f %7.2f" % (n[:90], calls, tot, avg, pct))
            if short(n):
                res["kernels"].setdefault(short(n), {}).update(calls=calls, avg_us=avg, pct=pct)
        for n, vg, sg, lds, gx, gy, wx in q(db, "select name,vgpr_count,sgpr_count,lds_size,grid_x,grid_y,workgroup_x from kernels group by name"):
            if short(n):
                res["kernels"][short(n)].update(vgpr=vg, sgpr=sg, lds=lds, grid=[gx, gy], wg=wx)
    for sub, ctr in (("pmc_w", "WRITE_SIZE"), ("pmc_r", "FETCH_SIZE")):
        for db in glob.glob(os.path.join(tag, sub, "*.db")):
            for n, avg, cnt in q(db, "select kernel_name, avg(value), count(*) from counters_collection "
                                     "where counter_name='%s' group by kernel_name" % ctr):
                if short(n):
                    res["kernels"].setdefault(short(n), {})[ctr + "_KiB_per_launch"] = avg
                    res["kernels"][short(n)]["launches_" + ctr] = cnt
    k = res["kernels"]
    bench = None
    bj = os.path.join(tag, "bench.json")
    if os.path.exists(bj):
        try:
            bench = json.loads(open(bj).read().strip().splitlines()[-1])
        except Exception:
            bench = None
    if "k_fill_ceiling" in k and "WRITE_SIZE_KiB_per_launch" in k["k_fill_ceiling"] and bench:
        known = bench["roofline"]["algorithmic_bytes_per_launch"]  # the fill writes the same buffer
        res["write_size_calibration"] = k["k_fill_ceiling"]["WRITE_SIZE_KiB_per_launch"] * 1024.0 / known
    dom = "k_synth_ev" if "k_synth_ev" in k else "k_synth"   # the dominant kernel of the bench workload
    res["dominant_kernel"] = dom
    if dom in k and "WRITE_SIZE_KiB_per_launch" in k[dom]:
        cal = res.get("write_size_calibration", 1.0) or 1.0
        w = k[dom]["WRITE_SIZE_KiB_per_launch"] * 1024.0 / cal
        rcal = None
        if "k_read_pattern" in k and "FETCH_SIZE_KiB_per_launch" in k["k_read_pattern"] and bench:
            per_wave = 32 * 2442 * 8
            waves = bench["roofline"]["algorithmic_bytes_per_launch"] // per_wave
            waves -= waves % 4
            rcal = k["k_read_pattern"]["FETCH_SIZE_KiB_per_launch"] * 1024.0 / (waves * per_wave)
            res["fetch_size_calibration"] = rcal
            res["fetch_size_calibration_note"] = "FETCH_SIZE of k_read_pattern / the bytes it reads (the tile states' pattern)"
        if rcal:
            r = k[dom].get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024.0 / rcal
        else:
            r = 2.0 * k[dom].get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024.0
            res["fetch_size_calibration_note"] = "no calibration pass: the guide's factor 2 for wide coalesced reads applied"
        res["k_synth_hbm_write_bytes_per_launch"] = w
        res["k_synth_hbm_read_bytes_per_launch"] = r
        res["k_synth_hbm_bytes_per_launch"] = w + r
    if bench:
        res["bench"] = {kk: bench[kk] for kk in ("value", "ms_per_step", "steps", "warmup", "repeats", "roofline",
                                                 "prepass_ms_per_launch", "config") if kk in bench}
    print(json.dumps(res, indent=1))
    if prefix:
        with open(prefix + "_kernel_stats.txt", "w") as f:
            f.write("\n".join(lines) + "\n")
        with open(prefix + "_pmc.json", "w") as f:
            json.dump(res, f, indent=1)
        with open(os.path.join(os.path.dirname(prefix), "pmc_latest.json"), "w") as f:
            json.dump({kk: res[kk] for kk in res if kk.startswith("k_synth_hbm")}, f, indent=1)


if __name__ == "__main__":
    main()
