#!/bin/bash
# One GPU-box session: parity tests, the bench with its defaults, a rocprofv3 kernel trace of the SAME command, and the
# HBM write / read counters of the headline leg (counters in passes of their own, kernel-trace only).
# Usage (from the repo root, via gpurun):  bash tools/gpu_round.sh <tag>
set -u
TAG="${1:-r02}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -4 "$OUT/pytest_gpu.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 2500 "$OUT/bench.json"
# per-kernel time of the TIMED LEG ALONE (--no-extras: no parity / cross-check / node-driver legs, whose launches run beside other
# kernels and used to be averaged in): avg(k_synth_ev) x 8 pushes <= ms_per_step of the same run must hold
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof" -o trace -- python "$ROOT/bench.py" --no-extras --no-cpu > "$ROOT/$OUT/prof_stdout.log" 2>&1 )
grep -o "^{\"metric\".*" "$OUT/prof_stdout.log" | tail -1 | cut -c1-400 > "$OUT/prof_bench_line.json"
# HBM bytes of the headline leg: separate --pmc passes
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$ROOT/$OUT/pmc_w" -o pmc -- python "$ROOT/bench.py" --no-extras --steps 4 --repeats 1 --warmup 1 > "$ROOT/$OUT/pmc_w_stdout.log" 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$ROOT/$OUT/pmc_r" -o pmc -- python "$ROOT/bench.py" --no-extras --steps 4 --repeats 1 --warmup 1 > "$ROOT/$OUT/pmc_r_stdout.log" 2>&1 )
# WRITE_SIZE calibration on a kernel that writes a known byte count (k_fill_ceiling runs in the resident leg of the full bench)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$ROOT/$OUT/pmc_w" -o pmc_cal -- python "$ROOT/tools/kbench.py" --steps 2 --warmup 4 --fill-ceiling > "$ROOT/$OUT/pmc_cal_stdout.log" 2>&1 )
# FETCH_SIZE calibration on a kernel that reads a known byte count with the tile states' own pattern (experiments build)
( cd /tmp && GPSBB_PY_LIB=exp timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$ROOT/$OUT/pmc_r" -o pmc_cal -- python "$ROOT/tools/kbench.py" --steps 2 --warmup 4 --read-cal > "$ROOT/$OUT/pmc_rcal_stdout.log" 2>&1 )
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"; head -c 3000 "$OUT/pmc_summary.json"
# the reference's own geometry (M1: 12 ch, 2.6 MS/s, 300 000-sample blocks, k_synth_pd): kernel stats and SQ counters
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof_m1" -o trace -- python "$ROOT/tools/m1_rate.py" > "$ROOT/$OUT/prof_m1_stdout.log" 2>&1 )
timeout 600 bash tools/pmc_sq.sh ${TAG}_m1_sq --fs 2.6e6 --nsamp 300000 --nch 12 --blocks 333 > gpurun_out/${TAG}_m1_sq.txt 2>&1
# SQ counters of ONE 400-block launch of the headline geometry (1e9 samples: what a push of the bench is)
timeout 900 bash tools/pmc_sq.sh ${TAG}_sq --blocks 400 --chain --smooth > gpurun_out/${TAG}_sq.txt 2>&1
timeout 200 python tools/seed_rate.py --host > "$OUT/seed_rate.txt" 2>&1
# what the co-run costs, from inside the kernels (trace build), and the package power beside it
timeout 900 bash tools/corun_session.sh ${TAG}_corun nopmc > "$OUT/corun.log" 2>&1
timeout 300 bash tools/power_probe.sh ${TAG}_power > "$OUT/power.log" 2>&1
# the drop-in call on one time axis: copied into a pageable iq_buff, and rendered into a registered one
timeout 300 bash tools/fill_timeline.sh ${TAG}_fill_tl > "$OUT/fill_timeline.txt" 2>&1
FILL_REG=1 timeout 300 bash tools/fill_timeline.sh ${TAG}_fill_tl_reg > "$OUT/fill_timeline_registered.txt" 2>&1
