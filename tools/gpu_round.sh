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
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -4 "$OUT/pytest_gpu.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 2500 "$OUT/bench.json"
# per-kernel time: the same command (defaults), CPU legs skipped
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof" -o trace -- python "$ROOT/bench.py" --no-cpu > "$ROOT/$OUT/prof_stdout.log" 2>&1 )
# HBM bytes of the headline leg: separate --pmc passes
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$ROOT/$OUT/pmc_w" -o pmc -- python "$ROOT/bench.py" --no-extras --steps 4 --repeats 1 --warmup 1 > "$ROOT/$OUT/pmc_w_stdout.log" 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$ROOT/$OUT/pmc_r" -o pmc -- python "$ROOT/bench.py" --no-extras --steps 4 --repeats 1 --warmup 1 > "$ROOT/$OUT/pmc_r_stdout.log" 2>&1 )
# WRITE_SIZE calibration on a kernel that writes a known byte count (k_fill_ceiling runs in the resident leg of the full bench)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$ROOT/$OUT/pmc_w" -o pmc_cal -- python "$ROOT/tools/kbench.py" --steps 2 --warmup 4 --fill-ceiling > "$ROOT/$OUT/pmc_cal_stdout.log" 2>&1 )
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"; head -c 3000 "$OUT/pmc_summary.json"
