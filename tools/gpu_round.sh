#!/bin/bash
# One GPU-box session: parity tests, bench, rocprofv3 kernel trace + HBM write counters.
# Usage (from the repo root, via gpurun):  bash tools/gpu_round.sh <tag>
set -u
TAG="${1:-r01}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 3000 "$OUT/bench.json"
# per-kernel time (same command as the bench, fewer steps, no CPU leg)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o trace -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu > "$OLDPWD/$OUT/prof_stdout.log" 2>&1 )
find "$OUT/prof" -name "*kernel_stats*" | head -3
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f"
# HBM write bytes: counters in their own pass, kernel-trace only
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OLDPWD/$OUT/pmc_w" -o pmc -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu > "$OLDPWD/$OUT/pmc_w_stdout.log" 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/$OUT/pmc_r" -o pmc -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu > "$OLDPWD/$OUT/pmc_r_stdout.log" 2>&1 )
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"; cat "$OUT/pmc_summary.json"
ls -R "$OUT" | head -50
