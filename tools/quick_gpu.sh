#!/bin/bash
# quick GPU check: parity tests, then bench (pipelined and synth-only).  bash tools/quick_gpu.sh [pytest|nopytest]
set -u
if [ "${1:-pytest}" = "pytest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
fi
P='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], "value %.4g  ms/step %.3f  synth %.3f  seed %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["seed_kernel_ms_per_launch"]))'
timeout 300 python tools/kbench.py --steps 10 --warmup 3 --no-cpu 2>&1 | python -c "$P" pipelined
timeout 300 python tools/kbench.py --steps 10 --warmup 3 --no-cpu --synth-only 2>&1 | python -c "$P" synth-only
