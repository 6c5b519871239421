#!/bin/bash
# usage: variant.sh "<EXTRA flags>" ...   : pipelined bench (k_synth ms, step ms) per build variant; run from repo root on the GPU box
for m in "$@"; do
  make -C pluto-gps-sim_amd/csrc EXTRA="$m" -B >/dev/null 2>&1
  r=$(timeout 200 python tools/kbench.py --no-cpu --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['ms_per_launch'], d['ms_per_step'], d['seed_kernel_ms_per_launch'])")
  r2=$(timeout 200 python tools/kbench.py --no-cpu --steps 8 --synth-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['ms_per_launch'])")
  echo "[$m]  pipelined: k_synth ms, step ms, seed ms: $r | synth alone: $r2"
done
make -C pluto-gps-sim_amd/csrc -B >/dev/null 2>&1
