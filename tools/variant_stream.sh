#!/bin/bash
# usage: variant_stream.sh "<EXTRA flags>" ...  : bench.py's headline leg (fresh chained pushes) per build variant; on the GPU box
for m in "$@"; do
  make -C pluto-gps-sim_amd/csrc EXTRA="$m" -B >/dev/null 2>&1
  r=$(timeout 300 python bench.py --no-extras --steps 10 --repeats 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g S/s, %.2f ms/step, synth %.3f ms, prepass %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['prepass_ms_per_launch']))")
  echo "[$m]  $r"
done
make -C pluto-gps-sim_amd/csrc -B >/dev/null 2>&1
