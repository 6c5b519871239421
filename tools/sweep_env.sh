#!/bin/bash
# one short bench.py run (experiments build) per set of knobs, same box, same session: what a knob does to the stream and to
# the 2.6 MS/s leg
#   bash tools/sweep_env.sh <tag> "A=1,B=2" "C=3" ...      ("-" = no knob)
TAG="$1"; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
k=0
for cfg in "$@"; do
  k=$((k+1))
  envs=""; [ "$cfg" != "-" ] && envs=$(echo "$cfg" | tr ',' ' ')
  env GPSBB_PY_LIB=exp $envs timeout 900 python bench.py --steps 20 --repeats 3 --cpu-budget 0.3 --parity-blocks 2 --parity-spots 2 > $OUT/bench_$k.json 2> $OUT/bench_$k.err
  echo "== $cfg: rc $?"
  python tools/bench_brief.py $OUT/bench_$k.json 2>&1 | head -2
done
