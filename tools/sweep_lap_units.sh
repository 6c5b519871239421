#!/bin/bash
# laps per lane (GPSBB_LAP_UNIT_CARR / _CODE) and the bursts of plain steps (GPSBB_LAP_NO_BURST), by what they do to the bench's
# stream and to the 2.6 MS/s leg: one short bench.py run each (experiments build), same box, same session
#   bash tools/sweep_lap_units.sh <tag> "<carr>,<code>[,noburst] ..."
TAG="${1:-lapunits}"; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in ${1:-"4,2 4,2,noburst 8,4 16,4"}; do
  IFS=, read uc ud nb <<< "$cfg"
  env GPSBB_PY_LIB=exp GPSBB_LAP_UNIT_CARR=$uc GPSBB_LAP_UNIT_CODE=$ud ${nb:+GPSBB_LAP_NO_BURST=1} \
    timeout 900 python bench.py --steps 20 --repeats 3 --cpu-budget 0.3 --parity-blocks 2 --parity-spots 2 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "== carrier laps per lane $uc, code $ud ${nb:+(no bursts)}: rc $?"
  python tools/bench_brief.py $OUT/bench_$cfg.json 2>&1 | head -2
done
