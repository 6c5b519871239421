#!/bin/bash
# The synthesis kernel alone and beside synthetic neighbours (tools/ubench/corunner.hip): what co-residency costs it.
set -u
run() { timeout 200 python tools/kbench.py --smooth --chain --synth-only --steps 400 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  synth ms/launch %.3f' % d['roofline']['ms_per_launch'])"; }
echo "alone"; run
for mode in 0 1 2; do
  for waves in 512 1024; do
    echo "corunner mode $mode waves $waves"
    timeout 60 tools/ubench/corunner $mode 25 $waves > /dev/null 2>&1 &
    P=$!
    sleep 1
    run
    kill $P 2>/dev/null; wait $P 2>/dev/null
  done
done
