#!/bin/bash
set -u
run() { timeout 200 python tools/kbench.py --smooth --chain --synth-only --steps 800 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  synth ms/launch %.3f' % d['roofline']['ms_per_launch'])"; }
echo alone; run
echo "beside gpsbb_chain_carrier (walk<1>, prefix, walk<3>, fix_par; no rows, no tiles)"
timeout 120 python tools/corun_prepass.py chain 60 > gpurun_out/corun_chain.log 2>&1 &
P=$!
for i in $(seq 1 40); do grep -q ready gpurun_out/corun_chain.log 2>/dev/null && break; sleep 1; done
run; run
kill $P 2>/dev/null; wait $P 2>/dev/null
tail -2 gpurun_out/corun_chain.log
