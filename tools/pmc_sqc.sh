#!/bin/bash
# instruction- and scalar-cache counters of the synthesis kernel, alone (a resident batch re-run, pre-pass skipped) and in the
# bench's stream (pre-passes of the next pushes beside it): is what the neighbours cost it cache misses?
#   bash tools/pmc_sqc.sh <tag>
TAG="${1:-sqc}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$PWD
C="SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$OUT/alone" -o pmc -- python "$ROOT/tools/kbench.py" --blocks 400 --chain --smooth --synth-only --steps 6 --warmup 2 > "$OUT/alone.log" 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C -d "$OUT/leg" -o pmc -- python "$ROOT/bench.py" --no-extras --steps 4 --repeats 1 --warmup 1 > "$OUT/leg.log" 2>&1 )
python - "$OUT" <<'PY'
import sqlite3, glob, sys, os
for sub in ("alone", "leg"):
    for db in glob.glob(os.path.join(sys.argv[1], sub, "*.db")):
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        for k, n, v, cnt in rows:
            if "k_synth_ev" in k:
                print("%-6s %-12s %-22s %16.1f  (n=%d)" % (sub, "k_synth_ev", n, v, cnt))
PY
