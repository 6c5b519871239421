#!/bin/bash
# What the chip clocks to: idle, under the synthesis kernel alone, and under the stream leg (pre-passes beside it).
#   bash tools/clock_under_load.sh <tag>     (needs tools/ubench/clock_probe built)
set -u
TAG="${1:-clock}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
mark() { echo "$(date +%s.%N) $*" >> "$OUT/marks.txt"; }
: > "$OUT/marks.txt"
mark "probe start"
timeout 150 tools/ubench/clock_probe 120 10 > "$OUT/probe.txt" 2>&1 &
PROBE=$!
sleep 2
mark "kbench synth-only start"
timeout 200 python tools/kbench.py --smooth --chain --synth-only --steps 1500 --warmup 4 2>/dev/null | cut -c1-400 > "$OUT/kbench.json"
mark "kbench end"
sleep 1
mark "bench leg start"
timeout 300 python bench.py --steps 40 --repeats 3 --no-extras 2>/dev/null | cut -c1-600 > "$OUT/bench.json"
mark "bench end"
sleep 1
kill $PROBE 2>/dev/null
wait $PROBE 2>/dev/null
python - "$OUT" <<'PY'
import sys, os
out = sys.argv[1]
marks = [(float(l.split()[0]), " ".join(l.split()[1:])) for l in open(os.path.join(out, "marks.txt"))]
marks = [(t - marks[0][0], n) for t, n in marks[1:]]
rows = []
for l in open(os.path.join(out, "probe.txt")):
    p = l.split()
    if len(p) >= 4 and p[1] == "s":
        rows.append((float(p[0]), float(p[2])))
print("marks", marks)
# histogram of MHz per interval between marks
edges = [0.0] + [m[0] for m in marks] + [1e9]
names = ["idle before"] + [m[1] for m in marks]
for i in range(len(edges) - 1):
    v = sorted(r[1] for r in rows if edges[i] <= r[0] < edges[i + 1])
    if v:
        print("%-28s n %4d  min %7.1f  p10 %7.1f  median %7.1f  p90 %7.1f  max %7.1f" % (names[i], len(v), v[0], v[len(v) // 10], v[len(v) // 2], v[9 * len(v) // 10], v[-1]))
PY
cat "$OUT/kbench.json" "$OUT/bench.json" | cut -c1-300
