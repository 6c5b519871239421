#!/bin/bash
# Socket power, clocks and temperature (rocm-smi, sampled in the background) while the synthesis kernel runs alone and while the
# headline's stream runs: is the chip at its power cap?   bash tools/power_probe.sh <tag>
set -u
TAG="${1:-r06_power}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
rocm-smi --showmaxpower --showpower --showclocks --showtemp --showperflevel > "$OUT/idle.txt" 2>&1
rocm-smi --showpowerplay > /dev/null 2>&1
sample() { while true; do echo "T $(date +%s.%N)"; rocm-smi -P -c -t --showvoltage 2>/dev/null | grep -E "Power|sclk|fclk|mclk|Temp|Voltage" ; sleep 0.05; done; }
sample > "$OUT/samples.txt" &
SP=$!
sleep 1.5
echo "M $(date +%s.%N) kbench synth-only (k_synth_ev alone) start" >> "$OUT/marks.txt"
timeout 300 python tools/kbench.py --smooth --chain --synth-only --steps 4000 --warmup 4 2>/dev/null | cut -c1-300 > "$OUT/kbench_ev.json"
echo "M $(date +%s.%N) end" >> "$OUT/marks.txt"
sleep 1.5
echo "M $(date +%s.%N) stream leg (bench.py --no-extras) start" >> "$OUT/marks.txt"
timeout 300 python bench.py --steps 40 --repeats 12 --no-extras 2>/dev/null | cut -c1-400 > "$OUT/bench.json"
echo "M $(date +%s.%N) end" >> "$OUT/marks.txt"
sleep 1.5
echo "M $(date +%s.%N) kbench synth-only M1 (k_synth_pd alone) start" >> "$OUT/marks.txt"
timeout 300 python tools/kbench.py --fs 2.6e6 --nsamp 300000 --nch 12 --blocks 1000 --synth-only --steps 5000 --warmup 4 2>/dev/null | cut -c1-300 > "$OUT/kbench_pd.json"
echo "M $(date +%s.%N) end" >> "$OUT/marks.txt"
sleep 1
kill $SP 2>/dev/null; wait $SP 2>/dev/null
cat "$OUT/idle.txt" | grep -v "^$" | head -40
python - "$OUT" <<'PY'
import sys, os, re
out = sys.argv[1]
marks = [(float(l.split()[1]), " ".join(l.split()[2:])) for l in open(os.path.join(out, "marks.txt"))]
cur, rows = None, []
for l in open(os.path.join(out, "samples.txt")):
    if l.startswith("T "):
        cur = {"t": float(l.split()[1])}; rows.append(cur)
    elif cur is not None:
        m = re.search(r"([-+]?\d+\.?\d*)\s*(W|Mhz|c|mV)?\)?\s*$", l.strip().replace("(", " ").replace(")", " "))
        key = "power" if "Power" in l else "sclk" if "sclk" in l else "temp_" + l.split(":")[1].strip()[:28] if "Temp" in l else "volt" if "Volt" in l else None
        nums = re.findall(r"[-+]?\d+\.\d+|\d+", l.split(":")[-1])
        if key and nums:
            cur.setdefault(key, float(nums[-1]) if key != "sclk" else float(nums[-1]))
phases = []
t = [m[0] for m in marks]
names = ["idle"] + [m[1] for m in marks]
edges = [0.0] + t + [1e18]
for i in range(len(edges) - 1):
    sel = [r for r in rows if edges[i] + 0.3 <= r["t"] < edges[i + 1] - 0.05]
    if not sel: continue
    line = "%-58s n %4d" % (names[i][:58], len(sel))
    for k in sorted({k for r in sel for k in r if k != "t"}):
        v = sorted(r[k] for r in sel if k in r)
        if v: line += "  %s med %.0f max %.0f" % (k[:20], v[len(v) // 2], v[-1])
    print(line)
PY
head -c 300 "$OUT/kbench_ev.json"; echo; head -c 400 "$OUT/bench.json"; echo; head -c 300 "$OUT/kbench_pd.json"; echo
head -30 "$OUT/samples.txt"
