"""the few numbers of a bench line that the kernel work is steered by:  python tools/bench_brief.py gpurun_out/<tag>/bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value %.4g  ms/step %.3f  k_synth_ev in leg %.4f ms (alone %.4f)  frac %.4f  prepass_ms_per_launch %s" % (
    d["value"], d["ms_per_step"], r["ms_per_launch"], r.get("alone", {}).get("ms_per_launch", float("nan")), r["frac"], d.get("prepass_ms_per_launch")))
m = d.get("m1", {})
if m:
    g = m.get("gpu", {})
    print("m1: value %.4g  ms/step %.4f  synth %.4f  prepass %.4f  roofline %s" % (g.get("value", 0), g.get("ms_per_step", 0), g.get("synth_kernel_ms", 0), g.get("prepass_ms", 0),
          {k: m.get("roofline", {}).get(k) for k in ("frac", "ms_per_launch")}))
for k in ("resident", "gather", "node_driver", "device_chain", "parity", "fixed_carrier"):
    if k in d:
        print(k, json.dumps(d[k])[:400])
