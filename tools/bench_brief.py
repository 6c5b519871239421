#!/usr/bin/env python3
"""the one JSON line of bench.py, condensed: python tools/bench_brief.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("value %.4g  ms/step %.2f  incl_seed %.4g  seed %.4f s  n_gpus %d" % (d["value"], d["ms_per_step"], d.get("value_incl_seed", 0), d["shard_seed_s"], d["n_gpus"]))
print("repeats", ["%.3f" % x for x in d["repeats"]["seconds"]])
r = d["roofline"]
print("roofline achieved %.0f GB/s frac %.4f  ms/launch %.3f  alone %s" % (r["achieved"], r["frac"], r["ms_per_launch"], json.dumps(r.get("alone"))[:120]))
print("prepass ms %.2f  chain %s" % (d["prepass_ms_per_launch"], json.dumps(d["device_chain"])))
if d.get("parity"):
    print("parity checked %d mismatching %d digest %s" % (d["parity"]["checked_blocks"], d["parity"]["mismatching_blocks"], d["parity"]["stream_end_state_digest"]))
if "m1" in d and "roofline" in d["m1"]:
    print("m1 roofline", json.dumps(d["m1"]["roofline"])[:300])
for k in ("resident", "m1", "gather", "cpu_baseline"):
    if k in d:
        print(k, json.dumps(d[k])[:420])
