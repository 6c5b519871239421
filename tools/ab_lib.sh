#!/bin/bash
# two builds of the library on one box, one session, alternating: a short bench.py run each (the headline, the node driver's one shard)
#   bash tools/ab_lib.sh <path of the other libgpsbb.so> [rounds]
OTHER="$1"; N="${2:-2}"
for ((r=0; r<N; r++)); do
  for lib in product "$OTHER"; do
    if [ "$lib" = product ]; then unset GPSBB_PY_LIB; else export GPSBB_PY_LIB="$lib"; fi
    python bench.py --steps 20 --repeats 3 --no-cpu --no-m1-stream > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err
    python - "$lib" <<'PY'
import json, sys
b = json.load(open("gpurun_out/ab_lib.json")); nd = b["node_driver"]
print("%-40s value %.4g  k_synth_ev %.4f (alone %.4f)  one_shard %.4g (%.3f)  %s" % (sys.argv[1][-40:], b["value"], b["roofline"]["ms_per_launch"], b["roofline"]["alone"]["ms_per_launch"],
      nd["one_shard"]["value"], nd["one_shard"]["vs_headline"], {k: "%.4g" % v["value"] for k, v in nd.get("all_gpus", {}).items() if isinstance(v, dict)}))
PY
  done
done
