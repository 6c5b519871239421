show() { python - "$1" <<PY
import json,sys
r=json.load(open(sys.argv[1]))
nd=r["node_driver"]; print(sys.argv[1].split("/")[-1], "value %.4g one_shard %.4g vs_headline %.3f"%(r["value"],nd["one_shard"]["value"],nd["one_shard"]["vs_headline"]), {k:"%.4g"%v["value"] for k,v in nd["all_gpus"].items() if isinstance(v,dict)})
PY
}
python bench.py --steps 20 --warmup 5 --no-cpu --no-m1-stream 2>/dev/null > gpurun_out/ne_a.json; show gpurun_out/ne_a.json
python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null > gpurun_out/ne_b.json; show gpurun_out/ne_b.json
GPSBB_PY_LIB=exp GPSBB_STREAM_SEED_STREAMS=4 python bench.py --steps 20 --warmup 5 --no-cpu --no-m1-stream 2>/dev/null > gpurun_out/ne_c.json; show gpurun_out/ne_c.json
GPSBB_PY_LIB=exp GPSBB_STREAM_SEED_STREAMS=4 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null > gpurun_out/ne_d.json; show gpurun_out/ne_d.json
