export GPSBB_PY_LIB=exp   # the environment knobs below exist in the experiments build only (libgpsbb_exp.so)
# sweep of the stream ring's depth and the number of pre-pass streams (bench.py's headline leg only)
# usage: bash tools/sweep_streams.sh "8 16 8" "4 8 6" ...   (seed streams, hardware queues, ring depth)
if [ $# -eq 0 ]; then set -- "8 16 8" "8 16 12" "6 12 8" "4 8 6"; fi
for cfg in "$@"; do set -- $cfg
echo "== seed streams $1, hw queues $2, depth $3"
GPSBB_STREAM_SEED_STREAMS=$1 GPU_MAX_HW_QUEUES=$2 python bench.py --no-extras --steps 10 --repeats 3 --warmup 2 --depth $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['prepass_ms_per_launch'], d['repeats']['seconds'])"
done
