for n in 2 3 4 5 6; do
  GPSBB_PY_LIB=exp GPSBB_STREAM_SEED_STREAMS=$n python bench.py --no-extras --steps 10 --repeats 2 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('seed_streams $n', 'value %.4g'%d['value'], 'kernel %.3f'%d['roofline']['ms_per_launch'], 'prepass %.2f'%d['prepass_ms_per_launch'])"
done
for dpt in 4 8; do
  GPSBB_PY_LIB=exp python bench.py --no-extras --steps 10 --repeats 2 --warmup 2 --depth $dpt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth $dpt', 'value %.4g'%d['value'], 'kernel %.3f'%d['roofline']['ms_per_launch'], 'prepass %.2f'%d['prepass_ms_per_launch'])"
done
for pb in 200 800; do
  GPSBB_PY_LIB=exp python bench.py --no-extras --steps 10 --repeats 2 --warmup 2 --push-blocks $pb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('push-blocks $pb', 'value %.4g'%d['value'], 'kernel %.3f'%d['roofline']['ms_per_launch'], 'prepass %.2f'%d['prepass_ms_per_launch'])"
done
