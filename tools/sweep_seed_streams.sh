for rep in 1 2; do for n in 2 3 4; do
GPSBB_PY_LIB=exp GPSBB_STREAM_SEED_STREAMS=$n python bench.py --steps 20 --repeats 5 --no-extras 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('seed_streams $n: value %.4g  synth %.4f  prepass %.3f  min %.4g max %.4g' % (r['value'], r['roofline']['ms_per_launch'], r['prepass_ms_per_launch'], r['repeats']['value_min'], r['repeats']['value_max']))"
done; done
