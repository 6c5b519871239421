export GPSBB_PY_LIB=exp FILL_CALLS=600
for rep in 1 2; do
for cfg in "-" "GPSBB_FILL_ONE_STREAM=0" "GPSBB_FILL_TAIL_KERNEL=0" "GPSBB_FILL_ONE_STREAM=0 GPSBB_FILL_TAIL_KERNEL=0" "FILL_PIN=1" "FILL_PIN=1 GPSBB_FILL_ONE_STREAM=0 GPSBB_FILL_TAIL_KERNEL=0"; do
  envs=""; [ "$cfg" != "-" ] && envs="$cfg"
  echo "== $cfg"; env $envs python tools/fill_timeline.py 2>&1 | grep -E "median|Register|rror"
done; done
