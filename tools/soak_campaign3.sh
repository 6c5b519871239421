#!/bin/bash
# Round 5's soak (GPU box): the flavours of tools/fuzz_parity.py with the pre-pass drawn per case (lap-parallel, row walks,
# automatic), then every flavour FORCED onto the lap-parallel pre-pass — as it is, and with its reference states pushed off the
# model (GPSBB_LAP_JITTER: links break, the repair kernel works) —, then long blocks.  One line per run.
#   bash tools/soak_campaign3.sh <first seed> <seeds per flavour>
S0=${1:-700}; N=${2:-4}
one() { local out; out=$("$@" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-330); echo "$out"; }
for ((s=S0; s<S0+N; s++)); do
  one timeout 900 python tools/fuzz_parity.py --stream --also-batch --cases 120 --seed $s
  one timeout 900 python tools/fuzz_parity.py --stream --low-rate --also-batch --cases 120 --seed $s
  one timeout 900 python tools/fuzz_parity.py --stream --ties --also-batch --cases 80 --seed $s
  one timeout 900 python tools/fuzz_parity.py --cases 300 --seed $s
  GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --also-batch --cases 120 --seed $((s+1000))
  GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --low-rate --also-batch --cases 120 --seed $((s+1000))
  GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --ties --also-batch --cases 80 --seed $((s+1000))
  GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --ev --cases 300 --seed $((s+1000))
  GPSBB_PY_LIB=exp GPSBB_LAP_JITTER=4000000000 GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --also-batch --cases 60 --seed $((s+2000))
  GPSBB_PY_LIB=exp GPSBB_LAP_JITTER=4000000000 GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --low-rate --cases 60 --seed $((s+2000))
  GPSBB_PY_LIB=exp GPSBB_LAP_JITTER=30000 GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --ties --cases 60 --seed $((s+2000))
  GPSBB_PY_LIB=exp GPSBB_LAP_JITTER=1000000000 GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --ev --cases 200 --seed $((s+2000))
  GPSBB_PY_LIB=exp GPSBB_LAP_UNIT_CARR=1 GPSBB_LAP_UNIT_CODE=1 GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --cases 60 --seed $((s+3000))
  GPSBB_PY_LIB=exp GPSBB_LAP_UNIT_CARR=9 GPSBB_LAP_UNIT_CODE=5 GPSBB_LAP_JITTER=2000000000 GPSBB_FUZZ_WHERE=3 one timeout 900 python tools/fuzz_parity.py --stream --also-batch --cases 60 --seed $((s+3000))
done
for ((s=S0; s<S0+(N+1)/2; s++)); do
  GPSBB_FUZZ_WHERE=3 one timeout 1200 python tools/fuzz_parity.py --stream --also-batch --cases 30 --seed $s --nsamp-max 2500000 --budget 4e8
  one timeout 900 python tools/fuzz_parity.py --shapes --seed $s
done
