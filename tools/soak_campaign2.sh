#!/bin/bash
# Round 3's wider soak (GPU box): every flavour of tools/fuzz_parity.py, the rare paths forced as well; prints one line per run.
#   bash tools/soak_campaign2.sh <first seed> <seeds per flavour>
S0=${1:-500}; N=${2:-4}
one() { local out; out=$("$@" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-260); echo "$out"; }
for ((s=S0; s<S0+N; s++)); do
  one timeout 900 python tools/fuzz_parity.py --stream --also-batch --cases 120 --seed $s
  one timeout 900 python tools/fuzz_parity.py --stream --low-rate --also-batch --cases 120 --seed $s
  one timeout 900 python tools/fuzz_parity.py --stream --ties --also-batch --cases 80 --seed $s
  one timeout 900 python tools/fuzz_parity.py --cases 300 --seed $s
  GPSBB_PY_LIB=exp GPSBB_PD_DANGER=4194304 one timeout 900 python tools/fuzz_parity.py --stream --low-rate --cases 40 --seed $s --budget 1.5e7
  GPSBB_PY_LIB=exp GPSBB_EV_DANGER=4194304 one timeout 900 python tools/fuzz_parity.py --stream --cases 40 --seed $s --budget 1.5e7
done
for ((s=S0; s<S0+(N+1)/2; s++)); do
  one timeout 1200 python tools/fuzz_parity.py --stream --also-batch --cases 30 --seed $s --nsamp-max 2500000 --budget 4e8
  one timeout 900 python tools/fuzz_parity.py --ev --seed $s
  one timeout 900 python tools/fuzz_parity.py --shapes --seed $s
done
