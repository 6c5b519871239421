#!/bin/bash
# a longer parity soak on the GPU box: bash tools/soak_campaign.sh <first seed> <seeds> ; stops at the first mismatch
S0=${1:-100}; N=${2:-8}
for ((s=S0; s<S0+N; s++)); do
  timeout 900 python tools/fuzz_parity.py --stream --cases 150 --seed $s 2>&1 | tail -1 | cut -c1-400 || exit 1
  grep -q MISMATCH <(timeout 1 true) 
done
for ((s=S0; s<S0+N/2; s++)); do
  timeout 1200 python tools/fuzz_parity.py --stream --cases 40 --seed $s --nsamp-max 2500000 --budget 4e8 2>&1 | tail -1 | cut -c1-400
done
for ((s=S0; s<S0+2; s++)); do
  timeout 900 python tools/fuzz_parity.py --ev --seed $s 2>&1 | tail -1 | cut -c1-300
  timeout 900 python tools/fuzz_parity.py --seed $s 2>&1 | tail -1 | cut -c1-300
done
