#!/bin/bash
# gpurun -- 'bash tools/run_tchain_r64.sh': what a 64-row tile would cost on the 128-row skeleton (timing builds, results wrong):
# loo 1024 = the waves of token panels 2 and 3 only refill the ring and keep the barriers; 1026 = + no epilogues (everything on chip)
for M in 32768 65536; do
  for b in 0 2 1024 1026; do
    echo "tiles $((M / 128)) loo $b: $(timeout 200 tools/tchain_exp_loo$b $M 30 60 2>&1 | grep -A4 '^=== mid: ' | grep -i ' us\|wg   0' | head -2 | tr '\n' ' ')"
  done
done
