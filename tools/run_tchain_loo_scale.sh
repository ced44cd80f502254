#!/bin/bash
# gpurun -- 'bash tools/run_tchain_loo_scale.sh' : the chain kernel alone on 71 / 256 / 284 / 512 row tiles, complete and with one component left out
# (timing builds tools/tchain_exp_loo<mask>: 1 no refill DMA, 2 no finish (epilogues), 4 no exchange, 8 no mid barrier, 32 no MFMA, 256 transitions without memory traffic)
for M in 9088 32768 36352; do
  for b in 0 1 2 32 256 258 259; do
    echo "tiles $((M / 128)) loo $b: $(timeout 120 tools/tchain_exp_loo$b $M 30 60 2>&1 | grep -A3 '^=== mid: ' | grep -i ' us' | head -1)"
  done
done
