#!/bin/bash
# gpurun -- 'bash tools/run_tchain64_loo.sh': the 64-row chain kernel alone on 142 / 512 / 568 tiles, complete and with one component left out
# (timing builds tools/tchain64_exp_loo<mask>: 1 no refill DMA, 2 no finish (epilogues), 32 no MFMA)
for M in 9088 32768 36352; do
  for b in "" _loo1 _loo2 _loo32 _loo3; do
    echo "tiles $((M / 64)) ${b:-full}: $(timeout 200 tools/tchain64_exp$b $M 30 60 2>&1 | grep -A4 '^=== mid: ' | grep -i ' us\|wg   0' | head -2 | tr '\n' ' ')"
  done
done
