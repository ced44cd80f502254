#!/bin/bash
# gpurun -- 'bash tools/run_tchain_loo.sh'   leave-one-out timing builds of the temporal chain kernel (tools/tchain_exp_loo<mask>)
mkdir -p gpurun_out
for b in tools/tchain_exp_loo*; do
  echo "=== $b"; timeout 120 $b 9088 30 100 2>&1 | grep -A4 "^=== mid: \|^=== first: " | grep -v "rows vs\|second run"
done 2>&1 | tee gpurun_out/tchain_loo.txt
