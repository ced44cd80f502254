#!/bin/bash
# gpurun -- 'bash tools/run_tchain_mix.sh': the two chain kernels with n launches of ONE forward's tiles side by side, every launch on its own weight stream
for n in 1 2 4; do
  echo "128-row kernel, 71 tiles, MIX $n: $(MIX=$n timeout 200 tools/tchain_exp 9088 40 20 2>&1 | grep -A6 '^=== mid: ' | grep 'MIX')"
  echo " 64-row kernel, 142 tiles, MIX $n: $(MIX=$n timeout 200 tools/tchain64_exp 9088 40 20 2>&1 | grep -A6 '^=== mid: ' | grep 'MIX')"
done
