#!/bin/bash
# gpurun -- 'bash tools/run_attn_loo.sh'   leave-one-out timing builds of the f16x3 attention kernel (tools/attn_loo_exp.hip), each under its own timeout
mkdir -p gpurun_out
for b in attn_loo_exp attn_loo_exp_1 attn_loo_exp_2 attn_loo_exp_4 attn_loo_exp_8 attn_loo_exp_16 attn_loo_exp_32 attn_loo_exp_64 attn_loo_exp_10 attn_loo_exp_42 attn_loo_exp_46 attn_loo_exp_47 attn_loo_exp_63 attn_loo_exp_127 attn_loo_exp; do
  [ -x tools/$b ] || continue
  timeout 120 tools/$b 2>&1 | tail -8
done 2>&1 | tee gpurun_out/attn_loo.txt
