#!/bin/bash
# gpurun --timeout 2400 -- 'bash tools/profile_r04_all.sh'  : every r04 profile the DESIGN / bench line quote, reduced to gpurun_out/sum/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_r04.sh r04_final
bash tools/profile_r04.sh r04_final_h36m81 --config h36m_81 --batch 256
bash tools/profile_r04.sh r04_final_dense351 --config dense_351 --batch 32
# training step (BASELINE config 5): kernel trace + MFMA-busy / traffic counters
W=gpurun_out/r04_train_w; rm -rf ${W}_*
python3 bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/sum/r04_final_bench_train.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d ${W}_trace -o run -- python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sum/r04_train_trace.log 2>&1
python3 tools/rocpd_summary.py stats gpurun_out/sum/r04_final_train_kernel_stats.csv $(find ${W}_trace -name '*.db' | head -1)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d ${W}_pmc_$n -o run -- python3 bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sum/r04_train_pmc_$n.log 2>&1
done
python3 tools/rocpd_summary.py pmc gpurun_out/sum/r04_final_train_pmc_summary.csv $(find ${W}_pmc_* -name '*.db')
rm -rf ${W}_*
ls -la gpurun_out/sum | grep r04
