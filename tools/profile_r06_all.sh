#!/bin/bash
# NOTE: the library switches UU3D_TCHAIN64 / UU3D_TCHAIN16 this script alternates existed up to commit ed9e71b; the two earlier chain kernels
# now live under tools/ (uu3d_tchain.h, uu3d_tchain64.h) with their stand-alone harnesses.  Check that commit out to rerun this A/B.
# gpurun --timeout 2400 -- 'bash tools/profile_r06_all.sh'  : the r06 profiles DESIGN.md / the bench line quote, reduced to gpurun_out/sum/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the benchmark workload (batch 128): traced / counted under the throughput schedule (two slots: the 64-row temporal chain), as the timed path runs it
TRACE_STREAMS=2 bash tools/profile_r06.sh r06_final
# the round-5 chain kernel on the same box (UU3D_TCHAIN64=0), bench lines only
UU3D_TCHAIN64=0 python3 bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/sum/r06_tchain128_bench_200.json 2>/dev/null
UU3D_TCHAIN64=0 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/sum/r06_tchain128_bench_driver_shape.json 2>/dev/null
# the driver's shape, three times
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/sum/r06_bench_driver_shape_$i.json 2>/dev/null; done
# the training step's kernels
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_w_train -o run -- python3 bench.py --mode train --steps 10 --warmup 3 > gpurun_out/sum/r06_final_train_trace.log 2>&1
python3 tools/rocpd_summary.py stats gpurun_out/sum/r06_final_train_kernel_stats.csv $(find gpurun_out/r06_w_train -name '*.db' | head -1)
rm -rf gpurun_out/r06_w_train
ls -la gpurun_out/sum | grep r06
