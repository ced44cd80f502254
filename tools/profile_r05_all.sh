#!/bin/bash
# gpurun --timeout 2400 -- 'bash tools/profile_r05_all.sh'  : the r05 profiles the DESIGN / bench line quote, reduced to gpurun_out/sum/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the benchmark workload (batch 128): traced / counted under the throughput schedule (two slots: the temporal chain), as the timed path runs it
TRACE_STREAMS=2 bash tools/profile_r05.sh r05_final
# the same with the round-4 launches (UU3D_TCHAIN=0) and under the latency schedule (what model(...) runs)
TRACE_STREAMS=2 UU3D_TCHAIN=0 bash tools/profile_r05.sh r05_no_tchain
TRACE_STREAMS=1 bash tools/profile_r05.sh r05_latency --no-secondary --no-cpu-baseline
# the reference's eval batch (512 windows per forward, 284 row tiles)
TRACE_STREAMS=2 bash tools/profile_r05.sh r05_tchain_b512 --batch 512 --no-secondary --no-cpu-baseline
# the driver's shape
python3 bench.py --steps 20 --warmup 5 > gpurun_out/sum/r05_bench_driver_shape.json 2>/dev/null
ls -la gpurun_out/sum | grep r05
