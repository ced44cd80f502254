#!/bin/bash
# gpurun --timeout 2400 -- 'bash tools/profile_r05_all.sh'  : the r05 profiles the DESIGN / bench line quote, reduced to gpurun_out/sum/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_r05.sh r05_final
# the temporal chain where it is chosen by size: the reference's eval batch (512 windows per forward, 284 row tiles); traced one batch at a time
TRACE_STREAMS=2 bash tools/profile_r05.sh r05_tchain_b512 --batch 512
TRACE_STREAMS=2 UU3D_TCHAIN=0 bash tools/profile_r05.sh r05_no_tchain_b512 --batch 512
ls -la gpurun_out/sum | grep r05
