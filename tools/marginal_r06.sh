#!/bin/bash
# gpurun -- 'bash tools/marginal_r06.sh' : what each launch class costs the pipelined step -- the bench loop on the TIMING build of the library
# (python uplift-upsample-3dhpe_amd/build.py --timing -> csrc/libuu3d_timing.so; UU3D_SKIP leaves launch classes out: results wrong, time only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export UU3D_LIB="$GRAFT_REPO_ROOT/uplift-upsample-3dhpe_amd/csrc/libuu3d_timing.so"
run() { UU3D_SKIP=$1 python3 bench.py --timing-experiment --steps 200 --warmup 16 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('skip=$1 ($2) ms_per_step', d['ms_per_step'])"; }
run 0 "nothing"
run 1 "spatial stack"
run 128 "temporal chain launches (6)"
run 16 "attention (6 launches)"
run 144 "temporal chain + attention"
run 256 "strided blocks 2 and 3"
run 512 "strided block 1 (attention, chain launch, convolution)"
run 768 "all strided blocks"
run 145 "spatial stack + temporal chain + attention"
run 0 "nothing"
