cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export UU3D_LIB="$GRAFT_REPO_ROOT/uplift-upsample-3dhpe_amd/csrc/libuu3d_timing.so"
run() { UU3D_SKIP=$1 python3 bench.py --timing-experiment --steps 200 --warmup 16 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('skip=$1 ($2) ms_per_step', d['ms_per_step'])"; }
run 0 "nothing"
run 1023 "every launch class with a bit: what is left = s2t, head1, head2, frame list, range check, graph launch, copies"
run 913 "spatial + chain + attention + all strided"
run 0 "nothing"
