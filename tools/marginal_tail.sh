#!/bin/bash
run() { UU3D_SKIP=$1 python3 bench.py --steps 200 --warmup 16 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('skip=$1 ($2) ms_per_step', d['ms_per_step'])"; }
run 0 "nothing"; run 256 "strided blocks 2 and 3"; run 512 "strided block 1 (attention, chain launch, convolution)"; run 768 "all strided blocks"; run 0 "nothing"
