#!/bin/bash
# gpurun -- 'bash tools/power_by_class.sh' : socket power and shader clock of the pipelined loop with only ONE launch class left in the forward
# (TIMING build, UU3D_SKIP leaves the others out: results wrong, power and time only).  Which classes run into the 1400 W cap?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export UU3D_LIB="$GRAFT_REPO_ROOT/uplift-upsample-3dhpe_amd/csrc/libuu3d_timing.so"
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed 's/^GPU\[0\]\s*: //; s/Current Socket Graphics Package Power (W): /W /; s/sclk clock level: [0-9]*: //' | tr '\n' ' '; }
one() {   # $1 = skip mask, $2 = label, $3 = steps
  UU3D_SKIP=$1 python3 bench.py --timing-experiment --steps $3 --warmup 16 --no-secondary --no-cpu-baseline > /tmp/pc_bench.json 2>/dev/null &
  local bp=$! ; local best="" ; local bw=0
  while kill -0 $bp 2>/dev/null; do
    s="$(sample)"; w=$(echo "$s" | grep -o 'W [0-9.]*' | head -1 | cut -d' ' -f2 | cut -d. -f1)
    if [ -n "$w" ] && [ "$w" -gt "$bw" ]; then bw=$w; best="$s"; fi
    sleep 0.3
  done
  wait $bp
  echo "$2: $(python3 -c "import json; d=json.loads(open('/tmp/pc_bench.json').readlines()[-1]); print('%.4f ms per step' % d['ms_per_step'])") | highest sample: $best"
}
one 0    "whole forward                 " 12000
one 895  "temporal chain launches only  " 20000
one 1022 "spatial stack only            " 40000
one 1007 "attention launches only       " 60000
one 255  "strided stack only            " 40000
one 1023 "nothing with a bit (s2t, heads)" 80000
