#!/bin/bash
# gpurun -- 'bash tools/power_tchain16.sh' : power, clock and time of the chain's mid launch with every CU busy (32768 rows), complete and leave-one-out builds
# (tools/tchain16_exp_loo<mask>: 1 no refill DMA, 2 no finish; results of the loo builds are wrong) -- is a launch's time its ENERGY (same watts, other clock) or stalls?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed 's/^GPU\[0\]\s*: //; s/Current Socket Graphics Package Power (W): /W /; s/sclk clock level: [0-9]*: //' | tr '\n' ' '; }
for M in 32768 9088; do
for l in 0 1 2 3; do
  ONLY_MID=1 tools/tchain16_exp_loo$l $M 12000 50 > /tmp/pt.txt 2>&1 &
  bp=$!; best=""; bw=0
  while kill -0 $bp 2>/dev/null; do
    s="$(sample)"; w=$(echo "$s" | grep -o 'W [0-9.]*' | head -1 | cut -d' ' -f2 | cut -d. -f1)
    if [ -n "$w" ] && [ "$w" -gt "$bw" ]; then bw=$w; best="$s"; fi
    sleep 0.3
  done
  wait $bp
  echo "rows $M loo $l: $(grep -o '[0-9.]* us per launch' /tmp/pt.txt | head -1) | highest sample: $best"
done
done
