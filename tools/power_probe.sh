#!/bin/bash
# gpurun -- 'bash tools/power_probe.sh'  : board power / clocks sampled while the bench runs (is the forward power limited?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocm-smi --showpower --showclocks -M 2>&1 | grep -v "^=\|^$" | head -30
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power (W)\|sclk" | sed 's/.*: *//' | tr '\n' ' '; echo; }
echo "idle: $(sample)"
probe() {  # $1 = label, $2 = settle seconds, rest = command
  label=$1; settle=$2; shift; shift
  "$@" > /tmp/bench_out.txt 2>/dev/null &
  pid=$!
  sleep $settle
  for i in $(seq 1 8); do echo "$label: $(sample)"; sleep 0.3; done
  wait $pid
  head -c 300 /tmp/bench_out.txt | python3 -c "import sys,json
try:
    d=json.loads(sys.stdin.readline()); print('$label', d['value'], d['ms_per_step'])
except Exception as e: pass"
}
probe "forward streams=4" 12 python3 bench.py --steps 30000 --warmup 100 --no-secondary --no-cpu-baseline
probe "forward streams=1" 12 python3 bench.py --steps 20000 --warmup 100 --streams 1 --no-secondary --no-cpu-baseline
probe "bare MFMA loops (tools/mfma_chain_exp x 60)" 3 bash -c 'for i in $(seq 1 60); do tools/mfma_chain_exp > /dev/null; done'
