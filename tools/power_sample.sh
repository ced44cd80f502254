#!/bin/bash
# gpurun -- 'bash tools/power_sample.sh' : socket power and clocks while the pipelined bench loop runs (is the step power bound?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo; }
echo "idle: $(sample)"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -2
python3 bench.py --steps 20000 --warmup 16 --no-secondary --no-cpu-baseline > /tmp/ps_bench.json 2>/dev/null &
BP=$!
n=0
while kill -0 $BP 2>/dev/null; do n=$((n+1)); echo "t=$n: $(sample)"; sleep 0.4; done
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/ps_bench.json').readlines()[-1]); print('bench: %.1f k seq/s, %.4f ms/step over %d steps' % (d['value']/1e3, d['ms_per_step'], d['steps']))"
