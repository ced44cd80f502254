#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace + separate --pmc passes (kernel-trace only, as the pool requires) of one bench
# command, reduced ON the box to small CSVs in gpurun_out/sum/ (the rocpd databases exceed what gpurun pulls back).
# TRACE_STREAMS=2: the traced / counted runs go through a two-slot pipeline = the THROUGHPUT schedule (where the temporal chain runs).
# Round 4: the traced / counted runs use ONE batch in flight (--streams 1: under rocprofv3 the queues serialise anyway, and a
# kernel's own duration is what the roofline wants); the timed bench lines (default: four in flight) are written next to them.
#   gpurun -- 'bash tools/profile_r06.sh <tag> [bench args...]'
set -u
tag=${1:-r06}; shift
args="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/sum; rm -rf gpurun_out/${tag}_w_*
W=gpurun_out/${tag}_w
python3 bench.py --steps 200 --warmup 20 $args > gpurun_out/sum/${tag}_bench_default.json 2> gpurun_out/sum/${tag}_bench_default.err
python3 bench.py --steps 200 --warmup 20 --streams 1 --no-secondary --no-cpu-baseline $args > gpurun_out/sum/${tag}_bench_streams1.json 2>/dev/null
python3 bench.py --steps 200 --warmup 20 --streams 2 --no-secondary --no-cpu-baseline $args > gpurun_out/sum/${tag}_bench_streams2.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d ${W}_trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --streams ${TRACE_STREAMS:-1} $args > gpurun_out/sum/${tag}_trace.log 2>&1
python3 tools/rocpd_summary.py stats gpurun_out/sum/${tag}_kernel_stats.csv $(find ${W}_trace -name '*.db' | head -1)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d ${W}_pmc_$n -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-graph --streams ${TRACE_STREAMS:-1} $args > gpurun_out/sum/${tag}_pmc_$n.log 2>&1
done
python3 tools/rocpd_summary.py pmc gpurun_out/sum/${tag}_pmc_summary.csv $(find ${W}_pmc_* -name '*.db')
rm -rf gpurun_out/${tag}_w_*
ls -la gpurun_out/sum/ | grep ${tag}
