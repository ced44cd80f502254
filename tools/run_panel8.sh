#!/bin/bash
# gpurun -- 'bash tools/run_panel8.sh [binaries...]'   (every binary under its own timeout: a barrier mismatch would hang)
mkdir -p gpurun_out
bins="$@"; [ -z "$bins" ] && bins="mfma_chain_exp panel8_exp"
for b in $bins; do
  [ -x tools/$b ] || continue
  echo "=== $b" ; timeout 90 tools/$b 9088 2>&1 | grep -v "max |err\|determinism\|max |4-wave" | tail -30; echo "rc=$?"
done 2>&1 | tee gpurun_out/panel8_exp.txt
