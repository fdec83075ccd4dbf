#!/bin/bash
# kernel timeline of create() at the headline size (N = 16384, fp32 and fp64): which kernels the 18 / 43 ms of the LDL^T are, busy / idle
set -o pipefail
out=$PWD/gpurun_out/r4t; mkdir -p $out
for p in f32 f64; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/big_$p -- python3 $GRAFT_REPO_ROOT/scripts/la_check.py 16384 $p 3 > $out/big_$p.log 2>&1 ) || { tail -20 $out/big_$p.log; exit 1; }
  grep create $out/big_$p.log | tail -n 1
  python3 scripts/timeline.py $out/big_$p | tee $out/timeline_big_$p.txt | head -30
done
