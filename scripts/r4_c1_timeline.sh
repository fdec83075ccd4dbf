#!/bin/bash
# kernel timeline of create() at the node's model size (N = 277, fp64 and fp32 mode): where the 0.6 ms of C1's create go
set -o pipefail
out=$PWD/gpurun_out/r4t; mkdir -p $out
for p in f64 f32; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/$p -- python3 $GRAFT_REPO_ROOT/scripts/la_check.py 277 $p 6 > $out/$p.log 2>&1 ) || { tail -20 $out/$p.log; exit 1; }
  grep create $out/$p.log | tail -n 2
  python3 scripts/timeline.py $out/$p | tee $out/timeline_$p.txt | head -40
done
