#!/bin/bash
# kernel timeline of create() at the node's model sizes (N = 277 and 724, fp64 and fp32 mode) -- the three launches of
# csrc/gpx_small.hip; GPX_DATAFLOW=0 in the environment gives the general chain for comparison
set -o pipefail
out=$PWD/gpurun_out/r5t${GPX_SMALL_CREATE:+_chain}; mkdir -p $out
for n in 277 724; do
for p in f64 f32; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/${n}_$p -- python3 $GRAFT_REPO_ROOT/scripts/la_check.py $n $p 6 > $out/${n}_$p.log 2>&1 ) || { tail -20 $out/${n}_$p.log; exit 1; }
  echo "## N = $n, $p mode" | tee -a $out/timeline.txt
  grep create $out/${n}_$p.log | tail -n 1 | sed 's/^/# /' | tee -a $out/timeline.txt
  python3 scripts/timeline.py $out/${n}_$p | tee -a $out/timeline.txt
done
done
