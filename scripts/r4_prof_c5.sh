#!/bin/bash
# rocprofv3 kernel stats of one C5-shaped object (containerB, N = 724, 128^3 grid, fp32 mode)
set -o pipefail
out=$PWD/gpurun_out/r4p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/scripts/c5_stages.py containerB gaussian 128 > $out/c5.log 2>&1 || { tail -20 $out/c5.log; exit 1; }
cd $GRAFT_REPO_ROOT && python3 scripts/prof_summary.py $out/prof $out/c5_kernel_stats.txt "scripts/c5_stages.py containerB gaussian 128 (3 precisions x 3 repetitions)"
