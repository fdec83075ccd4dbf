#!/bin/bash
# round-4 GPU call: the whole -m gpu suite, the sweep (profiles/r04_var_tile_sweep.txt), kernel stats of one C5 object
set -o pipefail
out=$PWD/gpurun_out/r4t; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { grep -v "^  File" $out/pytest.txt | tail -40; exit 1; }
tail -3 $out/pytest.txt
python3 scripts/var_tile_sweep.py 2>&1 | grep -v amdgpu.ids | tee $out/sweep.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/scripts/c5_stages.py containerB gaussian 128 > $out/c5.log 2>&1 ) || { tail -20 $out/c5.log; exit 1; }
python3 scripts/prof_summary.py $out/prof $out/c5_kernel_stats.txt "scripts/c5_stages.py containerB gaussian 128 (3 precisions x 3 repetitions)" | head -30
