#!/bin/bash
# A/B of the diagonal-fragment skipping in var_w1_kernel (GPX_VAR_DIAG_SKIP=0: off): parity tests, then the size sweep
set -o pipefail
out=$PWD/gpurun_out/r4d; mkdir -p $out
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/tests.log 2>&1 || { grep -v "^  File" $out/tests.log | tail -30; exit 1; }
tail -n 2 $out/tests.log
for s in 1 0 1 0; do
  GPX_VAR_DIAG_SKIP=$s timeout -k 10 200 python3 scripts/var_tile_sweep.py "skip=$s" 2>&1 | grep -E "N= *(1536|2048|3072|4096|8192|16384)" | tee -a $out/sweep.txt
done
