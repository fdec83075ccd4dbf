#!/bin/bash
# round 6, first GPU call: whole GPU suite on the build with the two-wave sub-block LDL^T in both precisions, then the LDL^T sweep
set -o pipefail
mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -x -q > gpurun_out/r6a/gpu_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r6a/gpu_tests.txt
[ $rc -ne 0 ] && exit $rc
GPX_TRAIN_F64_MAX=0 timeout -k 10 300 python scripts/ldlt_sweep.py 5 > gpurun_out/r6a/ldlt_sweep.txt 2>&1 || exit 1
cat gpurun_out/r6a/ldlt_sweep.txt
