#!/bin/bash
# round 6: state of the tree -- whole GPU suite, then the default bench line
set -o pipefail
mkdir -p gpurun_out/r6b
python -m pytest tests -m gpu -x -q > gpurun_out/r6b/gpu_tests.txt 2>&1; rc=$?
tail -5 gpurun_out/r6b/gpu_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err || exit 1
tail -c 3000 gpurun_out/r6b/bench.json
