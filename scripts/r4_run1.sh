#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4b
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4b/pytest.txt 2>&1 || { tail -30 gpurun_out/r4b/pytest.txt; exit 1; }
tail -3 gpurun_out/r4b/pytest.txt
python3 scripts/var_tile_sweep.py "var_cols CF=2" > gpurun_out/r4b/sweep.txt 2>&1 || exit 1
for o in bowlA containerB mugD; do
  python3 scripts/c5_stages.py $o gaussian 128 >> gpurun_out/r4b/c5_stages.txt 2>&1 || exit 1
done
cat gpurun_out/r4b/sweep.txt gpurun_out/r4b/c5_stages.txt
