#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4c/pytest.txt 2>&1 || { tail -40 gpurun_out/r4c/pytest.txt; exit 1; }
tail -3 gpurun_out/r4c/pytest.txt
GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 512 724 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/probe_gen.txt
GPX_VAR_COLS_GEN=0 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 512 724 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c/probe_nogen.txt
for o in bowlA containerB mugD; do
  python3 scripts/c5_stages.py $o gaussian 128 2>&1 | grep F32 >> gpurun_out/r4c/c5_stages.txt || exit 1
done
cat gpurun_out/r4c/c5_stages.txt
