#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4h
VC_SIZES=100,277,300,512,724,1000 python3 scripts/vc_check.py 2>&1 | grep -v amdgpu.ids || exit 1
for sh in 0 2 3; do
  echo "== shape $sh gen"; GPX_VAR_COLS_SHAPE=$sh GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 336 512 724 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4h/probe_s${sh}_gen.txt
done
echo "== shape 0 nogen"; GPX_VAR_COLS_SHAPE=0 GPX_VAR_COLS_GEN=0 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 512 724 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4h/probe_s0_nogen.txt
