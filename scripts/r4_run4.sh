#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4e
export GPX_LIB=$PWD/gaussian-object-modelling_amd/lib_t/libgpx.so
GPX_VAR_COLS_CF=2 GPX_VAR_COLS_GEN=0 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/t_nogen_cf2.txt
GPX_VAR_COLS_CF=2 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/t_gen_cf2.txt
GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/t_gen_cf3.txt
