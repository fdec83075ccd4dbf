#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4f
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids || exit 1
GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 336 512 724 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/probe_gen.txt
GPX_VAR_COLS_CF=2 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 336 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/probe_gen_cf2.txt
GPX_VAR_COLS_GEN=0 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 512 724 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/probe_nogen.txt
GPX_VAR_COLS_CF=2 GPX_VAR_COLS_GEN=0 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/probe_nogen_cf2.txt
