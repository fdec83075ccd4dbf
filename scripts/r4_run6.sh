#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4g
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids || exit 1
GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 336 512 724 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4g/probe_w2_gen.txt
GPX_VAR_COLS_GEN=0 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 512 724 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4g/probe_w2_nogen.txt
