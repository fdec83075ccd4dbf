#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4i
GPX_VAR_COLS_SHAPE=4 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids || exit 1
echo "== shape 4 gen"; GPX_VAR_COLS_SHAPE=4 GPX_VC_DBG=1 python3 scripts/vc_probe.py 277 336 512 724 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4i/probe_s4_gen.txt
