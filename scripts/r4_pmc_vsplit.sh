#!/bin/bash
# SQ counters of vsplit_gemm_kernel (the F32_SPLIT contraction): two passes of <= 4 counters over a short F32_SPLIT bench run
set -o pipefail
out=$PWD/gpurun_out/r4v; mkdir -p $out
PMC_TIMEOUT=400 bash scripts/pmc_pass.sh $out/p1 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" -- python3 bench.py --precision f32split --steps 1 --warmup 0 --nq 131072 --no-cpu-baseline --no-fast-mode --no-configs || exit 1
PMC_TIMEOUT=400 bash scripts/pmc_pass.sh $out/p2 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" -- python3 bench.py --precision f32split --steps 1 --warmup 0 --nq 131072 --no-cpu-baseline --no-fast-mode --no-configs || exit 1
python3 scripts/pmc_summary.py vsplit_gemm $out/p1 $out/p2 | tee $out/summary.txt
