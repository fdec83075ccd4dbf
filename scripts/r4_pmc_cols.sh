#!/bin/bash
# SQ / TCC counters of var_cols_kernel on a C5-shaped object (mugD, N = 277; containerB, N = 724), 128^3 queries, fp32 mode
set -o pipefail
out=$PWD/gpurun_out/r4c; mkdir -p $out
for o in mugD containerB; do
  PMC_TIMEOUT=300 bash scripts/pmc_pass.sh $out/${o}_p1 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" -- python3 scripts/c5_stages.py $o gaussian 128 || exit 1
  PMC_TIMEOUT=300 bash scripts/pmc_pass.sh $out/${o}_p2 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" -- python3 scripts/c5_stages.py $o gaussian 128 || exit 1
  PMC_TIMEOUT=300 bash scripts/pmc_pass.sh $out/${o}_p3 "FETCH_SIZE" -- python3 scripts/c5_stages.py $o gaussian 128 || exit 1
  PMC_TIMEOUT=300 bash scripts/pmc_pass.sh $out/${o}_p4 "WRITE_SIZE" -- python3 scripts/c5_stages.py $o gaussian 128 || exit 1
  echo "== $o"; python3 scripts/pmc_summary.py var_cols_kernel $out/${o}_p1 $out/${o}_p2 $out/${o}_p3 $out/${o}_p4 | tee $out/${o}_summary.txt
done
