#!/bin/bash
# round 6: vector-ALU / matrix-pipe accounting of the two small-model variance kernels: executed vector
# instructions, MFMAs, matrix-pipe busy cycles and the cycles in which the vector ALU and the matrix pipe work at the same time
set -o pipefail
out=$PWD/gpurun_out/r6h; mkdir -p $out
: > $out/summary.txt
for prec in f64 f32; do
 for n in 277 512 724; do
  d=$out/${prec}_$n
  PMC_TIMEOUT=200 bash scripts/pmc_pass.sh ${d}_a "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" -- python3 scripts/small_one.py $n $prec 3 || exit 1
  PMC_TIMEOUT=200 bash scripts/pmc_pass.sh ${d}_b "SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" -- python3 scripts/small_one.py $n $prec 3 || exit 1
  PMC_TIMEOUT=200 bash scripts/pmc_pass.sh ${d}_c "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_F32" -- python3 scripts/small_one.py $n $prec 3 || exit 1
  echo "== $prec N = $n" | tee -a $out/summary.txt
  python3 scripts/pmc_summary.py var_cols ${d}_a ${d}_b ${d}_c | tee -a $out/summary.txt
 done
done
