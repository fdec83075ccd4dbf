#!/bin/bash
# accuracy of the F32_SPLIT variants against the fp64 pipeline on the headline workload (bench.py's fast_mode.accuracy)
set -o pipefail
out=$PWD/gpurun_out/r4s; mkdir -p $out
for dma in ${VARIANTS:-2 1}; do
  GPX_SPLIT_DMA=$dma timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $out/acc_dma$dma.json 2> $out/acc_dma$dma.err || { tail -20 $out/acc_dma$dma.err; exit 1; }
  python3 - $out/acc_dma$dma.json $dma <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("GPX_SPLIT_DMA=%s fast_mode %s\n   native accuracy %s" % (sys.argv[2], json.dumps({k:v for k,v in d["fast_mode"].items() if k!="variance_accuracy"}), json.dumps(d.get("accuracy"))))
PY
done
