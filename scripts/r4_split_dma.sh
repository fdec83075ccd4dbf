#!/bin/bash
# A/B of the F32_SPLIT contraction kernels: GPX_SPLIT_DMA = 2 (LDS-DMA, 16x16x32), 1 (LDS-DMA, 32x32x16), 0 (register-staged)
set -o pipefail
out=$PWD/gpurun_out/r4s; mkdir -p $out
for v in ${VARIANTS_TEST:-2}; do
GPX_SPLIT_DMA=$v timeout -k 10 500 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/tests$v.log 2>&1 || { tail -30 $out/tests$v.log; exit 1; }
tail -1 $out/tests$v.log
done
for dma in ${VARIANTS:-2 1 0 2 1 0}; do
  GPX_SPLIT_DMA=$dma timeout -k 10 200 python3 bench.py --precision f32split --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --no-configs > $out/bench_dma$dma.json 2> $out/bench_dma$dma.err || { tail -20 $out/bench_dma$dma.err; exit 1; }
  python3 - $out/bench_dma$dma.json $dma <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("GPX_SPLIT_DMA=%s ms_per_step %.1f kernel %.3f ms frac %.3f accuracy %s" % (sys.argv[2], d["ms_per_step"], r["avg_launch_ms"], r["frac"], json.dumps(d.get("accuracy"))[:400]))
PY
done
