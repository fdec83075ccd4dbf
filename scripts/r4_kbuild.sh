#!/bin/bash
set -o pipefail
out=$PWD/gpurun_out/r4k; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "mugd or ragged or reference_pin or kpp or update or fuzz or n16384" > $out/pytest.txt 2>&1 || { grep -v "^  File" $out/pytest.txt | tail -30; exit 1; }
tail -2 $out/pytest.txt
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-configs > $out/bench.json 2> $out/bench.err || { tail $out/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print('kbuild', d['roofline_kbuild']); print('kqp', {k:d['roofline_kqp'][k] for k in ('avg_launch_ms','frac')}); print('stages', d['stages_ms'])"
