#!/bin/bash
# round-4 closing GPU call: suite, bench line, rocprofv3 kernel stats of the same bench command
set -o pipefail
out=$PWD/gpurun_out/r4z; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { grep -v "^  File" $out/pytest.txt | tail -40; exit 1; }
tail -3 $out/pytest.txt
timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err ) || { tail -20 $out/bench_prof.err; exit 1; }
python3 scripts/prof_summary.py $out/prof $out/bench_kernel_stats.txt "python3 bench.py --no-cpu-baseline (default --steps 3 --warmup 1; all legs)" | head -25
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4z/bench.json').read().strip().splitlines()[-1])
print("ms_per_step",d["ms_per_step"],"value",d["value"],"frac",d["roofline"]["frac"])
print({k:(round(v["kernel_ms"],3),round(v["frac"],3),round(v["variance_stage_ms"],3)) for k,v in d["roofline_small"]["sizes"].items()})
for k,v in d.get("configs",{}).items(): print(k,{kk:vv for kk,vv in v.items() if kk in ("ms_per_step","ms","survivors","error")})
print("kbuild",d["roofline_kbuild"]["frac"],"kqp",d["roofline_kqp"]["frac"])
PY
