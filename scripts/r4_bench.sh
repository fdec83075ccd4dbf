#!/bin/bash
set -o pipefail
out=$PWD/gpurun_out/r4b; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "small_model or forced_fp32 or ragged or c5_objects or fuzz" > $out/pytest_subset.txt 2>&1 || { grep -v "^  File" $out/pytest_subset.txt | tail -30; exit 1; }
tail -2 $out/pytest_subset.txt
timeout -k 10 900 python3 bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/bench.json').read().strip().splitlines()[-1])
print("ms_per_step",d["ms_per_step"],"value",d["value"])
print("roofline",{k:d["roofline"][k] for k in ("achieved","frac","avg_launch_ms")})
print("roofline_small",json.dumps(d.get("roofline_small",{}).get("sizes"),indent=0))
for k,v in d.get("configs",{}).items(): print(k,{kk:vv for kk,vv in v.items() if kk in ("ms_per_step","ms","survivors","error","value","full_variance_equivalent_ms","ms_per_object")})
print("eigen_on_box",d.get("eigen_on_box"))
print("fast_mode",d.get("fast_mode",{}).get("ms_per_step"),"f64",d.get("f64",{}).get("ms_per_step"))
print("cpu_baseline",d["cpu_baseline"]["value"],d["cpu_baseline"].get("full_size_measured",{}).get("value"))
PY
