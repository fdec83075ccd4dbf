#!/bin/bash
# HBM traffic of the headline variance kernel as built at the end of round 4: FETCH_SIZE and WRITE_SIZE in separate passes of the same bench command
set -o pipefail
out=$PWD/gpurun_out/r4p; mkdir -p $out
cmd="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fast-mode --no-configs"
PMC_TIMEOUT=400 bash scripts/pmc_pass.sh $out/fetch "FETCH_SIZE" -- $cmd || exit 1
PMC_TIMEOUT=400 bash scripts/pmc_pass.sh $out/write "WRITE_SIZE" -- $cmd || exit 1
python3 scripts/pmc_traffic.py $out/fetch $out/write $out/r04_pmc_traffic_w1.json $cmd | grep -i "var_w1\|kbuild\|kqp" 
