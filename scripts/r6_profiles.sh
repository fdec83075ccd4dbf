#!/bin/bash
# round-6 profiles of the bench command: rocprofv3 kernel stats of the default bench line, then the two PMC passes (FETCH_SIZE,
# WRITE_SIZE -- separate passes, no other trace domain) of a short run for the dominant kernel's HBM traffic
set -o pipefail
out=$PWD/gpurun_out/r6p; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err ) || { tail -20 $out/bench_prof.err; exit 1; }
python3 scripts/prof_summary.py $out/prof $out/bench_kernel_stats.txt "python3 bench.py --no-cpu-baseline (default --steps 3 --warmup 1; all legs)" | head -12
echo "kernel stats done" 
PMC_TIMEOUT=400 bash scripts/pmc_pass.sh $out/pmc_fetch "FETCH_SIZE" -- python3 bench.py --no-cpu-baseline --no-fast-mode --steps 1 --warmup 1 || exit 1
PMC_TIMEOUT=400 bash scripts/pmc_pass.sh $out/pmc_write "WRITE_SIZE" -- python3 bench.py --no-cpu-baseline --no-fast-mode --steps 1 --warmup 1 || exit 1
python3 scripts/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic_w1.json "python3 bench.py --no-cpu-baseline --no-fast-mode --steps 1 --warmup 1" | head -12
# the surface sampler's per-kernel split (fp32 screen, fp64 mean of the candidates, variance of the survivors)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_surface -- python3 $GRAFT_REPO_ROOT/scripts/surface_profile.py > $out/surface.txt 2> $out/surface.err ) || { tail -20 $out/surface.err; exit 1; }
python3 scripts/prof_summary.py $out/prof_surface $out/surface_kernel_stats.txt "python3 scripts/surface_profile.py (C3 model, gpx_model_sample_surface over the 128^3 lattice)" | head -14
# the driver's own line, un-profiled, last (what BENCH_r06 will look like)
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
tail -c 1500 $out/bench.json
