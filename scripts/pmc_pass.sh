#!/bin/bash
# One rocprofv3 counter pass, bounded and loud:  scripts/pmc_pass.sh OUTDIR "CTR1 CTR2" -- python3 bench.py ...
#   * at most 4 counters per pass (TCC: FETCH_SIZE costs 3 of 4 slots, WRITE_SIZE 2 -- one per pass; an over-subscribed
#     list makes rocprofv3 abort with "exceeds the capabilities of the hardware" and the profiled process then lingers:
#     round 1 burnt five metered minutes that way, gpurun_out/pmc_inv.log);
#   * --kernel-trace is always on (kernel names in the counter CSV), never a sys/hip/hsa trace next to --pmc;
#   * the program itself follows "--" (no env / bash -c hop: the profiler's library initialises the GPU first);
#   * the whole pass runs under `timeout -k 10`; an abort, a fault or a timeout gives a non-zero exit code.
set -u
out=$1; ctrs=$2; shift 2
[ "$1" = "--" ] && shift
n=$(echo $ctrs | wc -w)
if [ "$n" -gt 4 ]; then echo "pmc_pass: $n counters in one pass (max 4): split the list" >&2; exit 2; fi
limit=${PMC_TIMEOUT:-300}
mkdir -p "$out"
out=$(realpath "$out")   # rocprofv3 runs from /tmp below: a relative OUTDIR would land there ...
args=()                   # ... and so would a relative program / script path: make every argument that names a file absolute
for a in "$@"; do if [ -e "$a" ]; then args+=("$(realpath "$a")"); else args+=("$a"); fi; done
set -- "${args[@]}"
log="$out/pass.log"
( cd /tmp && TMPDIR=/tmp timeout -k 10 "$limit" rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -- "$@" ) > "$log" 2>&1
rc=$?
if grep -q "exceeds the capabilities\|Could not construct profile\|Memory access fault\|caught signal" "$log"; then
    echo "pmc_pass: rocprofv3 aborted -- see $log" >&2; grep -m3 "exceeds\|Could not\|fault\|signal" "$log" >&2; exit 3
fi
if [ $rc -ne 0 ]; then echo "pmc_pass: exit code $rc (124 = timeout after ${limit}s) -- see $log" >&2; exit $rc; fi
echo "pmc_pass: ok ($ctrs) -> $out"
