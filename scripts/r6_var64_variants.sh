#!/bin/bash
# round 6: variants of the small fp64 kernel (make EXTRA=... OUTDIR=../lib_<name>): scripts/var64_two.py on each; arguments: library names
mkdir -p gpurun_out/r6f
out=gpurun_out/r6f/variants_$1.txt
: > $out
for v in "$@"; do
  echo "== lib_$v" >> $out
  GPX_LIB=$PWD/gaussian-object-modelling_amd/lib_$v/libgpx.so timeout -k 10 200 python scripts/var64_two.py 277 512 724 >> $out 2>&1 || exit 1
done
cat $out
