#!/bin/bash
# round-4 GPU call: the whole -m gpu suite, then the small-model probes
set -o pipefail
out=gpurun_out/r4s; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
python3 scripts/vc_probe.py 277 512 724 1024 2>&1 | grep -v amdgpu.ids | tee $out/probe.txt
for o in bowlA bowlB containerA containerB jug kettle pot mugD; do
  python3 scripts/c5_stages.py $o gaussian 128 2>&1 | grep F32 >> $out/c5_stages.txt || exit 1
done
cat $out/c5_stages.txt
