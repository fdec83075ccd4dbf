#!/bin/bash
# round-4 first GPU call: Eigen probe on the box + baseline of the small-model variance path
set -o pipefail
mkdir -p gpurun_out/r4a
{
  echo "== eigen probe"
  for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/rocm/include/eigen3 /usr/include/Eigen; do
    [ -e "$d" ] && echo "FOUND $d"
  done
  find / -xdev \( -name 'LDLT.h' -o -name 'signature_of_eigen3_matrix_library' \) 2>/dev/null | head -20
  echo '#include <Eigen/Dense>' > /tmp/e.cpp; echo 'int main(){return EIGEN_WORLD_VERSION;}' >> /tmp/e.cpp
  g++ -I/usr/include/eigen3 /tmp/e.cpp -o /tmp/e 2>&1 | head -3
  python3 -c "import numpy, sys; print('numpy', numpy.__version__)"
  echo "== nproc $(nproc)"
} > gpurun_out/r4a/eigen_probe.txt 2>&1
python3 scripts/var_tile_sweep.py "one-wave(r3)" > gpurun_out/r4a/sweep.txt 2>&1 || exit 1
for o in bowlA bowlB containerA containerB jug kettle pot mugD; do
  python3 scripts/c5_stages.py $o gaussian 128 >> gpurun_out/r4a/c5_stages.txt 2>&1 || exit 1
done
