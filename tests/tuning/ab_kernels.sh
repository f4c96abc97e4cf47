#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
# Same-box A/B of two builds on the kernel micro-benchmark (us per wavefront launch):  bash tests/tuning/ab_kernels.sh libA.so libB.so B [modes] [rounds]
set -uo pipefail
A=$1; B=$2; BATCH=${3:-1024}; MODES=${4:-fwd}; R=${5:-4}
for r in $(seq $R); do
  for lib in $A $B; do
    echo -n "$(basename $lib): "
    MVAE_LIB=$(pwd)/$lib timeout -k 10 300 python3 tests/bench_kernels.py 48 $BATCH $MODES 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
  done
done
