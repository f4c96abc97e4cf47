#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
# Main-loop / epilogue decomposition of the decoder wavefront step kernels with the DIAGNOSTIC library (csrc/build.sh tune;
# MVAE_DBG=1: main loop only, =2: epilogue only -- results wrong, timings only).  Run on the GPU box from the repo root:
#   bash tests/decompose.sh "512 1024 128" -> gpurun_out/decomp.txt
set -uo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/decomp.txt
mkdir -p $ROOT/gpurun_out
: > $OUT
export MVAE_LIB=$ROOT/tests/tuning/lib/libmvae_hip_tune.so
for B in ${1:-512}; do
  for dbg in 0 1 2; do
    echo "== B=$B MVAE_DBG=$dbg (0 = full kernel, 1 = main loop only, 2 = epilogue only)" >> $OUT
    MVAE_DBG=$dbg timeout -k 10 300 python3 tests/bench_kernels.py 24 $B fwd,bwd >> $OUT 2>&1 || exit 1
  done
done
cat $OUT
