#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
ROOT=$(pwd)
export MVAE_LIB=$ROOT/tests/tuning/lib/libmvae_hip_tune.so
for rep in 1 2; do
for mode in 0 1 2 3; do
  for dbg in 0 1; do
    echo "== B=1024 GM_MODE=$mode DBG=$dbg"
    MVAE_GM_MODE=$mode MVAE_DBG=$dbg timeout -k 10 120 python3 tests/bench_kernels.py 24 1024 fwd 2>&1 | grep fwd
  done
done
done
echo "== old kernel (MVAE_FWD_GM=0)"; MVAE_FWD_GM=0 python3 tests/bench_kernels.py 24 1024 fwd 2>&1 | grep fwd
