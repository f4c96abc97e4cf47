#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
ROOT=$(pwd)
echo "== LSTM B=1024 bwd candidates (product lib)"
for sp in 2562 1281 2; do r=$(MVAE_BWD_SPLIT=$sp python3 tests/bench_kernels.py 24 1024 bwd 2>&1 | grep "bwd :"); echo "BWD_SPLIT=$sp $r"; done
echo "== LSTM B=512 bwd 1281"
for sp in 2 1281; do r=$(MVAE_BWD_SPLIT=$sp python3 tests/bench_kernels.py 24 512 bwd 2>&1 | grep "bwd :"); echo "BWD_SPLIT=$sp $r"; done
echo "== GRU H=512 NL=3 B=1024 / 128 (models2d / moses decoder shape)"
export BK_CELL=gru BK_H=512 BK_NL=3
for B in 1024 128; do
  for sp in 1 0 2 1284 644 1281; do r=$(MVAE_BWD_SPLIT=$sp python3 tests/bench_kernels.py 24 $B bwd 2>&1 | grep "bwd :"); echo "B=$B BWD_SPLIT=$sp $r"; done
  r=$(python3 tests/bench_kernels.py 24 $B fwd 2>&1 | grep "fwd :"); echo "B=$B $r"
done
export MVAE_LIB=$ROOT/tests/tuning/lib/libmvae_hip_tune.so
for dbg in 0 1 2; do r=$(MVAE_DBG=$dbg python3 tests/bench_kernels.py 24 1024 fwd,bwd 2>&1 | grep "wd :" | tr '\n' ' '); echo "GRU B=1024 DBG=$dbg $r"; done
