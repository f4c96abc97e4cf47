#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
# product library; schedule knobs only
for B in 128 256 512 1024; do
  for gm in 0 256256 256128 128128 128064; do
    r=$(MVAE_FWD_GM=$gm timeout -k 10 120 python3 tests/bench_kernels.py 24 $B fwd 2>&1 | grep "fwd :")
    echo "B=$B FWD_GM=$gm $r"
  done
  for sp in 0 2 2562 1284 644; do
    r=$(MVAE_BWD_SPLIT=$sp timeout -k 10 120 python3 tests/bench_kernels.py 24 $B bwd 2>&1 | grep "bwd :")
    echo "B=$B BWD_SPLIT=$sp $r"
  done
done
