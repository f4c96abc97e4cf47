#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
# Forward gate-major epilogue with the DIAGNOSTIC library: is its exposed time store ISSUE or HBM write bandwidth?
#   MVAE_DBG=2: epilogue only;  6: + every store into the tile's first 16 rows (same instructions, 1/16 of the write footprint);
#   10: + saved gates not stored (half the bytes);  0 / 4 / 8: the same three with the main loop in front.
set -uo pipefail
ROOT=$(pwd)
export MVAE_LIB=$ROOT/tests/tuning/lib/libmvae_hip_tune.so
for dbg in 0 4 8 2 6 10; do
  echo "== B=1024 MVAE_DBG=$dbg"
  MVAE_DBG=$dbg timeout -k 10 300 python3 tests/bench_kernels.py 24 1024 fwd 2>&1 | grep -v amdgpu.ids || exit 1
done
