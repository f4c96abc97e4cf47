#!/bin/bash
# Build libmvae_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# `build.sh tune` builds the diagnostic variant libmvae_hip_tune.so (-DMVAE_TUNING: MVAE_DBG can skip the main loop / the epilogue of
# the step kernels -- wrong results, timing decompositions only).  The product library has no such hook.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
if [ "${1:-}" = "tune" ]; then
  mkdir -p build/tune ../../tests/tuning/lib
  pids=()
  for f in gemm rnn rnn_rowres rnn_persist rnn_persist_bwd elementwise conv capi; do
    $HIPCC $FLAGS -DMVAE_TUNING -c $f.hip -o build/tune/$f.o &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tests/tuning/lib/libmvae_hip_tune.so build/tune/gemm.o build/tune/rnn.o build/tune/rnn_rowres.o build/tune/rnn_persist.o build/tune/rnn_persist_bwd.o build/tune/elementwise.o build/tune/conv.o build/tune/capi.o
  echo "built $(cd ../../tests/tuning/lib && pwd)/libmvae_hip_tune.so"
  exit 0
fi
pids=()
for f in gemm rnn rnn_rowres rnn_persist rnn_persist_bwd elementwise conv capi; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.hpp -nt build/$f.o ] || [ tile.hpp -nt build/$f.o ] || [ tile_pipe.hpp -nt build/$f.o ] || [ kernels.hpp -nt build/$f.o ] || [ persist_common.hpp -nt build/$f.o ] || [ ../../include/mvae.h -nt build/$f.o ]; then
    # -Rpass-analysis: per-kernel VGPR / scratch / spill figures go to build/$f.usage.txt (tests assert that no kernel uses scratch)
    $HIPCC $FLAGS -Rpass-analysis=kernel-resource-usage -c $f.hip -o build/$f.o 2> build/$f.usage.txt &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait $p || rc=1; }; done
if [ $rc -ne 0 ]; then grep -h -B2 -A6 "error" build/*.usage.txt | head -60; exit 1; fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libmvae_hip.so build/gemm.o build/rnn.o build/rnn_rowres.o build/rnn_persist.o build/rnn_persist_bwd.o build/elementwise.o build/conv.o build/capi.o
echo "built $(cd .. && pwd)/libmvae_hip.so"
