#!/bin/bash
# PMC passes over the kernel micro-benchmark (one counter group per pass: gfx950 cannot schedule FETCH_SIZE and WRITE_SIZE together).
# Run on the GPU box from the repo root:  bash tests/collect_pmc.sh [T] [B] [modes]   ->  gpurun_out/pmc/<group>/*counter_collection.csv
set -uo pipefail
T=${1:-16}; B=${2:-512}; MODES=${3:-fwd,bwd,gemm}
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
run() {   # name counters...
  local name=$1; shift
  echo "[pmc] pass $name: $*"
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/gpurun_out/pmc/$name" -o "$name" -- python3 "$ROOT/tests/bench_kernels.py" "$T" "$B" "$MODES" > "$ROOT/gpurun_out/pmc_$name.log" 2>&1
  local rc=$?
  echo "[pmc] pass $name rc=$rc"
  return $rc
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run tcc TCC_HIT_sum TCC_MISS_sum && \
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
