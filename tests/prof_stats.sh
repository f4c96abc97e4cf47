#!/bin/bash
# rocprofv3 kernel statistics of a bench.py run (run on the GPU box from the repo root):
#   bash tests/prof_stats.sh <name> [bench.py args...]   ->  gpurun_out/prof_<name>/ + gpurun_out/<name>_kernel_stats.csv
set -uo pipefail
ROOT=$(pwd)
NAME=$1; shift
export TMPDIR=/tmp
cd /tmp
rm -rf "$ROOT/gpurun_out/prof_$NAME"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$NAME" -o "$NAME" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary "$@" > "$ROOT/gpurun_out/prof_$NAME.log" 2>&1
rc=$?
f=$(ls "$ROOT"/gpurun_out/prof_$NAME/*/*kernel_stats.csv "$ROOT"/gpurun_out/prof_$NAME/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$ROOT/gpurun_out/${NAME}_kernel_stats.csv"
tail -2 "$ROOT/gpurun_out/prof_$NAME.log"
echo "rc=$rc stats=$f"
