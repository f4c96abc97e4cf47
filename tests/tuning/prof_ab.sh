#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
# Per-kernel average durations of two library builds on ONE box: rocprofv3 --kernel-trace --stats over the kernel micro-benchmark
# (bench.py refuses MVAE_* variables, so the library is switched under tests/bench_kernels.py):
#   bash tests/tuning/prof_ab.sh libA.so libB.so [T] [B] [modes]
A=$1; B_=$2; T=${3:-48}; BATCH=${4:-128}; MODES=${5:-bwd}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in $A $B_; do
  export MVAE_LIB=$GRAFT_REPO_ROOT/$lib
  d=gpurun_out/prof_ab_$(basename $lib .so)
  rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python3 tests/bench_kernels.py $T $BATCH $MODES > /dev/null 2>&1
  echo "== $(basename $lib)"
  python3 - "$d/p_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print(f"  {r['Name'][:80]:80s} x{r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
done
