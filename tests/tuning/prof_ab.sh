cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in prev new; do
  f=molecular-vae_amd/libmvae_hip.so; [ $lib = prev ] && f=molecular-vae_amd/libmvae_hip_prev.so
  export MVAE_LIB=$GRAFT_REPO_ROOT/$f
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ab_$lib -o p --output-format csv -- python3 tests/bench_kernels.py 48 128 bwd > /dev/null 2>&1
  echo "== $lib"; python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_ab_$lib/p_kernel_stats.csv")))
for r in rows[:4]: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
done
