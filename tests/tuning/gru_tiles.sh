export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
export BK_CELL=gru BK_H=512 BK_NL=3
echo "== default"; python tests/bench_kernels.py 32 1024 fwd,bwd 2>&1 | grep -v amdgpu
for bm in 64 128; do for bj in 32 64; do for nb in 2 3 4; do
echo "== fwd BM=$bm BJ=$bj NBUF=$nb"; MVAE_BM=$bm MVAE_BJ=$bj MVAE_NBUF_FWD=$nb python tests/bench_kernels.py 32 1024 fwd 2>&1 | grep -v amdgpu
done; done; done
for bm in 64 128; do for nb in 3 4 5; do
echo "== bwd fused BM=$bm NBUF=$nb"; MVAE_BWD_SPLIT=0 MVAE_BM=$bm MVAE_NBUF_BWD=$nb python tests/bench_kernels.py 32 1024 bwd 2>&1 | grep -v amdgpu
done; done
for sp in 2 1284 644 2562; do echo "== bwd split $sp"; MVAE_BWD_SPLIT=$sp python tests/bench_kernels.py 32 1024 bwd 2>&1 | grep -v amdgpu; done
