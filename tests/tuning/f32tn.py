import os, sys, torch
sys.path.insert(0, os.getcwd())
import molecular_vae_amd as mv
from molecular_vae_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (M, N, K) in ((4096, 1024, 122880), (288, 72, 122880), (4096, 1024, 15360), (288, 72, 15360), (512, 1344, 1024), (120, 2304, 56320)):
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev); cs = torch.empty(M, device=dev)
    t0 = timeit(lambda: ops.gemm_tn(A, B, C, M, N, K))
    t1 = timeit(lambda: ops.gemm_tn_f32_colsum(A, B, C, cs, M, N, K))
    print(f"M={M} N={N} K={K}: plain {t0:.3f} ms, with colsum {t1:.3f} ms", flush=True)
    del A, B
