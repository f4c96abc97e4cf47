import torch, time
dev = torch.device("cuda")
def bench(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for mb in (33.5, 67, 134):
    n = int(mb * 1e6 / 2)
    bufs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(12)]
    src = [torch.randn(n, device=dev).bfloat16() for _ in range(4)]
    i = [0]
    def fill():
        bufs[i[0] % 12].zero_(); i[0] += 1
    def copy():
        bufs[i[0] % 12].copy_(src[i[0] % 4]); i[0] += 1
    us = bench(fill); print(f"{mb} MB zero_: {us:.1f} us  {mb / us * 1e-6 * 1e6 / 1e6:.2f} TB/s")
    us = bench(copy); print(f"{mb} MB copy_: {us:.1f} us  write {mb / us:.2f} TB/s (+ same read)")
