"""Measures the achievable fp64 FMA rate and HBM copy bandwidth on this GPU
(development aid: sanity-checks the peaks used in bench.py's roofline)."""
import time
import torch
torch.cuda.set_device(0)
n = 1 << 28
a = torch.empty(n, dtype=torch.float64, device='cuda').normal_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print('copy: %.2f TB/s (read+write)' % (2 * n * 8 / dt / 1e12))
# fp64 GEMM through the library as an fp64 peak proxy
m = 8192
x = torch.randn(m, m, dtype=torch.float64, device='cuda')
y = torch.randn(m, m, dtype=torch.float64, device='cuda')
for _ in range(2):
    z = x @ y
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    z = x @ y
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print('dgemm %d: %.1f TFLOP/s' % (m, 2 * m**3 / dt / 1e12))
