import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proxytransformation_amd import train as T
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (M, N, K) in [(4146, 256, 256), (4146, 1024, 256), (4146, 256, 1024), (4146, 768, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); dy = torch.randn(M, N, device=dev)
    print(f"M={M} N={N} K={K}: fwd NT {timeit(lambda: T.mm(x, w, tb=True)):.1f} us | dx NN {timeit(lambda: T.mm(dy, w)):.1f} us | dW TN {timeit(lambda: T.mm(dy, x, ta=True)):.1f} us | colsum {timeit(lambda: T.colsum(dy)):.1f} us")
x = torch.randn(124380, 256, device=dev)
print("colsum 124k x 256:", timeit(lambda: T.colsum(x)), "us;  mode2:", timeit(lambda: T.colsum(x, mode=2)))
