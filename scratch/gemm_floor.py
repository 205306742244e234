"""Hot and cold timing of the tuned linear kernel at the proxy-block shapes (R rows stands for 2 branches x 1024)."""
import sys, torch
sys.path.insert(0, '.')
from proxytransformation_amd import _abi
lib = _abi.lib()
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
def run(R, N, K, gelu, cold, iters=50):
    x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(R, N, device=dev)
    junk = torch.empty(192 << 20, dtype=torch.uint8, device=dev)
    evs = []
    for i in range(iters + 5):
        if cold: junk.add_(1)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lib.ptx_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), R, N, K, gelu, st); e.record()
        evs.append((a, e))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(e) * 1e3 for a, e in evs[5:])
    return t[len(t) // 2]
for name, R, N, K, g in [("fc1", 2048, 1024, 256, 1), ("fc2", 2048, 256, 1024, 0), ("proj", 2048, 256, 256, 0),
                         ("qkv", 2048, 768, 256, 0), ("c_proj", 784, 256, 256, 0), ("tiny", 64, 64, 64, 0)]:
    print(name, R, N, K, "hot %.1f us  cold %.1f us" % (run(R, N, K, g, False), run(R, N, K, g, True)))
