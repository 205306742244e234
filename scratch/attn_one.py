"""One shape of ptx_proxy_attention in a loop (for rocprofv3 runs): python scratch/attn_one.py B n Lp impl reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxytransformation_amd import _abi
B, n, Lp, impl, reps = (int(x) for x in sys.argv[1:6])
heads, C = 8, 256
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * n, 3 * C, generator=g).cuda()
pt = torch.randn(B * Lp, C, generator=g).cuda()
out = torch.empty((B * n, C), device="cuda")
lib = _abi.lib()
scratch = torch.zeros(lib.ptx_proxy_attention_scratch_bytes(B, n, Lp, heads, C, impl) // 4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.ptx_proxy_attention(qkv.data_ptr(), pt.data_ptr(), None, out.data_ptr(), scratch.data_ptr(), B, n, Lp, heads, C, impl, st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    lib.ptx_proxy_attention(qkv.data_ptr(), pt.data_ptr(), None, out.data_ptr(), scratch.data_ptr(), B, n, Lp, heads, C, impl, st)
b.record()
torch.cuda.synchronize()
print(f"B={B} n={n} Lp={Lp} impl={impl}: {1e3 * a.elapsed_time(b) / reps:.2f} us per call")
