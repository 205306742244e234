"""Phase stamps (s_memtime) of work-group 0 of k_proxy_attn from a lab build that exports ptx_lab_dbg (see scratch/README.md):
python scratch/attn_stamp.py B n Lp"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from proxytransformation_amd import _abi
B, n, Lp = (int(x) for x in sys.argv[1:4])
impl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
heads, C = 8, 256
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * n, 3 * C, generator=g).cuda()
pt = torch.randn(B * Lp, C, generator=g).cuda()
out = torch.empty((B * n, C), device="cuda")
lib = _abi.lib()
scratch = torch.zeros(lib.ptx_proxy_attention_scratch_bytes(B, n, Lp, heads, C, impl) // 4, device="cuda")
raw = ctypes.CDLL(_abi.LIB_PATH)
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    lib.ptx_proxy_attention(qkv.data_ptr(), pt.data_ptr(), None, out.data_ptr(), scratch.data_ptr(), B, n, Lp, heads, C, impl, st)
torch.cuda.synchronize()
buf = np.zeros((4, 8, 16), np.uint64)
raw.ptx_lab_dbg(ctypes.c_void_p(buf.ctypes.data))
names = ["start", "kv issued", "proxies staged", "kv stashed", "sync", "stage A done", "merged", "stage B done",
         "partials out", "ticket", "slices merged"]
t0 = buf[:1 if impl != 3 else 4, :, 0].min()
for z in range(1 if impl != 3 else 4):
    print(f"slice {z}: cycles since the first wave's start, per wave:")
    for k, nm in enumerate(names):
        print(f"{nm:>16s} " + " ".join(f"{int(buf[z, w, k]) - int(t0) if buf[z, w, k] else -1:7d}" for w in range(8)))
