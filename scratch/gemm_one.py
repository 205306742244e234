"""One shape of the tuned linear kernel, 30 launches (for rocprofv3 --kernel-trace --stats)."""
import sys, torch
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from proxytransformation_amd import _abi
lib = _abi.lib()
if os.environ.get('GEMM_POLICY'):
    lib.ptx_gemm_policy(int(os.environ['GEMM_POLICY']))
R, N, K, g = map(int, sys.argv[1:5])
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
y = torch.empty(R, N, device=dev)
for i in range(30):
    lib.ptx_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), R, N, K, g, st)
torch.cuda.synchronize()
