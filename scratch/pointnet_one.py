"""ptx_pointnet alone, 30 launches at cfg4's shape (for rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from proxytransformation_amd.synth import CONFIGS
from tests.util import build_module
from tests.gpu_util import Stages
cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mod = build_module(cfg)[0].cuda()
st = Stages(mod, B, cfg.N, cfg.L, cfg.V)
s = st.shape
kc = torch.randn(B, s.Mk, 3, device="cuda"); kcl = torch.randn(B, s.Mk, s.K, 3, device="cuda")
mode = sys.argv[3] if len(sys.argv) > 3 else "hot"
junk = torch.empty(200 << 20, dtype=torch.uint8, device="cuda")
a = torch.randn(512, 512, device="cuda"); v = torch.randn(100000, device="cuda")
for i in range(30):
    if mode in ("data", "both"):
        junk.add_(1)
    if mode in ("code", "both"):        # many different kernels: evict the instruction caches
        b = a @ a; b = torch.softmax(b, 1); b = torch.nn.functional.layer_norm(b, (512,)); c = torch.sort(v)[0]
        c = torch.cumsum(c, 0); d = torch.tanh(b) + torch.erf(b) * torch.sigmoid(b); e = d.t().contiguous().sum(0)
        f = torch.nn.functional.gelu(d).max(1)[0]; g = torch.topk(v, 64)[0]; h = (a > 0).float().mean()
    st.pointnet(kc, kcl)
torch.cuda.synchronize()
print("Mk", s.Mk, "K", s.K)
