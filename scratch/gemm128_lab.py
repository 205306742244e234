"""r06: k_gemm128x (128 x 128 tiles) against k_gemm64x (64 x 64) through ptx_linear: error vs float64 and time per launch.
usage: python scratch/gemm128_lab.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.getcwd())
from proxytransformation_amd import _abi
lib = _abi.lib()
dev = torch.device("cuda")


def load(path):
    """A lab build of the library (scratch/g128_variants.py), typed like the product's."""
    h = ctypes.CDLL(os.path.abspath(path))
    for name in ("ptx_linear", "ptx_gemm_policy", "ptx_last_error"):
        fn = getattr(h, name)
        fn.restype, fn.argtypes = _abi.SIGNATURES[name]
    return h


def run(R, N, K, policy, reps=100, gelu=0, res=False):
    g = torch.Generator().manual_seed(R + N + K)
    x = torch.randn(R, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev); y = torch.empty(R, N, device=dev)
    r = torch.randn(R, N, generator=g).to(dev) if res else None
    st = torch.cuda.current_stream().cuda_stream
    prev = lib.ptx_gemm_policy(policy)

    def call():
        rc = lib.ptx_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), R, N, K, gelu, st)
        assert rc == 0, lib.ptx_last_error()
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t() + b.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r.double()
    bound = 4e-6 * (x.double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    err = ((y.double() - ref).abs() / bound).max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    lib.ptx_gemm_policy(prev)
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2.0 * R * N * K / us / 1e6, err


if "--libs" in sys.argv:
    libs = sys.argv[sys.argv.index("--libs") + 1:]
    shapes = [(16384, 2048, 512), (16384, 512, 2048), (8192, 1024, 256), (8192, 768, 256), (4146, 1024, 256)]
    base = {shp: run(*shp, 1) for shp in shapes}
    print("product     " + "  ".join(f"{base[s][0]:7.1f} us {base[s][1]:6.1f} TF" for s in shapes))
    for path in libs:
        lib = load(path)
        r = {shp: run(*shp, 1) for shp in shapes}
        assert all(v[2] <= 1.0 for v in r.values()), (path, r)
        print(f"{os.path.basename(path)[9:-3]:12s}" + "  ".join(f"{r[s][0]:7.1f} us {r[s][1]:6.1f} TF" for s in shapes), flush=True)
    lib = _abi.lib()
    again = {shp: run(*shp, 1) for shp in shapes}
    print("product     " + "  ".join(f"{again[s][0]:7.1f} us {again[s][1]:6.1f} TF" for s in shapes))
    sys.exit(0)
shapes = [(8192, 768, 256), (8192, 256, 256), (8192, 1024, 256), (8192, 256, 1024), (16384, 1536, 512), (16384, 2048, 512),
          (16384, 512, 2048), (16384, 512, 512), (4146, 768, 256), (4146, 1024, 256), (4146, 256, 1024), (2048, 1024, 1024),
          (4100, 700, 256)]
for shp in shapes:
    a = run(*shp, 0)
    b_ = run(*shp, 1)
    print(f"R={shp[0]:6d} N={shp[1]:5d} K={shp[2]:5d}:  64x64 {a[0]:8.1f} us {a[1]:7.1f} TF err/bound {a[2]:.3f}   |  128x128 {b_[0]:8.1f} us {b_[1]:7.1f} TF "
          f"err/bound {b_[2]:.3f}   x{a[0] / b_[0]:.2f}", flush=True)
a = run(8192, 768, 256, 1, gelu=1, res=True)
print("gelu+res 128x128:", a)
