"""The linear operator (ptx_linear, PRE's nn.Linear layers) against float64, over the launch regimes of gemm.hip:
latency-regime tiles (fp32 matrix instruction, K sliced across waves), 64x64 tiles with the fp32 instruction, and
64x64 tiles with the operands split three ways into bf16 (K = 128 .. 1024).  The split is meant to be an fp32
product in a different summation order, so the bar is the fp32 bar: |y - y64| <= 4e-6 * sum_k |x_k w_k| elementwise
(fp32 round-off of a 1024-term sum stays well below that), also for operands spanning 2^-20 .. 2^20."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _linear(x, w, b=None, res=None, gelu=0):
    from proxytransformation_amd import _abi
    lib = _abi.lib()
    R, K = x.shape
    N = w.shape[0]
    y = torch.empty((R, N), dtype=torch.float32, device=x.device)
    rc = lib.ptx_linear(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                        res.data_ptr() if res is not None else None, y.data_ptr(), R, N, K, gelu,
                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.ptx_last_error().decode()
    return y


def _operands(R, N, K, seed, wide):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(R, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.1
    if wide:        # per-element power-of-two scales: the three parts of every operand have to carry their share
        x = x * torch.pow(2.0, torch.randint(-20, 21, (R, K), generator=g).float())
        w = w * torch.pow(2.0, torch.randint(-20, 21, (N, K), generator=g).float())
    return x.cuda(), w.cuda()


@pytest.mark.parametrize("R,N,K", [(2048, 768, 128), (2048, 768, 256), (2048, 768, 512), (2048, 1024, 1024), (2048, 512, 2048),   # split, 64x64
                                   (2048, 768, 160), (2000, 700, 256),                                        # fp32 64x64 / ragged
                                   (784, 256, 256), (300, 96, 1024), (64, 64, 64)])                            # latency regime
@pytest.mark.parametrize("wide", [False, True])
def test_linear_matches_float64(R, N, K, wide):
    x, w = _operands(R, N, K, 1000 + R + N + K, wide)
    b = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    y = _linear(x, w, b)
    y64 = x.double() @ w.double().t() + b.double()
    bound = 4e-6 * (x.double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    worst = ((y.double() - y64).abs() / bound).max().item()
    assert worst <= 1.0, f"R={R} N={N} K={K} wide={wide}: error / bound = {worst:.3f}"


def test_split_kernel_epilogues():
    """bias + GELU + residual through the split kernel's epilogue (same code as the fp32 kernel's)."""
    R, N, K = 2048, 768, 256
    x, w = _operands(R, N, K, 7, False)
    b = torch.randn(N, generator=torch.Generator().manual_seed(8)).cuda()
    res = torch.randn(R, N, generator=torch.Generator().manual_seed(9)).cuda()
    y = _linear(x, w, b, res, gelu=1)
    ref = torch.nn.functional.gelu(x.double() @ w.double().t() + b.double()) + res.double()
    err = (y.double() - ref).abs().max().item()
    assert err <= 2e-5, err


def test_split_is_exact_on_bf16_representable_operands():
    """Operands that ARE bf16 values have zero second and third parts: the split kernel then adds exact products in
    fp32 -- with small integers the result is exact."""
    R, N, K = 2048, 768, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-8, 9, (R, K), generator=g).float().cuda()
    w = torch.randint(-8, 9, (N, K), generator=g).float().cuda()
    y = _linear(x, w)
    assert torch.equal(y.double(), x.double() @ w.double().t())
