"""The linear operator (ptx_linear, PRE's nn.Linear layers) against float64, over the launch regimes of gemm.hip:
latency-regime tiles (fp32 matrix instruction, K sliced across waves), 64x64 tiles with the fp32 instruction,
64x64 tiles with the operands split three ways into bf16 (K = 128 .. 1024) and -- r06 -- 128x128 tiles with the same
split (k_gemm128x: K a multiple of 256, chosen from 256 tiles per launch on; forced here through ptx_gemm_policy).  The split is meant to be an fp32
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


@pytest.fixture
def force_128_tiles():
    from proxytransformation_amd import _abi
    lib = _abi.lib()
    prev = lib.ptx_gemm_policy(1)                # whenever the shape allows it
    yield
    lib.ptx_gemm_policy(prev)


@pytest.mark.parametrize("R,N,K", [(2048, 768, 256), (2048, 768, 512), (2048, 1024, 1024), (1024, 512, 2048), (1024, 256, 4096),
                                   (2000, 700, 256), (4100, 130, 768), (129, 129, 256)])      # ragged rows / columns, K = 3 C
@pytest.mark.parametrize("wide", [False, True])
def test_128_tile_kernel_matches_float64(force_128_tiles, R, N, K, wide):
    """k_gemm128x at the fp32 bar of the 64x64 split kernel: |y - y64| <= 4e-6 sum_k |x_k w_k| elementwise, also for operands spanning
    2^-20 .. 2^20, ragged last row / column tiles included (clamped loads, guarded stores)."""
    x, w = _operands(R, N, K, 2000 + R + N + K, wide)
    b = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    y = _linear(x, w, b)
    y64 = x.double() @ w.double().t() + b.double()
    bound = 4e-6 * (x.double().abs() @ w.double().abs().t() + b.double().abs()) + 1e-30
    worst = ((y.double() - y64).abs() / bound).max().item()
    assert worst <= 1.0, f"R={R} N={N} K={K} wide={wide}: error / bound = {worst:.3f}"


def test_128_tile_kernel_epilogues_and_exactness(force_128_tiles):
    R, N, K = 2048, 768, 256
    x, w = _operands(R, N, K, 7, False)
    b = torch.randn(N, generator=torch.Generator().manual_seed(8)).cuda()
    res = torch.randn(R, N, generator=torch.Generator().manual_seed(9)).cuda()
    y = _linear(x, w, b, res, gelu=1)
    ref = torch.nn.functional.gelu(x.double() @ w.double().t() + b.double()) + res.double()
    assert (y.double() - ref).abs().max().item() <= 2e-5
    # bf16-representable operands: the second and third parts are zero, the products are exact
    g = torch.Generator().manual_seed(3)
    xi = torch.randint(-8, 9, (R, K), generator=g).float().cuda()
    wi = torch.randint(-8, 9, (N, K), generator=g).float().cuda()
    assert torch.equal(_linear(xi, wi).double(), xi.double() @ wi.double().t())


def test_default_policy_picks_the_tile_by_launch_size():
    """256 tiles of 128 x 128 per launch is where the large tile starts (one per CU).  Both kernels add the same six bf16 products per
    16 k in the same order (small terms first, k ascending) into one fp32 accumulator per output: their results are BIT-IDENTICAL, so
    the tile policy can never change a forward's output."""
    from proxytransformation_amd import _abi
    lib = _abi.lib()
    assert lib.ptx_gemm_policy(-1) == 256
    x, w = _operands(8192, 768, 256, 11, False)          # 64 x 6 = 384 tiles = 1.5 per CU: the 128 x 128 kernel
    y_big = _linear(x, w)
    prev = lib.ptx_gemm_policy(0)
    try:
        y_small = _linear(x, w)
    finally:
        lib.ptx_gemm_policy(prev)
    y64 = x.double() @ w.double().t()
    bound = 4e-6 * (x.double().abs() @ w.double().abs().t()) + 1e-30
    assert ((y_big.double() - y64).abs() / bound).max().item() <= 1.0
    assert ((y_small.double() - y64).abs() / bound).max().item() <= 1.0
    assert torch.equal(y_big, y_small)


def test_forward_with_every_legal_gemm_on_128_tiles(force_128_tiles):
    """The eval forward with k_gemm128x forced wherever K is a multiple of 256 -- qkv, proxy_proj with the folded norm_img
    (LayerNorm-consumer epilogue), proj with its LayerNorm partials (producer epilogue), the image chain's table GEMMs -- against
    the oracle at the usual bars (indices bit-identical, coordinates within 1e-4)."""
    import numpy as np
    from oracle import oracle
    from proxytransformation_amd.synth import PreshapeConfig, make_scene_batch
    from tests.gpu_util import t
    from tests.util import assert_close, build_module, oracle_kwargs
    cfg = PreshapeConfig("g128", B=3, N=12000, grid_size=6, dynamic_drop_radio=0.5, L=12, V=9, seed_base=6100)
    m, sd = build_module(cfg)
    m = m.cuda()
    pts, text, mask, img = make_scene_batch(cfg)
    ref = oracle.forward(sd, **oracle_kwargs(cfg), points=pts, text_feats=text, text_mask=mask, img_feat=img, num_threads=8)
    m._centers_override = torch.from_numpy(ref["centers"])
    d = m.forward_debug([t(p) for p in pts], {"text_feats": t(text), "text_token_mask": t(mask)}, t(img))
    for k in ("idx2", "order", "picks", "keep", "kidx", "drop_idx"):
        assert np.array_equal(d[k].cpu().numpy().astype(np.int64), ref[k]), k
    assert_close(d["img_proxy"].cpu().numpy(), ref["img_proxy"], atol=5e-5, rtol=1e-5, what="img_proxy")
    assert_close(d["translate"].cpu().numpy(), ref["translate"], atol=5e-5, rtol=1e-5, what="translate")
    assert_close(d["transform"].cpu().numpy(), ref["transform"], atol=5e-5, rtol=1e-5, what="transform")
    for b in range(cfg.B):
        assert_close(d["outputs"][b].cpu().numpy(), ref["outputs"][b], atol=1e-4, what=f"scene {b}")
