"""The two attention products of ProxyAttention (PRE:230-250) through the C ABI (ptx_proxy_attention): the fused
single-launch kernel (csrc/fattn.hip, split-operand bf16 matrix pipe) and the two-launch fp32 kernel (csrc/attn.hip)
against a float64 evaluation of the reference's formulas -- ragged proxy / token counts, padded text tokens, both the
register-resident (n <= 256) and the streaming key-tile paths."""
import numpy as np
import pytest
import torch

from proxytransformation_amd import _abi

pytestmark = pytest.mark.gpu


def _ref(qkv, pt, mask, B, n, Lp, heads, C):
    """PRE:225-252 in float64."""
    hd = C // heads
    scale = hd ** -0.5
    x = qkv.double().view(B, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]                                              # (B,h,n,hd)
    p = pt.double().view(B, Lp, heads, hd).permute(0, 2, 1, 3)              # (B,h,Lp,hd)
    a1 = ((p * scale) @ k.transpose(-2, -1)).softmax(-1)                    # PRE:232-236 (no mask)
    pv = a1 @ v                                                             # (B,h,Lp,hd)
    a2 = (q * scale) @ p.transpose(-2, -1)                                  # (B,h,n,Lp)
    if mask is not None:
        a2 = a2.masked_fill(mask.view(B, 1, 1, Lp) == 0, -1e9)              # PRE:245-247
    out = a2.softmax(-1) @ pv
    return out.transpose(1, 2).reshape(B * n, C)                            # PRE:252


@pytest.mark.parametrize("impl", [1, 2, 3], ids=["fused", "two-launch", "fused-split"])
@pytest.mark.parametrize("B,n,Lp,masked", [(2, 256, 196, False), (3, 256, 64, True), (1, 64, 16, True), (2, 100, 37, True),
                                           (1, 691, 50, False), (2, 691, 77, True), (1, 300, 256, False), (1, 1024, 5, True)])
def test_proxy_attention_matches_float64(impl, B, n, Lp, masked):
    heads, C = 8, 256
    g = torch.Generator().manual_seed(1000 * n + Lp)
    qkv = torch.randn(B * n, 3 * C, generator=g).cuda()
    pt = (torch.randn(B * Lp, C, generator=g) * 1.5).cuda()
    mask = None
    if masked:
        mask = (torch.rand(B, Lp, generator=g) > 0.3).to(torch.uint8)
        mask[:, 0] = 1
        if B > 1:
            mask[1, :] = 0                                                   # a scene with every token padded: uniform weights
        mask = mask.cuda()
    out = torch.full((B * n, C), float("nan"), device="cuda")
    nbytes = _abi.lib().ptx_proxy_attention_scratch_bytes(B, n, Lp, heads, C, impl)
    assert nbytes >= (4 * B * Lp * C if impl != 3 else 4 * B * heads * 4 * n * 34)
    scratch = torch.full((nbytes // 4,), float("nan"), device="cuda")
    _abi.check(_abi.lib().ptx_proxy_attention(qkv.data_ptr(), pt.data_ptr(), None if mask is None else mask.data_ptr(),
                                              out.data_ptr(), scratch.data_ptr(), B, n, Lp, heads, C, impl,
                                              torch.cuda.current_stream().cuda_stream), "ptx_proxy_attention")
    ref = _ref(qkv, pt, mask, B, n, Lp, heads, C)
    err = (out.double() - ref).abs().max().item()
    assert torch.isfinite(out).all()
    assert err < 2e-5, f"max abs error {err:.3e} against the float64 reference (outputs are O(1))"


def test_fused_and_two_launch_forms_agree_on_large_scores():
    """Scores of a few hundred (sharp soft-max): both forms must still agree to fp32 level."""
    B, n, Lp, heads, C = 1, 256, 96, 8, 256
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * n, 3 * C, generator=g) * 6).cuda()
    pt = (torch.randn(B * Lp, C, generator=g) * 6).cuda()
    outs = []
    for impl in (1, 2, 3):
        out = torch.empty((B * n, C), device="cuda")
        scratch = torch.empty(_abi.lib().ptx_proxy_attention_scratch_bytes(B, n, Lp, heads, C, impl) // 4, device="cuda")
        _abi.check(_abi.lib().ptx_proxy_attention(qkv.data_ptr(), pt.data_ptr(), None, out.data_ptr(), scratch.data_ptr(),
                                                  B, n, Lp, heads, C, impl, torch.cuda.current_stream().cuda_stream), "attn")
        outs.append(out)
    ref = _ref(qkv, pt, None, B, n, Lp, heads, C)
    for o in outs:
        assert (o.double() - ref).abs().max().item() < 2e-4 * ref.abs().max().item()


def test_split_form_leaves_its_tickets_zero_and_repeats_bit_for_bit():
    """The forward reuses the ticket words call after call without clearing them: the merging work-group must leave them
    zero, and the merge order is fixed (slices in index order), so two runs give identical bits."""
    B, n, Lp, heads, C = 2, 256, 196, 8, 256
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B * n, 3 * C, generator=g).cuda()
    pt = torch.randn(B * Lp, C, generator=g).cuda()
    lib = _abi.lib()
    scratch = torch.empty(lib.ptx_proxy_attention_scratch_bytes(B, n, Lp, heads, C, 3) // 4, device="cuda")
    outs = []
    for _ in range(3):
        out = torch.empty((B * n, C), device="cuda")
        _abi.check(lib.ptx_proxy_attention(qkv.data_ptr(), pt.data_ptr(), None, out.data_ptr(), scratch.data_ptr(), B, n, Lp,
                                           heads, C, 3, torch.cuda.current_stream().cuda_stream), "attn")
        torch.cuda.synchronize()
        assert int(scratch[:B * heads].view(torch.int32).abs().max()) == 0
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("B", [1, 3])
def test_split_merge_is_stable_under_uneven_load(B):
    """the slices of a (scene, head) publish their partials with write-through (sc1) stores around a relaxed
    agent-scope ticket instead of release / acquire fences, so visibility across XCDs rests on the write-through path.  The
    guide's hand-off rule: test under UNEVEN load with the consumer's caches warm and check every word.  300 back-to-back calls
    (the partial buffer is re-used call after call: a stale line would be the previous call's values, so the inputs alternate
    between two sets), a second stream keeps the memory system busy with copies of changing size, every output word is compared
    with the quiet first pass of its input set."""
    n, Lp, heads, C = 256, 196, 8, 256
    lib = _abi.lib()
    g = torch.Generator().manual_seed(77 + B)
    sets = [(torch.randn(B * n, 3 * C, generator=g).cuda(), (torch.randn(B * Lp, C, generator=g) * 1.5).cuda()) for _ in range(2)]
    scratch = torch.zeros(lib.ptx_proxy_attention_scratch_bytes(B, n, Lp, heads, C, 3) // 4, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(k):
        out = torch.empty((B * n, C), device="cuda")
        _abi.check(lib.ptx_proxy_attention(sets[k][0].data_ptr(), sets[k][1].data_ptr(), None, out.data_ptr(), scratch.data_ptr(),
                                           B, n, Lp, heads, C, 3, st), "attn")
        return out
    ref = [run(0), run(1)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(big)
    bad = 0
    for it in range(300):
        if it % 3 != 2:                                         # uneven: bursts of copies, then nothing
            with torch.cuda.stream(side):
                m = (1 + it % 7) << 22
                dst[:m].copy_(big[:m], non_blocking=True)
        k = (it * 5 + it // 3) & 1
        out = run(k)
        bad += int((out != ref[k]).sum())
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} output words differ from the quiet pass"
