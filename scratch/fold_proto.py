"""float64 prototype of the folded attention-pool forward / backward (train mode) against torch autograd of the plain form."""
import torch
torch.manual_seed(0)
dt = torch.float64
nimg, Cin, hw, C, heads = 5, 24, 9, 16, 4
hd = C // heads; T = hw + 1; scale = hd ** -0.5
X = torch.randn(nimg, Cin, hw, dtype=dt, requires_grad=True)
wc = torch.randn(C, Cin, dtype=dt, requires_grad=True); bc = torch.randn(C, dtype=dt, requires_grad=True)
pos = torch.randn(T, C, dtype=dt, requires_grad=True)
wq, wk, wv = (torch.randn(C, C, dtype=dt, requires_grad=True) for _ in range(3))
bq, bk, bv = (torch.randn(C, dtype=dt, requires_grad=True) for _ in range(3))
leaves = dict(X=X, wc=wc, bc=bc, pos=pos, wq=wq, wk=wk, wv=wv, bq=bq, bk=bk, bv=bv)
# plain
conv = torch.einsum("oc,icp->ipo", wc, X) + bc                     # (nimg, hw, C)
tok = torch.cat([conv.mean(1, keepdim=True), conv], 1) + pos       # (nimg, T, C)
q = tok[:, 0] @ wq.T + bq
k = tok @ wk.T + bk; v = tok @ wv.T + bv
S = scale * torch.einsum("ihd,ithd->iht", q.view(nimg, heads, hd), k.view(nimg, T, heads, hd))
P = torch.softmax(S, -1)
o = torch.einsum("iht,ithd->ihd", P, v.view(nimg, T, heads, hd)).reshape(nimg, C)
do = torch.randn(nimg, C, dtype=dt)
o.backward(do)
ref = {k_: v_.grad.clone() for k_, v_ in leaves.items()}
o_ref = o.detach()
# folded
with torch.no_grad():
    Xd = X.detach()
    xbar = Xd.mean(2)                                              # (nimg, Cin)
    t0 = xbar @ wc.T + bc + pos[0]
    qf = t0 @ wq.T + bq
    w = scale * torch.einsum("ihd,hdc->ihc", qf.view(nimg, heads, hd), wk.view(heads, hd, C))          # w' (nimg, heads, C)
    posb = pos + bc                                                 # rows >= 1 used
    e = w @ wc                                                      # e' (nimg, heads, Cin)
    sp = torch.einsum("ihc,tc->iht", w, posb)                       # (nimg, heads, T)
    qbk = scale * torch.einsum("ihd,hd->ih", qf.view(nimg, heads, hd), bk.view(heads, hd))
    Sf = torch.empty(nimg, heads, T, dtype=dt)
    Sf[:, :, 0] = torch.einsum("ihc,ic->ih", w, t0) + qbk
    Sf[:, :, 1:] = torch.einsum("ihc,icp->ihp", e, Xd) + sp[:, :, 1:] + qbk[:, :, None]
    Pf = torch.softmax(Sf, -1)
    Ypool = torch.einsum("ihp,icp->ihc", Pf[:, :, 1:], Xd)          # (nimg, heads, Cin)
    g = Ypool @ wc.T + torch.einsum("ihp,pc->ihc", Pf[:, :, 1:], posb[1:]) + Pf[:, :, :1] * t0[:, None, :]
    of = (torch.einsum("ihc,hdc->ihd", g, wv.view(heads, hd, C)) + bv.view(heads, hd)).reshape(nimg, C)
    print("forward", (of - o_ref).abs().max().item())
    # backward
    doh = do.view(nimg, heads, hd)
    dg = torch.einsum("ihd,hdc->ihc", doh, wv.view(heads, hd, C))
    G = {}
    G["bv"] = do.sum(0)
    G["wv"] = torch.einsum("ihd,ihc->hdc", doh, g).reshape(C, C)
    Bv = dg @ wc                                                    # (nimg, heads, Cin)
    dP = torch.empty(nimg, heads, T, dtype=dt)
    dP[:, :, 0] = torch.einsum("ihc,ic->ih", dg, t0)
    dP[:, :, 1:] = torch.einsum("ihc,icp->ihp", Bv, Xd) + torch.einsum("ihc,pc->ihp", dg, posb[1:])
    dS = Pf * (dP - (Pf * dP).sum(-1, keepdim=True))
    Yd = torch.einsum("ihp,icp->ihc", dS[:, :, 1:], Xd)
    u = Yd @ wc.T + torch.einsum("ihp,pc->ihc", dS[:, :, 1:], posb[1:]) + dS[:, :, :1] * t0[:, None, :]
    sig = dS.sum(-1)                                                # ~0
    dq = scale * (torch.einsum("ihc,hdc->ihd", u, wk.view(heads, hd, C)) + sig[:, :, None] * bk.view(heads, hd)).reshape(nimg, C)
    G["wk"] = scale * torch.einsum("ihd,ihc->hdc", qf.view(nimg, heads, hd), u).reshape(C, C)
    G["bk"] = scale * torch.einsum("ihd,ih->hd", qf.view(nimg, heads, hd), sig).reshape(C)
    dt0 = (dS[:, :, :1] * w + Pf[:, :, :1] * dg).sum(1) + dq @ wq
    G["wq"] = dq.T @ t0; G["bq"] = dq.sum(0)
    Acat = torch.cat([w.reshape(-1, C), dg.reshape(-1, C), dt0], 0)
    Wcat = torch.cat([dS.reshape(-1, T), Pf.reshape(-1, T)], 0)
    Ycat = torch.cat([Yd.reshape(-1, Cin), Ypool.reshape(-1, Cin), xbar], 0)
    G["wc"] = Acat.T @ Ycat
    dposv = torch.empty(T, C, dtype=dt)
    dposv[0] = dt0.sum(0)
    dposv[1:] = Wcat[:, 1:].T @ Acat[: 2 * nimg * heads]
    G["pos"] = dposv
    om = Wcat[:, 1:].sum(1)                                          # omega (2 * nimg * heads)
    G["bc"] = (om[:, None] * Acat[: 2 * nimg * heads]).sum(0) + dt0.sum(0)
    Bcat = torch.cat([e.reshape(-1, Cin), Bv.reshape(-1, Cin)], 0)
    b0 = (dt0 @ wc) / hw
    Wd = dS[:, :, 1:]; Wp = Pf[:, :, 1:]
    G["X"] = torch.einsum("ihp,ihc->icp", Wd, e) + torch.einsum("ihp,ihc->icp", Wp, Bv) + b0[:, :, None]
for k_ in ref:
    print(f"{k_:4s} {(G[k_] - ref[k_]).abs().max().item():.3e}  (scale {ref[k_].abs().max().item():.2e})")
