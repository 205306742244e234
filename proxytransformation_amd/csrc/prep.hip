// Parameter-only tables (run once per set of weights by ptx_prepare, never on the per-scene
// path): folded eval-mode BatchNorm scale/shift, the per-slot bias tables of ProxyAttention
// (PRE:212-215), and the products that fold Conv2d(512,256,1) into AttentionPool2d's key /
// value projections (see imgproxy.hip).  One thread per output element: clarity over speed.
#include "common.h"

namespace ptx {

// eval BatchNorm as ATen applies it: alpha = w / sqrt(var + eps), beta = b - mean * alpha
__global__ void k_prep_bn(const float *w, const float *b, const float *mean, const float *var,
                          int n, float eps, float *ab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float alpha = w[i] / sqrtf(var[i] + eps);
    ab[i] = alpha;
    ab[n + i] = b[i] - mean[i] * alpha;
}

// bias[j][y*s+x] = bilinear(pb[j] 4x4 -> s x s, align_corners=False)[y][x] + pc[j][y] + pr[j][x]
// for the first C entries of the s x s grid: s = sqrt(C) for the reference's C = 256 (PRE:196, the whole grid);
// for a C that is no perfect square (512: the reference cannot run it) s = ceil(sqrt(C)), cropped.
__global__ void k_prep_posbias(const float *pb, const float *pc, const float *pr, int Mk, int s, int C,
                               float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mk * C) return;
    const int j = i / C, yx = i - j * C, y = yx / s, x = yx - y * s;
    const float sc = 4.0f / (float)s;
    float sy = sc * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = sc * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < 3 ? 1 : 0), x1 = x0 + (x0 < 3 ? 1 : 0);
    const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
    const float *p = pb + (size_t)j * 16;
    const float v = ly0 * (lx0 * p[y0 * 4 + x0] + lx1 * p[y0 * 4 + x1]) +
                    ly1 * (lx0 * p[y1 * 4 + x0] + lx1 * p[y1 * 4 + x1]);
    out[i] = v + (pc[(size_t)j * s + y] + pr[(size_t)j * s + x]);
}

// Token 0 of AttentionPool2d is linear in the image mean (PRE:156-157):
//   [q | k0 | v0] = [Wq; Wk; Wv] (Wc mean(f) + bc + pos_0) + [bq; 0; 0] = W3 mean(f) + b3
// W3 (3C, in_dim) = [Wq; Wk; Wv] Wc,  b3 (3C) = [Wq; Wk; Wv] (bc + pos_0) + [bq; 0; 0]
__global__ void k_prep_w3(const float *qw, const float *kw, const float *vw, const float *qb,
                          const float *cw, const float *cb, const float *pos, int C, int in_dim,
                          float *w3, float *b3)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * C * (in_dim + 1)) return;
    const int n = i / (in_dim + 1), c = i - n * (in_dim + 1);
    const int part = n / C, r = n - part * C;
    const float *wrow = (part == 0 ? qw : (part == 1 ? kw : vw)) + (size_t)r * C;
    float s = 0.0f;
    if (c < in_dim) {
        for (int j = 0; j < C; ++j) s = fmaf(wrow[j], cw[(size_t)j * in_dim + c], s);
        w3[(size_t)n * in_dim + c] = s;
    } else {
        for (int j = 0; j < C; ++j) s = fmaf(wrow[j], cb[j] + pos[j], s);
        b3[n] = s + (part == 0 ? qb[r] : 0.0f);
    }
}

// T1[h][t][d], t < in_dim : scale * sum_j Wk[h*hd+d][j] * Wc[j][t]
//              t = in_dim+i: scale * sum_j Wk[h*hd+d][j] * (bc[j] + pos[i][j])   (i = 0 unused: 0)
__global__ void k_prep_t1(const float *kw, const float *cw, const float *cb, const float *pos, int C,
                          int in_dim, int hw, int heads, float scale, float *t1)
{
    const int hd = C / heads, KT1 = in_dim + hw + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= heads * KT1 * hd) return;
    const int h = i / (KT1 * hd), rem = i - h * KT1 * hd, t = rem / hd, d = rem - t * hd;
    const float *wk = kw + (size_t)(h * hd + d) * C;
    float s = 0.0f;
    if (t < in_dim) {
        for (int j = 0; j < C; ++j) s = fmaf(wk[j], cw[(size_t)j * in_dim + t], s);
    } else if (t > in_dim) {
        const float *pi = pos + (size_t)(t - in_dim) * C;
        for (int j = 0; j < C; ++j) s = fmaf(wk[j], cb[j] + pi[j], s);
    }
    t1[i] = s * scale;
}

// T2[h][d][t], t < in_dim : sum_j Wv[h*hd+d][j] * Wc[j][t]
//              t = in_dim+i: sum_j Wv[h*hd+d][j] * (bc[j] + pos[i][j]) for 1 <= i <= hw, else 0
__global__ void k_prep_t2(const float *vw, const float *cw, const float *cb, const float *pos, int C,
                          int in_dim, int hw, int heads, int KT2p, float *t2)
{
    const int hd = C / heads;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= heads * hd * KT2p) return;
    const int h = i / (hd * KT2p), rem = i - h * hd * KT2p, d = rem / KT2p, t = rem - d * KT2p;
    const float *wv = vw + (size_t)(h * hd + d) * C;
    float s = 0.0f;
    if (t < in_dim) {
        for (int j = 0; j < C; ++j) s = fmaf(wv[j], cw[(size_t)j * in_dim + t], s);
    } else if (t > in_dim && t <= in_dim + hw) {
        const float *pi = pos + (size_t)(t - in_dim) * C;
        for (int j = 0; j < C; ++j) s = fmaf(wv[j], cb[j] + pi[j], s);
    }
    t2[i] = s;
}

// LayerNorm fold (GemmProb::lnp_in): Wg[n][k] = W[n][k] gamma[k];  s[n] = sum_k Wg[n][k] (of the ROUNDED products, the
// values the GEMM multiplies);  c[n] = sum_k W[n][k] beta[k] + bias[n].  One wave per output row.
__global__ void k_prep_lnfold(const float *W, const float *gamma, const float *beta, const float *bias, int N, int K,
                              float *Wg, float *sv, float *cv)
{
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    float s1 = 0.0f, c1 = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float wv = W[(size_t)n * K + k], g = wv * gamma[k];
        Wg[(size_t)n * K + k] = g;
        s1 += g;
        c1 = fmaf(wv, beta[k], c1);
    }
    s1 = wave_sum(s1); c1 = wave_sum(c1);
    if (lane == 0) { sv[n] = s1; cv[n] = c1 + (bias ? bias[n] : 0.0f); }
}

int run_prepare(const PtxShape &s, const PtxWeights &w, float *prep, hipStream_t st)
{
    const PrepLayout P = prep_layout(s);
    const int C = s.C, T = 256;
    hipLaunchKernelGGL(k_prep_bn, dim3(1), dim3(T), 0, st, w.offset.bn_w, w.offset.bn_b,
                       w.offset.bn_mean, w.offset.bn_var, kSlotHidden, s.bn_eps, prep + P.off_ab);
    hipLaunchKernelGGL(k_prep_bn, dim3(cdiv(C, T)), dim3(T), 0, st, w.encoder.bn_w, w.encoder.bn_b,
                       w.encoder.bn_mean, w.encoder.bn_var, C, s.bn_eps, prep + P.enc_ab);
    hipLaunchKernelGGL(k_prep_bn, dim3(1), dim3(T), 0, st, w.text_trans_norm.w, w.text_trans_norm.b,
                       w.text_trans_norm.mean, w.text_trans_norm.var, 3, s.bn_eps, prep + P.ttn_ab);
    hipLaunchKernelGGL(k_prep_bn, dim3(1), dim3(T), 0, st, w.img_trans_norm.w, w.img_trans_norm.b,
                       w.img_trans_norm.mean, w.img_trans_norm.var, 9, s.bn_eps, prep + P.itn_ab);
    PTX_LAUNCHED("k_prep_bn");
    int sd = 1;
    while (sd * sd < C) ++sd;
    const int nb = s.Mk * C;
    hipLaunchKernelGGL(k_prep_posbias, dim3(cdiv(nb, T)), dim3(T), 0, st, w.text.pb_bias, w.text.pc_bias,
                       w.text.pr_bias, s.Mk, sd, C, prep + P.posb_t);
    hipLaunchKernelGGL(k_prep_posbias, dim3(cdiv(nb, T)), dim3(T), 0, st, w.img.pb_bias, w.img.pc_bias,
                       w.img.pr_bias, s.Mk, sd, C, prep + P.posb_i);
    PTX_LAUNCHED("k_prep_posbias");
    hipLaunchKernelGGL(k_prep_w3, dim3(cdiv(3 * C * (s.in_dim + 1), T)), dim3(T), 0, st, w.q_w, w.k_w, w.v_w,
                       w.q_b, w.cm_w, w.cm_b, w.pos, C, s.in_dim, prep + P.w3, prep + P.b3);
    PTX_LAUNCHED("k_prep_w3");
    const float scale = (float)(1.0 / sqrt((double)P.hd));                    // head_dim ** -0.5
    hipLaunchKernelGGL(k_prep_t1, dim3(cdiv(s.heads * P.KT1 * P.hd, T)), dim3(T), 0, st, w.k_w, w.cm_w,
                       w.cm_b, w.pos, C, s.in_dim, s.hw, s.heads, scale, prep + P.t1);
    PTX_LAUNCHED("k_prep_t1");
    hipLaunchKernelGGL(k_prep_t2, dim3(cdiv(s.heads * P.hd * P.KT2p, T)), dim3(T), 0, st, w.v_w, w.cm_w,
                       w.cm_b, w.pos, C, s.in_dim, s.hw, s.heads, P.KT2p, prep + P.t2);
    PTX_LAUNCHED("k_prep_t2");
    hipLaunchKernelGGL(k_prep_lnfold, dim3(cdiv(C, 4)), dim3(T), 0, st, w.img.pp_w, w.norm_img_w, w.norm_img_b, w.img.pp_b,
                       C, C, prep + P.ppg_w, prep + P.ppg_s, prep + P.ppg_c);
    const PtxBlock *blk[2] = {&w.text, &w.img};
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL(k_prep_lnfold, dim3(cdiv(s.hidden, 4)), dim3(T), 0, st, blk[i]->fc1_w, blk[i]->norm2_w,
                           blk[i]->norm2_b, blk[i]->fc1_b, s.hidden, C, prep + P.fc1g_w[i], prep + P.fc1g_s[i],
                           prep + P.fc1g_c[i]);
    PTX_LAUNCHED("k_prep_lnfold");
    // the slot-bias rows through the qkv projection (+ its bias): (LN1(x) + posb) W^T + b = LN1(x) W^T + [posb W^T + b] -- the bracket is
    // parameter-only, which lets the early-proxy path project LN1(x) of ALL clusters before the selection is known (api.hip)
    {
        GemmBatch g{}; g.n = 2;
        for (int i = 0; i < 2; ++i)
            g.p[i] = GemmProb{prep + (i == 0 ? P.posb_t : P.posb_i), blk[i]->qkv_w, prep + P.qkvb[i], blk[i]->qkv_b, nullptr, nullptr, nullptr,
                              s.Mk, 3 * C, C, C, C, 3 * C, 0, 0, 0, EPI_NONE};
        PTX_TRY(launch_gemm(g, st));
    }
    // the fused Mlp's weights, split once into bf16 planes in MFMA fragment order (mlp.hip)
    if (mlp_fused_supported(C, s.hidden, 1, 0))
        for (int i = 0; i < 2; ++i) {
            PTX_TRY(launch_prep_planes(prep + P.fc1g_w[i], s.hidden, C, prep + P.mlp_w1p[i], st));
            PTX_TRY(launch_prep_planes(blk[i]->fc2_w, C, s.hidden, prep + P.mlp_w2p[i], st));
        }
    return PTX_OK;
}

}  // namespace ptx
