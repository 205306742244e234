// Image proxies: get_img_proxy = Conv2d(512,256,1) + AttentionPool2d + LayerNorm (PRE:335-342,
// 144-177), restated so that the only work proportional to the image is two streaming passes.
//
// AttentionPool2d returns token 0 only (PRE:177) and attention rows are independent, so only
// query 0 is needed.  Everything between the image and that row is linear in the image:
//   x0      = Wc mean_p(f_p) + bc + pos_0                         (mean token, PRE:156-157)
//   q       = Wq x0 + bq ; k0 = Wk x0 ; v0 = Wv x0
//   s_h(0)  = scale q_h . k0_h
//   s_h(p)  = w_h . f_p + e_h(p),  w_h = scale (Wk Wc)_h^T q_h,  e_h(p) = scale q_h . (Wk (bc + pos_p))_h
//             (the k bias adds the same constant to every score of a head and cancels in softmax)
//   a_h     = softmax over the 1 + hw tokens
//   o_h     = a_h(0) v0_h + (Wv Wc)_h g_h + sum_p a_h(p) (Wv (bc + pos_p))_h + bv_h,  g_h = sum_p a_h(p) f_p
//   out     = LayerNorm(Wo o + bo)
// The table products (Wk Wc, Wv Wc, projections of bc + pos) depend on parameters only and are
// built once by ptx_prepare; the per-image small matrix products run as grouped GEMMs (gemm.hip).
// This file holds the three kernels that touch the image itself (HBM-bound, in_dim*hw*4 B / image):
//   k_img_mean    f -> mean_p f                      (pass 1)
//   k_img_scores  s_h(p), softmax -> a_h             (pass 2)
//   k_img_gather  g_h = sum_p a_h(p) f_p             (pass 3, LDS-transposed)
#include "common.h"

namespace ptx {

// one wave per group of 4 channel rows = 4*hw contiguous floats = hw float4
__global__ __launch_bounds__(256) void k_img_mean(const float *__restrict__ img, int ngroups,
                                                  int hw, float *__restrict__ fm)
{
    const int lane = lane_id();
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= ngroups) return;
    const float4 *src = reinterpret_cast<const float4 *>(img + (size_t)g * 4 * hw);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int j = lane; j < hw; j += 64) {
        const float4 v = src[j];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = 4 * j + c;
            const int row = (e >= hw) + (e >= 2 * hw) + (e >= 3 * hw);
            s0 += row == 0 ? vv[c] : 0.f; s1 += row == 1 ? vv[c] : 0.f;
            s2 += row == 2 ? vv[c] : 0.f; s3 += row == 3 ? vv[c] : 0.f;
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if (lane == 0) {
        const float inv = 1.0f / (float)hw;
        *reinterpret_cast<float4 *>(fm + (size_t)g * 4) = make_float4(s0 * inv, s1 * inv, s2 * inv, s3 * inv);
    }
}

int launch_img_mean(const float *img, int nimg, int in_dim, int hw, float *fm, hipStream_t st)
{
    PTX_REQUIRE(in_dim % 4 == 0, "img mean: in_dim=%d must be a multiple of 4", in_dim);
    PTX_REQUIRE((reinterpret_cast<uintptr_t>(img) & 15) == 0, "img_feat must be 16-byte aligned");
    const int ngroups = nimg * (in_dim / 4);
    hipLaunchKernelGGL(k_img_mean, dim3(cdiv(ngroups, 4)), dim3(256), 0, st, img, ngroups, hw, fm);
    PTX_LAUNCHED("k_img_mean");
    return PTX_OK;
}

// One work-group per image.  Wave w owns pixels [64w, 64w+64); every lane walks all channels
// of its pixel (coalesced 256-B rows), 8 head scores accumulate in registers; w_h comes from LDS.
constexpr int kMaxHeads = 8;

__global__ __launch_bounds__(256) void k_img_scores(
    const float *__restrict__ img, const float *__restrict__ we, const float *__restrict__ qkv0,
    int in_dim, int hw, int heads, int C, int KT1, int KT2p, float scale, float *__restrict__ gbuf)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *w_s = sm;                               // [heads][in_dim]
    float *S = sm + (size_t)heads * in_dim;        // [heads][hw + 1]
    const int im = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const float *wim = we + (size_t)im * heads * KT1;
    for (int i = tid; i < heads * in_dim; i += 256) {
        const int h = i / in_dim, c = i - h * in_dim;
        w_s[i] = wim[(size_t)h * KT1 + c];
    }
    __syncthreads();
    const float *f = img + (size_t)im * in_dim * hw;
    for (int p0 = wid * 64; p0 < hw; p0 += 256) {
        const int p = p0 + lane;
        const bool ok = p < hw;
        float acc[kMaxHeads];
#pragma unroll
        for (int h = 0; h < kMaxHeads; ++h) acc[h] = 0.0f;
        for (int c = 0; c < in_dim; c += 4) {
            float fv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) fv[u] = ok ? f[(size_t)(c + u) * hw + p] : 0.0f;
#pragma unroll
            for (int h = 0; h < kMaxHeads; ++h) {
                if (h < heads) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(&w_s[h * in_dim + c]);
                    acc[h] = fmaf(w4.x, fv[0], acc[h]); acc[h] = fmaf(w4.y, fv[1], acc[h]);
                    acc[h] = fmaf(w4.z, fv[2], acc[h]); acc[h] = fmaf(w4.w, fv[3], acc[h]);
                }
            }
        }
        if (ok) {
#pragma unroll
            for (int h = 0; h < kMaxHeads; ++h)
                if (h < heads) S[h * (hw + 1) + 1 + p] = acc[h] + wim[(size_t)h * KT1 + in_dim + 1 + p];
        }
    }
    if (tid < heads) {                              // token 0: s_h(0) = scale * q_h . k0_h
        const int hd = C / heads;
        const float *q = qkv0 + (size_t)im * 3 * C + tid * hd;
        const float *k0 = q + C;
        float s = 0.0f;
        for (int d = 0; d < hd; ++d) s = fmaf(q[d], k0[d], s);
        S[tid * (hw + 1)] = s * scale;
    }
    __syncthreads();
    for (int h = wid; h < heads; h += 4) {          // softmax over hw + 1 tokens, one wave per head
        const float *sh = S + h * (hw + 1);
        float mx = -INFINITY;
        for (int i = lane; i <= hw; i += 64) mx = fmaxf(mx, sh[i]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int i = lane; i <= hw; i += 64) sum += expf(sh[i] - mx);
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float *dst = gbuf + ((size_t)im * heads + h) * KT2p + in_dim;
        for (int i = lane; i < KT2p - in_dim; i += 64) dst[i] = i <= hw ? expf(sh[i] - mx) * inv : 0.0f;
    }
}

int launch_img_scores(const float *img, const float *we, const float *qkv0, int nimg, int in_dim,
                      int hw, int heads, int C, int KT1, int KT2p, float scale, float *gbuf,
                      hipStream_t st)
{
    PTX_REQUIRE(heads <= kMaxHeads && in_dim % 4 == 0, "img scores: heads=%d in_dim=%d unsupported", heads, in_dim);
    const size_t lds = sizeof(float) * ((size_t)heads * in_dim + (size_t)heads * (hw + 1));
    PTX_REQUIRE(lds <= 64 * 1024, "img scores: %zu B of LDS", lds);
    hipLaunchKernelGGL(k_img_scores, dim3(nimg), dim3(256), lds, st, img, we, qkv0, in_dim, hw, heads,
                       C, KT1, KT2p, scale, gbuf);
    PTX_LAUNCHED("k_img_scores");
    return PTX_OK;
}

// One work-group per (image, 64-channel chunk).  The chunk (64*hw contiguous floats) is staged
// in LDS with coalesced 16-B loads; lane = channel then reads its row with stride hw (odd for
// hw = 225 -> conflict-free), wave = pixel quarter, so the attention weights are wave-uniform.
constexpr int kGatherCh = 64;

__global__ __launch_bounds__(256) void k_img_gather(const float *__restrict__ img, int in_dim, int hw,
                                                    int heads, int KT2p, float *__restrict__ gbuf)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tile = sm;                                       // [64][hw]
    float *a_s = sm + (size_t)kGatherCh * hw;               // [hw][8]
    float *red = a_s + (size_t)hw * kMaxHeads;              // [4][8][64]
    const int chunks = in_dim / kGatherCh;
    const int im = blockIdx.x / chunks, c0 = (blockIdx.x - im * chunks) * kGatherCh;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const float4 *src = reinterpret_cast<const float4 *>(img + ((size_t)im * in_dim + c0) * hw);
    float4 *dst = reinterpret_cast<float4 *>(tile);
    const int n4 = kGatherCh * hw / 4;
    for (int j = tid; j < n4; j += 256) dst[j] = src[j];
    for (int i = tid; i < hw * kMaxHeads; i += 256) {
        const int p = i / kMaxHeads, h = i - p * kMaxHeads;
        a_s[i] = h < heads ? gbuf[((size_t)im * heads + h) * KT2p + in_dim + 1 + p] : 0.0f;
    }
    __syncthreads();
    const int per = (hw + 3) / 4;
    const int pbeg = wid * per, pend = min(hw, pbeg + per);
    float acc[kMaxHeads];
#pragma unroll
    for (int h = 0; h < kMaxHeads; ++h) acc[h] = 0.0f;
    const float *row = tile + (size_t)lane * hw;
    for (int p = pbeg; p < pend; ++p) {
        const float fv = row[p];
        const float4 a0 = *reinterpret_cast<const float4 *>(&a_s[p * kMaxHeads]);
        const float4 a1 = *reinterpret_cast<const float4 *>(&a_s[p * kMaxHeads + 4]);
        acc[0] = fmaf(a0.x, fv, acc[0]); acc[1] = fmaf(a0.y, fv, acc[1]);
        acc[2] = fmaf(a0.z, fv, acc[2]); acc[3] = fmaf(a0.w, fv, acc[3]);
        acc[4] = fmaf(a1.x, fv, acc[4]); acc[5] = fmaf(a1.y, fv, acc[5]);
        acc[6] = fmaf(a1.z, fv, acc[6]); acc[7] = fmaf(a1.w, fv, acc[7]);
    }
#pragma unroll
    for (int h = 0; h < kMaxHeads; ++h) red[(wid * kMaxHeads + h) * 64 + lane] = acc[h];
    __syncthreads();
    for (int h = wid; h < heads; h += 4) {
        const float v = red[(0 * kMaxHeads + h) * 64 + lane] + red[(1 * kMaxHeads + h) * 64 + lane] +
                        red[(2 * kMaxHeads + h) * 64 + lane] + red[(3 * kMaxHeads + h) * 64 + lane];
        gbuf[((size_t)im * heads + h) * KT2p + c0 + lane] = v;
    }
}

int launch_img_gather(const float *img, int nimg, int in_dim, int hw, int heads, int KT2p,
                      float *gbuf, hipStream_t st)
{
    PTX_REQUIRE(in_dim % kGatherCh == 0 && heads <= kMaxHeads, "img gather: in_dim=%d heads=%d", in_dim, heads);
    PTX_REQUIRE((kGatherCh * hw) % 4 == 0, "img gather: hw=%d", hw);
    const size_t lds = sizeof(float) * ((size_t)kGatherCh * hw + (size_t)hw * kMaxHeads + 4 * kMaxHeads * 64);
    PTX_REQUIRE(lds <= 160 * 1024, "img gather: %zu B of LDS", lds);
    if (lds > 64 * 1024)
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_img_gather),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_img_gather, dim3(nimg * (in_dim / kGatherCh)), dim3(256), lds, st, img, in_dim,
                       hw, heads, KT2p, gbuf);
    PTX_LAUNCHED("k_img_gather");
    return PTX_OK;
}

}  // namespace ptx
