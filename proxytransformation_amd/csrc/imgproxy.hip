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
//   k_img_gather  g_h = sum_p a_h(p) f_p             (pass 3, v_mfma_f32_16x16x4_f32 from global)
#include "common.h"

namespace ptx {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-B load at 4-B alignment
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per group of 4 channel rows = 4*hw contiguous floats = hw float4
__global__ __launch_bounds__(256) void k_img_mean(const float *__restrict__ img, int ngroups,
                                                  int hw, float *__restrict__ fm, uint32_t *gate, uint32_t gate_seq)
{
    mean_prologue(gate, gate_seq);
    const int blk = blockIdx.x;
    const int lane = lane_id();
    const int g = __builtin_amdgcn_readfirstlane(blk * 4 + (threadIdx.x >> 6));
    if (g >= ngroups) return;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(img + (size_t)g * 4 * hw);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int j = lane; j < hw; j += 64) {
        const f32x4 v = __builtin_nontemporal_load(src + j);      // streamed once per pass: keep L2 for weights / tokens
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

int launch_img_mean(const float *img, int nimg, int in_dim, int hw, float *fm, hipStream_t st, uint32_t *gate, uint32_t gate_seq)
{
    PTX_REQUIRE(in_dim % 4 == 0, "img mean: in_dim=%d must be a multiple of 4", in_dim);
    PTX_REQUIRE((reinterpret_cast<uintptr_t>(img) & 15) == 0, "img_feat must be 16-byte aligned");
    const int ngroups = nimg * (in_dim / 4);
    hipLaunchKernelGGL(k_img_mean, dim3(cdiv(ngroups, 4)), dim3(256), 0, st, img, ngroups, hw, fm, gate, gate_seq);
    PTX_LAUNCHED("k_img_mean");
    return PTX_OK;
}

constexpr int kMaxHeads = 16;     // heads per image: 4, 8 (the reference's) or 16 (r04: num_heads generality, head_dim 32 / 64)

// ---- pass 2: scores + softmax ---------------------------------------------------------------
// One work-group (8 waves) per image; wave w streams channels [w*in_dim/8, (w+1)*in_dim/8).
// Lane j owns pixels 4j..4j+3 of EVERY row (one 16-B load per lane and row; rows are hw*4 B
// apart, so the loads are only 4-B aligned -- global_load_dwordx4 takes that); the hw%4 tail
// pixels are covered by one more lane that re-reads the last 4 pixels, so a row is exactly one
// load instruction.  The 8 head weights of a row are wave-uniform: they sit in 8 VGPRs per wave
// (lane = channel) and are broadcast with v_readlane, so the inner loop is 32 FMAs + 8 readlanes
// per 16-B load and touches memory only for the image.
// What the time is (r01, MI355X, cfg2 B=4, fp32 features): 66 us of streaming (5.5 TB/s) + ~10 us of
// tree / softmax epilogue that cannot overlap because all 784 work-groups are resident and in step.
// Measured alternatives that were NOT faster: LDS-broadcast weights, interleaving the
// waves' rows so an image is read front to back, 2 / 4 / 16 waves per image, unroll 2 / 8, explicit register double-buffering, 16-B
// aligned rows, and (image, 64- or 128-channel chunk) work-groups with a last-arriver reduction
// (write-through partials + agent-scope counter): the per-chunk hand-off costs more than the
// 784-images-over-256-CUs imbalance it removes.
constexpr int kScoreWaves = 8;

template <int HEADS>
__global__ __launch_bounds__(kScoreWaves * 64) void k_img_scores(
    const float *__restrict__ img, const float *__restrict__ we, const float *__restrict__ qkv0,
    int in_dim, int hw, int C, int KT1, int KT2p, float scale, float *__restrict__ gbuf)
{
    constexpr int heads = HEADS, NW = kScoreWaves;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *red = sm;                               // [NW/2][heads][64 lanes x 4]
    float *S = sm + (NW / 2) * heads * 256;        // [heads][hw + 1]
    // pass 2 walks the images in the opposite order of pass 1 (what pass 1 streamed last is the
    // most likely to still sit in the 256 MiB Infinity Cache)
    const int im = gridDim.x - 1 - blockIdx.x, tid = threadIdx.x, lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *wim = we + (size_t)im * heads * KT1;
    const float *f = img + (size_t)im * in_dim * hw;
    const int nv4 = hw >> 2, tail = hw & 3;
    const bool edge = tail != 0 && lane == nv4;
    const bool vec = lane < nv4 || edge;
    const int poff = edge ? hw - 4 : 4 * lane;
    const int cfirst = edge ? 4 - tail : 0;        // first valid component of this lane
    float acc[HEADS][4];
#pragma unroll
    for (int h = 0; h < HEADS; ++h) { acc[h][0] = acc[h][1] = acc[h][2] = acc[h][3] = 0.0f; }
    const int cper = in_dim / NW, cbeg = wid * cper;        // a multiple of 8, <= 256 (validated by the host)
    // the wave's head weights w_h[cbeg .. cbeg+cper) live in 8 VGPRs (lane = channel) and are
    // broadcast per row with v_readlane: no memory instruction besides the image load in the loop
    // (8 scalar loads per row cost ~20 % of this kernel)
    float wreg[HEADS];
    // token 0: s_h(0) = scale * q_h . k0_h -- wave h computes head h up front (one load round trip that
    // overlaps the first image loads; as a 32-step scalar loop after the stream it was a chain of
    // dependent round trips that cost ~15 us per launch)
    for (int h0 = wid; h0 < heads; h0 += NW) {
        const int hd = C / heads;
        const float *q = qkv0 + (size_t)im * 3 * C + h0 * hd;
        const float qk = lane < hd ? q[lane] * q[C + lane] : 0.0f;
        const float s0 = wave_sum(qk);
        if (lane == 0) S[h0 * (hw + 1)] = s0 * scale;
    }
    // Branch-free inner loop: lanes beyond the row re-read lane 0's pixels (their accumulators are
    // never stored), so the body is one basic block and UNR row loads are in flight per wave.  (With
    // an `if (lane < nv4)` around the load hipcc neither unrolled nor hoisted it: one exposed
    // ~1 us round trip per row and wave -- the kernel ran at 88 us regardless of the bytes moved.)
    constexpr int UNR = 8;
    const float *fl = f + (vec ? poff : 0);
    // r05: any in_dim up to 2048 (a stock ResNet-50 C5): the wave's channel slice in chunks of <= 64, the chunk's head weights
    // re-loaded into the lanes (wave-uniform trip counts: no divergence)
    for (int c0 = 0; c0 < cper; c0 += 64) {
    const int cn = min(64, cper - c0);
#pragma unroll
    for (int h = 0; h < HEADS; ++h) wreg[h] = lane < cn ? wim[(size_t)h * KT1 + cbeg + c0 + lane] : 0.0f;
    for (int cc = 0; cc < cn; cc += UNR) {                  // cper is a multiple of 8 (validated by the host)
        f4u t[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            t[u] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(fl + (size_t)(cbeg + c0 + cc + u) * hw));
        __builtin_amdgcn_sched_barrier(0);                  // keep all UNR loads ahead of the first FMA
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                const float wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wreg[h]), cc + u));
                acc[h][0] = fmaf(wv, t[u].x, acc[h][0]); acc[h][1] = fmaf(wv, t[u].y, acc[h][1]);
                acc[h][2] = fmaf(wv, t[u].z, acc[h][2]); acc[h][3] = fmaf(wv, t[u].w, acc[h][3]);
            }
        }
    }
    }
    // positional score terms e_h(p) of this lane's pixels: requested before the tree so that the
    // round trip hides behind it (used by wave 0 only)
    float ev[HEADS][4];
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            ev[h][c] = (wid == 0 && vec && c >= cfirst) ? wim[(size_t)h * KT1 + in_dim + 1 + poff + c] : 0.0f;
    // fixed-order tree over the NW channel slices: upper half parks, lower half adds
    // (red is indexed by lane, not by pixel: the edge lane overlaps its neighbour's pixels)
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
        if (wid >= half && wid < 2 * half) {
#pragma unroll
            for (int h = 0; h < HEADS; ++h)
                *reinterpret_cast<float4 *>(red + ((size_t)(wid - half) * heads + h) * 256 + 4 * lane) =
                    make_float4(acc[h][0], acc[h][1], acc[h][2], acc[h][3]);
        }
        __syncthreads();
        if (wid < half) {
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                const float4 t = *reinterpret_cast<const float4 *>(red + ((size_t)wid * heads + h) * 256 + 4 * lane);
                acc[h][0] += t.x; acc[h][1] += t.y; acc[h][2] += t.z; acc[h][3] += t.w;
            }
        }
        __syncthreads();
    }
    if (wid == 0 && vec) {
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
            float *d = S + h * (hw + 1) + 1 + poff;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c >= cfirst) d[c] = acc[h][c] + ev[h][c];
        }
    }
    __syncthreads();
    for (int h = wid; h < heads; h += NW) {         // softmax over hw + 1 tokens, one wave per head
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
    PTX_REQUIRE((heads == 4 || heads == 8 || heads == 16) && in_dim % (8 * kScoreWaves) == 0 && in_dim / kScoreWaves <= 256,
                "img scores: heads=%d in_dim=%d unsupported", heads, in_dim);
    PTX_REQUIRE(hw >= 4 && (hw >> 2) + ((hw & 3) ? 1 : 0) <= 64, "img scores: hw=%d (supported: 4..256 pixels)", hw);
    const size_t lds = sizeof(float) * ((size_t)(kScoreWaves / 2) * heads * 256 + (size_t)heads * (hw + 1));
    PTX_REQUIRE(lds <= 160 * 1024, "img scores: %zu B of LDS", lds);
#define PTX_SCORES(H_)                                                                                                              \
    do {                                                                                                                          \
        if (lds > 64 * 1024)                                                                                                      \
            PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_img_scores<H_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_img_scores<H_>, dim3(nimg), dim3(kScoreWaves * 64), lds, st, img, we, qkv0, in_dim, hw, C, KT1, KT2p, scale, gbuf); \
    } while (0)
    if (heads == 4) PTX_SCORES(4); else if (heads == 8) PTX_SCORES(8); else PTX_SCORES(16);
#undef PTX_SCORES
    PTX_LAUNCHED("k_img_scores");
    return PTX_OK;
}

// ---- pass 3: g_h = sum_p a_h(p) f_p on the matrix cores ------------------------------------------
// G^T (channels x heads) = F (channels x pixels) . A^T (pixels x heads) with v_mfma_f32_16x16x4_f32:
// a wave owns 16 channel rows; lane (ci = l & 15, kq = l >> 4) loads 16 B = 4 pixels of row ci at
// pixel 16*kb + 4*kq straight from global (no LDS staging of the image), MFMA step t contracts
// pixel 16*kb + 4*kq + t; the B operand a_h(p) (h = l & 15, zero for h >= heads) comes from a
// 7 KB LDS copy of the softmax output.  fp32 in / fp32 accumulate: exact products.
// (The quad-contiguous map of imgpool.hip -- 4 channel rows x 64 pixels per MFMA, 2.25x the f32 matrix work for
// loads that stream at full rate -- was measured at 93 us against 92 for this kernel: with the f32 MFMA at
// 1/16 of the bf16 rate the pass turns matrix-bound.)
constexpr int kGatherCh = 64;      // channels per work-group (4 waves x 16)

__global__ __launch_bounds__(256) void k_img_gather(const float *__restrict__ img, int in_dim, int hw,
                                                    int heads, int KT2p, float *__restrict__ gbuf)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hwp = (hw + 3) & ~3;
    float *a_s = sm;                                        // [heads][hwp], zero padded
    const int chunks = in_dim / kGatherCh;
    const int im = blockIdx.x / chunks, c0 = (blockIdx.x - im * chunks) * kGatherCh;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int ci = lane & 15, kq = lane >> 4;
    const int cb = c0 + wid * 16;
    const float *row = img + ((size_t)im * in_dim + cb + ci) * hw;
    const bool live = ci < heads;
    const float *arow = a_s + (size_t)(live ? ci : 0) * hwp;
    // 32 pixels per step: lane (ci, kq) reads pixels 32*kb + 8*kq .. +7 (two 16-B loads), so the
    // four kq groups of a row consume one full 128-B line back to back; MFMA step (j, t) contracts
    // pixel 32*kb + 8*kq + 4*j + t on both operands.
    // (default cache policy on purpose: f0 / f1 and the neighbouring kq lanes share 128-B lines;
    //  non-temporal loads here were measured 1.7x slower)
    const int nkb = hw >> 5;
    constexpr int PRE = 4;                                   // steps whose image loads are issued before the
    f4u pf0[PRE], pf1[PRE];                                  // softmax weights are staged (they do not depend on them)
#pragma unroll
    for (int kb = 0; kb < PRE; ++kb) {
        const int p0 = 32 * min(kb, nkb - 1) + 8 * kq;
        pf0[kb] = *reinterpret_cast<const f4u *>(row + p0);
        pf1[kb] = *reinterpret_cast<const f4u *>(row + p0 + 4);
    }
    {   // all loads of the probabilities first, then the LDS stores (as one loop the compiler waited for each
        // load -- and for the feature loads above -- in turn)
        constexpr int NR = 16;                                  // heads * hwp <= 16 * 256 (validated by the host)
        float av[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = tid + 256 * r, h = i / hwp, p = i - h * hwp;
            av[r] = (i < heads * hwp && p < hw) ? gbuf[((size_t)im * heads + h) * KT2p + in_dim + 1 + p] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = tid + 256 * r;
            if (i < heads * hwp) a_s[i] = av[r];
        }
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](int kb, const f4u &f0, const f4u &f1) {
        const int p0 = 32 * kb + 8 * kq;
        float4 b0 = *reinterpret_cast<const float4 *>(arow + p0);
        float4 b1 = *reinterpret_cast<const float4 *>(arow + p0 + 4);
        if (!live) { b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = b0; }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.w, b0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.x, b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.y, b1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.z, b1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.w, b1.w, acc, 0, 0, 0);
    };
#pragma unroll
    for (int kb = 0; kb < PRE; ++kb)
        if (kb < nkb) step(kb, pf0[kb], pf1[kb]);
#pragma unroll 3
    for (int kb = PRE; kb < nkb; ++kb) {
        const int p0 = 32 * kb + 8 * kq;
        const f4u f0 = *reinterpret_cast<const f4u *>(row + p0);
        const f4u f1 = *reinterpret_cast<const f4u *>(row + p0 + 4);
        step(kb, f0, f1);
    }
    for (int pp = 32 * nkb; pp < hw; pp += 4) {             // pixel tail (1 pixel for 15 x 15)
        const int p = pp + kq;
        const float fa = p < hw ? row[p] : 0.0f;
        const float ba = (p < hw && live) ? arow[p] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, ba, acc, 0, 0, 0);
    }
    // D[row = 4*kq + r][col = l & 15] = G^T[channel cb + 4*kq + r][head l & 15]
    if (live) {
        float *dst = gbuf + ((size_t)im * heads + ci) * KT2p + cb + 4 * kq;
        *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

int launch_img_gather(const float *img, int nimg, int in_dim, int hw, int heads, int KT2p,
                      float *gbuf, hipStream_t st)
{
    PTX_REQUIRE(in_dim % kGatherCh == 0 && heads <= kMaxHeads, "img gather: in_dim=%d heads=%d", in_dim, heads);
    PTX_REQUIRE(hw <= 256, "img gather: hw=%d (max 256 pixels)", hw);
    const int hwp = (hw + 3) & ~3;
    const size_t lds = sizeof(float) * (size_t)heads * hwp;
    PTX_REQUIRE(lds <= 64 * 1024, "img gather: %zu B of LDS", lds);
    hipLaunchKernelGGL(k_img_gather, dim3(nimg * (in_dim / kGatherCh)), dim3(256), lds, st, img, in_dim,
                       hw, heads, KT2p, gbuf);
    PTX_LAUNCHED("k_img_gather");
    return PTX_OK;
}

}  // namespace ptx
