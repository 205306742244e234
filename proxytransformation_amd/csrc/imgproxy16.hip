// The three image-streaming kernels of imgproxy.hip for image features STORED as bf16 or fp16
// (an AMP backbone hands over half-precision feature maps).  bf16 features of the path's own shape
// (in_dim = 512, 128 < hw <= 255) take the two-pass matrix-pipe route of imgpool.hip after the mean pass of
// this file; these three-pass kernels serve fp16 storage and any other shape.  Only the
// storage changes: every element is widened to fp32 on load and all arithmetic, accumulation and
// outputs are fp32 exactly as in the fp32 kernels, so the result equals the fp32 path run on the
// same (rounded) inputs.  The dominant HBM stream of the path halves (in_dim*hw*2 B per image).
//
// Layout facts used below: a row (one channel of one image) is hw*2 bytes, i.e. only 2-byte aligned
// (hw = 225 is odd); gfx950 global_load_dwordx4 accepts that.  A 16-B load holds 8 pixels, so a
// 15x15 row needs only 29 lanes: every instruction serves TWO rows (lanes 0-31 row r, lanes 32-63
// row r+1).
#include <hip/hip_fp16.h>
#include <cstdlib>

#include "common.h"

namespace ptx {

__device__ __forceinline__ float half_bits_to_float(unsigned short u)
{
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}

typedef unsigned int u4u2 __attribute__((ext_vector_type(4), aligned(2)));    // 16-B load at 2-B alignment
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxHeads = 16;     // heads per image: 4, 8 (the reference's) or 16 (r04: num_heads generality, head_dim 32 / 64)

// widen the 8 half-precision values packed in 4 dwords (element 2i = low half of dword i)
template <int DT>   // 1 = bf16, 2 = fp16
__device__ __forceinline__ void widen8(const u32x4 &d, float (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (DT == 1) {
            v[2 * i] = __uint_as_float(d[i] << 16);
            v[2 * i + 1] = __uint_as_float(d[i] & 0xffff0000u);
        } else {
            v[2 * i] = half_bits_to_float((unsigned short)(d[i] & 0xffffu));
            v[2 * i + 1] = half_bits_to_float((unsigned short)(d[i] >> 16));
        }
    }
}
template <int DT>
__device__ __forceinline__ float widen1(unsigned short u)
{
    return DT == 1 ? __uint_as_float((unsigned int)u << 16) : half_bits_to_float(u);
}

// per-lane view of a row: lane j (within its half-wave) owns pixels 8j..8j+7; the hw % 8 tail pixels
// are covered by one more lane that re-reads the LAST 8 pixels and keeps only the last `tail` ones
struct RowLanes {
    int half, j, poff, cfirst; bool act;
    __device__ RowLanes(int lane, int hw) {
        half = lane >> 5; j = lane & 31;
        const int nv8 = hw >> 3, tail = hw & 7;
        const bool edge = tail != 0 && j == nv8;
        act = j < nv8 || edge;
        poff = edge ? hw - 8 : 8 * j;
        cfirst = edge ? 8 - tail : 0;
    }
};

// ---- pass 1: per-channel means.  One wave per kMeanGroups x 8 rows: all 4 * kMeanGroups load instructions (2 rows
// each) are issued before the first reduction, so a wave keeps 14 KB in flight instead of 3.6 KB (r02: one group per
// wave ran at 3.5 TB/s at 4 scenes / GPU and 2.9 TB/s at 32 -- 400k short-lived waves).
// r03: this pass's load map alone (scratch/rowspan_bench.hip, "rows2, 4 waves") runs at 6.3 / 6.7 TB/s at 4 / 32 scenes per GPU,
// the pass inside the step at 5.4 / 4.5-4.7 -- next to the clustering stream's kernels.  The reduction is not what holds it:
// with the eight values of a lane summed by four v_dot2c_f32_bf16 against a per-lane vector of ones / zeros (instead of 8
// conversions + 8 selects + 8 adds; parity-green) the pass stays at 4.5 TB/s at 32 scenes and the step within +-0.7 % -- not kept.
constexpr int kMeanGroups = 4;

template <int DT>
__global__ __launch_bounds__(256) void k_img_mean16(const unsigned short *__restrict__ img, int ngroups,
                                                    int hw, float *__restrict__ fm, uint32_t *gate, uint32_t gate_seq)
{
    mean_prologue(gate, gate_seq);
    const int blk = blockIdx.x;
    const int lane = lane_id();
    const int g0 = __builtin_amdgcn_readfirstlane((blk * 4 + (threadIdx.x >> 6)) * kMeanGroups);
    if (g0 >= ngroups) return;
    const RowLanes rl(lane, hw);
    u32x4 d[kMeanGroups][4];
#pragma unroll
    for (int gg = 0; gg < kMeanGroups; ++gg) {
        const int g = min(g0 + gg, ngroups - 1);                // tail groups re-read the last one (not stored)
        const unsigned short *base = img + (size_t)g * 8 * hw;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            d[gg][u] = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(base + (size_t)(2 * u + rl.half) * hw + (rl.act ? rl.poff : 0)));
    }
    const float inv = 1.0f / (float)hw;
#pragma unroll
    for (int gg = 0; gg < kMeanGroups; ++gg) {
        float out = 0.0f;                                       // lane r (r < 8) ends up with the mean of row r
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[8], s = 0.0f;
            widen8<DT>(d[gg][u], v);
#pragma unroll
            for (int c = 0; c < 8; ++c) s += (rl.act && c >= rl.cfirst) ? v[c] : 0.0f;
            s += PTX_ROR_F(s, 8); s += PTX_ROR_F(s, 4); s += PTX_ROR_F(s, 2); s += PTX_ROR_F(s, 1);   // 16-lane rows
            const float lo = PTX_LANE_F(s, 0) + PTX_LANE_F(s, 16), hi = PTX_LANE_F(s, 32) + PTX_LANE_F(s, 48);
            if (lane == 2 * u) out = lo * inv;
            if (lane == 2 * u + 1) out = hi * inv;
        }
        if (lane < 8 && g0 + gg < ngroups) fm[(size_t)(g0 + gg) * 8 + lane] = out;
    }
}

// ---- pass 2: scores + softmax.  Exactly the structure of k_img_scores (imgproxy.hip): one work-group
// of 8 waves per image, wave w owns channels [w*in_dim/8, (w+1)*in_dim/8), lane j owns pixels
// 4j..4j+3 of every row (one 8-B load per lane and row here), head weights in 8 VGPRs broadcast by
// v_readlane, 32 accumulators per lane.  (Two rows per 16-B-load instruction -- 64 accumulators per
// lane plus a per-half weight select -- was measured 1.7x slower: 128+ VGPRs, one work-group per CU.)
constexpr int kScoreWaves = 8;
typedef unsigned int u2u2 __attribute__((ext_vector_type(2), aligned(2)));    // 8-B load at 2-B alignment
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int HEADS, int DT>
__global__ __launch_bounds__(kScoreWaves * 64) void k_img_scores16(
    const unsigned short *__restrict__ img, const float *__restrict__ we, const float *__restrict__ qkv0,
    int in_dim, int hw, int C, int KT1, int KT2p, float scale, float *__restrict__ gbuf)
{
    constexpr int heads = HEADS, NW = kScoreWaves;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *red = sm;                               // [NW/2][heads][64 lanes x 4]
    float *S = sm + (NW / 2) * heads * 256;        // [heads][hw + 1]
    const int im = gridDim.x - 1 - blockIdx.x, tid = threadIdx.x, lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *wim = we + (size_t)im * heads * KT1;
    const unsigned short *f = img + (size_t)im * in_dim * hw;
    const int nv4 = hw >> 2, tail = hw & 3;
    const bool edge = tail != 0 && lane == nv4;
    const bool vec = lane < nv4 || edge;
    const int poff = edge ? hw - 4 : 4 * lane;
    const int cfirst = edge ? 4 - tail : 0;
    float acc[HEADS][4];
#pragma unroll
    for (int h = 0; h < HEADS; ++h) { acc[h][0] = acc[h][1] = acc[h][2] = acc[h][3] = 0.0f; }
    const int cper = in_dim / NW, cbeg = wid * cper;        // a multiple of 8, <= 256 (validated by the host)
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
    // branch-free inner loop (see k_img_scores): lanes beyond the row re-read lane 0's pixels
    constexpr int UNR = 8;
    const unsigned short *fl = f + (vec ? poff : 0);
    // r05: any in_dim up to 2048: the wave's channel slice in chunks of <= 64 (see k_img_scores)
    for (int c0 = 0; c0 < cper; c0 += 64) {
    const int cn = min(64, cper - c0);
#pragma unroll
    for (int h = 0; h < HEADS; ++h) wreg[h] = lane < cn ? wim[(size_t)h * KT1 + cbeg + c0 + lane] : 0.0f;
    for (int cc = 0; cc < cn; cc += UNR) {                  // cper is a multiple of 8 (validated by the host)
        u32x2 d[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            d[u] = __builtin_nontemporal_load(reinterpret_cast<const u2u2 *>(fl + (size_t)(cbeg + c0 + cc + u) * hw));
        __builtin_amdgcn_sched_barrier(0);                  // keep all UNR loads ahead of the first FMA
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float v0, v1, v2, v3;
            if (DT == 1) {
                v0 = __uint_as_float(d[u][0] << 16); v1 = __uint_as_float(d[u][0] & 0xffff0000u);
                v2 = __uint_as_float(d[u][1] << 16); v3 = __uint_as_float(d[u][1] & 0xffff0000u);
            } else {
                v0 = half_bits_to_float((unsigned short)(d[u][0] & 0xffffu)); v1 = half_bits_to_float((unsigned short)(d[u][0] >> 16));
                v2 = half_bits_to_float((unsigned short)(d[u][1] & 0xffffu)); v3 = half_bits_to_float((unsigned short)(d[u][1] >> 16));
            }
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
                const float wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wreg[h]), cc + u));
                acc[h][0] = fmaf(wv, v0, acc[h][0]); acc[h][1] = fmaf(wv, v1, acc[h][1]);
                acc[h][2] = fmaf(wv, v2, acc[h][2]); acc[h][3] = fmaf(wv, v3, acc[h][3]);
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
    // fixed-order tree over the NW channel slices
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

// ---- pass 3: g_h = sum_p a_h(p) f_p, v_mfma_f32_16x16x4_f32 with the A operand widened from one
// 16-B load per lane and step (8 pixels): lane (ci, kq) covers pixels 32 kb + 8 kq .. +7.
constexpr int kGatherCh = 64;

template <int DT>
__global__ __launch_bounds__(256) void k_img_gather16(const unsigned short *__restrict__ img, int in_dim,
                                                      int hw, int heads, int KT2p, float *__restrict__ gbuf)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hwp = (hw + 3) & ~3;
    float *a_s = sm;                                        // [heads][hwp], zero padded
    const int chunks = in_dim / kGatherCh;
    const int im = blockIdx.x / chunks, c0 = (blockIdx.x - im * chunks) * kGatherCh;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int ci = lane & 15, kq = lane >> 4;
    const int cb = c0 + wid * 16;
    const unsigned short *row = img + ((size_t)im * in_dim + cb + ci) * hw;
    const bool live = ci < heads;
    const float *arow = a_s + (size_t)(live ? ci : 0) * hwp;
    const int nkb = hw >> 5;
    constexpr int PRE = 4;
    u32x4 pf[PRE];
#pragma unroll
    for (int kb = 0; kb < PRE; ++kb)
        pf[kb] = *reinterpret_cast<const u4u2 *>(row + 32 * min(kb, nkb - 1) + 8 * kq);
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
    auto step = [&](int kb, const u32x4 &d) {
        const int p0 = 32 * kb + 8 * kq;
        float fv[8];
        widen8<DT>(d, fv);
        float4 b0 = *reinterpret_cast<const float4 *>(arow + p0);
        float4 b1 = *reinterpret_cast<const float4 *>(arow + p0 + 4);
        if (!live) { b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = b0; }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[0], b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[1], b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[2], b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[3], b0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[4], b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[5], b1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[6], b1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[7], b1.w, acc, 0, 0, 0);
    };
#pragma unroll
    for (int kb = 0; kb < PRE; ++kb)
        if (kb < nkb) step(kb, pf[kb]);
#pragma unroll 3
    for (int kb = PRE; kb < nkb; ++kb) {
        const u32x4 d = *reinterpret_cast<const u4u2 *>(row + 32 * kb + 8 * kq);
        step(kb, d);
    }
    for (int pp = 32 * nkb; pp < hw; pp += 4) {             // pixel tail (1 pixel for 15 x 15)
        const int p = pp + kq;
        const float fa = p < hw ? widen1<DT>(row[p]) : 0.0f;
        const float ba = (p < hw && live) ? arow[p] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, ba, acc, 0, 0, 0);
    }
    if (live) {
        float *dst = gbuf + ((size_t)im * heads + ci) * KT2p + cb + 4 * kq;
        *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
int launch_img_mean16(const void *img, int dt, int nimg, int in_dim, int hw, float *fm, hipStream_t st, uint32_t *gate, uint32_t gate_seq)
{
    PTX_REQUIRE(in_dim % 8 == 0, "img mean: in_dim=%d must be a multiple of 8", in_dim);
    PTX_REQUIRE(hw >= 8 && (hw >> 3) + ((hw & 7) ? 1 : 0) <= 32, "half-precision image features: hw=%d (supported: 8..255)", hw);
    const int ngroups = nimg * (in_dim / 8);
    const unsigned short *p = static_cast<const unsigned short *>(img);
    const dim3 grid(cdiv(ngroups, 4 * kMeanGroups));
    // (r03: 26 KB of unused LDS per work-group -- six resident work-groups per CU instead of seven, so that the clustering
    //  stream's k_minmax finds a slot at once instead of waiting for work-groups of this launch to retire -- shortens k_minmax
    //  17 -> 13 us, but the step by 0.4 % over five alternating pairs of runs, and costs 0.6 % at 32 scenes: not kept)
    if (dt == 1) hipLaunchKernelGGL((k_img_mean16<1>), grid, dim3(256), 0, st, p, ngroups, hw, fm, gate, gate_seq);
    else         hipLaunchKernelGGL((k_img_mean16<2>), grid, dim3(256), 0, st, p, ngroups, hw, fm, gate, gate_seq);
    PTX_LAUNCHED("k_img_mean16");
    return PTX_OK;
}

int launch_img_scores16(const void *img, int dt, const float *we, const float *qkv0, int nimg, int in_dim,
                        int hw, int heads, int C, int KT1, int KT2p, float scale, float *gbuf, hipStream_t st)
{
    const unsigned short *p = static_cast<const unsigned short *>(img);
    PTX_REQUIRE((heads == 4 || heads == 8 || heads == 16) && in_dim % (8 * kScoreWaves) == 0 && in_dim / kScoreWaves <= 256,
                "img scores: heads=%d in_dim=%d unsupported", heads, in_dim);
    PTX_REQUIRE(hw >= 4 && (hw >> 2) + ((hw & 3) ? 1 : 0) <= 64, "img scores: hw=%d (supported: 4..256 pixels)", hw);
    const size_t lds = sizeof(float) * ((size_t)(kScoreWaves / 2) * heads * 256 + (size_t)heads * (hw + 1));
    PTX_REQUIRE(lds <= 160 * 1024, "img scores: %zu B of LDS", lds);
#define PTX_SCORES16(H_, DT_)                                                                                                       \
    do {                                                                                                                          \
        if (lds > 64 * 1024)                                                                                                      \
            PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_img_scores16<H_, DT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_img_scores16<H_, DT_>), dim3(nimg), dim3(kScoreWaves * 64), lds, st, p, we, qkv0, in_dim, hw, C, KT1, KT2p, scale, gbuf); \
    } while (0)
    if (dt == 1) { if (heads == 4) PTX_SCORES16(4, 1); else if (heads == 8) PTX_SCORES16(8, 1); else PTX_SCORES16(16, 1); }
    else { if (heads == 4) PTX_SCORES16(4, 2); else if (heads == 8) PTX_SCORES16(8, 2); else PTX_SCORES16(16, 2); }
#undef PTX_SCORES16
    PTX_LAUNCHED("k_img_scores16");
    return PTX_OK;
}

int launch_img_gather16(const void *img, int dt, int nimg, int in_dim, int hw, int heads, int KT2p, float *gbuf,
                        hipStream_t st)
{
    PTX_REQUIRE(in_dim % kGatherCh == 0 && heads <= kMaxHeads, "img gather: in_dim=%d heads=%d", in_dim, heads);
    PTX_REQUIRE(hw <= 256, "img gather: hw=%d (max 256 pixels)", hw);
    const unsigned short *p = static_cast<const unsigned short *>(img);
    const int hwp = (hw + 3) & ~3;
    const size_t lds = sizeof(float) * (size_t)heads * hwp;
    const dim3 grid(nimg * (in_dim / kGatherCh));
    if (dt == 1) hipLaunchKernelGGL(k_img_gather16<1>, grid, dim3(256), lds, st, p, in_dim, hw, heads, KT2p, gbuf);
    else         hipLaunchKernelGGL(k_img_gather16<2>, grid, dim3(256), lds, st, p, in_dim, hw, heads, KT2p, gbuf);
    PTX_LAUNCHED("k_img_gather16");
    return PTX_OK;
}

}  // namespace ptx
