// The three image-streaming kernels of imgproxy.hip for image features STORED as bf16 or fp16
// (an AMP backbone hands over half-precision feature maps; BASELINE config 2 names bf16).  Only the
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
constexpr int kMaxHeads = 8;

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

// ---- pass 1: per-channel means.  One wave per 8 rows (4 load instructions of 2 rows each).
template <int DT>
__global__ __launch_bounds__(256) void k_img_mean16(const unsigned short *__restrict__ img, int ngroups,
                                                    int hw, float *__restrict__ fm)
{
    const int lane = lane_id();
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= ngroups) return;
    const RowLanes rl(lane, hw);
    const unsigned short *base = img + (size_t)g * 8 * hw;
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        s[u] = 0.0f;
        if (rl.act) {
            const u32x4 d = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(base + (size_t)(2 * u + rl.half) * hw + rl.poff));
            float v[8];
            widen8<DT>(d, v);
#pragma unroll
            for (int c = 0; c < 8; ++c) s[u] += c >= rl.cfirst ? v[c] : 0.0f;
        }
    }
    const float inv = 1.0f / (float)hw;
    float out = 0.0f;                                       // lane r (r < 8) ends up with the mean of row r
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float v = s[u];
        v += PTX_ROR_F(v, 8); v += PTX_ROR_F(v, 4); v += PTX_ROR_F(v, 2); v += PTX_ROR_F(v, 1);   // 16-lane rows
        const float lo = PTX_LANE_F(v, 0) + PTX_LANE_F(v, 16), hi = PTX_LANE_F(v, 32) + PTX_LANE_F(v, 48);
        if (lane == 2 * u) out = lo * inv;
        if (lane == 2 * u + 1) out = hi * inv;
    }
    if (lane < 8) fm[(size_t)g * 8 + lane] = out;
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
    const int cper = in_dim / NW, cbeg = wid * cper;        // cper <= 64 (validated by the host)
    float wreg[HEADS];
#pragma unroll
    for (int h = 0; h < HEADS; ++h) wreg[h] = lane < cper ? wim[(size_t)h * KT1 + cbeg + lane] : 0.0f;
    // token 0: s_h(0) = scale * q_h . k0_h -- wave h computes head h up front (one load round trip that
    // overlaps the first image loads; as a 32-step scalar loop after the stream it was a chain of
    // dependent round trips that cost ~15 us per launch)
    if (wid < heads) {
        const int hd = C / heads;
        const float *q = qkv0 + (size_t)im * 3 * C + wid * hd;
        const float qk = lane < hd ? q[lane] * q[C + lane] : 0.0f;
        const float s0 = wave_sum(qk);
        if (lane == 0) S[wid * (hw + 1)] = s0 * scale;
    }
    // branch-free inner loop (see k_img_scores): lanes beyond the row re-read lane 0's pixels
    constexpr int UNR = 8;
    const unsigned short *fl = f + (vec ? poff : 0);
    for (int cc = 0; cc < cper; cc += UNR) {               // cper is a multiple of 8 (validated by the host)
        u32x2 d[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            d[u] = __builtin_nontemporal_load(reinterpret_cast<const u2u2 *>(fl + (size_t)(cbeg + cc + u) * hw));
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
    for (int i = tid; i < heads * hwp; i += 256) {
        const int h = i / hwp, p = i - h * hwp;
        a_s[i] = p < hw ? gbuf[((size_t)im * heads + h) * KT2p + in_dim + 1 + p] : 0.0f;
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

// ---- between pass 2 and pass 3 (bf16 path): softmax of the raw scores and everything the gather
// work-groups would otherwise each redo in their prologue.  One work-group per image, thread = token
// (0 = the mean token, t >= 1 = pixel t - 1).  Writes a_h to the [g_h | a_h] rows of gbuf (the o-projection
// GEMM reads them) and the LDS image of k_img_gather_bf: the exact three-way bf16 split of a_h(p).
// (With the softmax in the gather prologue, a chain of 16 loads and two block reductions in front of
// every work-group's first MFMA, the gather took 63-70 us instead of 50.)
constexpr int kGbPad = 264;             // row stride in elements (256 pixels + 8)
constexpr int kApartsLd = 3 * kMaxHeads * kGbPad + 3 * kMaxHeads * 8;   // + the straddling-window fragments

__global__ __launch_bounds__(512) void k_img_softmax_bf(const float *__restrict__ sraw, int in_dim, int hw, int KT2p,
                                                        float *__restrict__ gbuf, unsigned short *__restrict__ aparts)
{
    // one wave per head, lane l owns tokens l, l + 64, l + 128, l + 192: wave reductions only, no barrier
    // (as thread = token with two block reductions the launch took 11-13 us)
    constexpr int heads = kMaxHeads;
    const int im = blockIdx.x, lane = lane_id(), h = threadIdx.x >> 6;
    unsigned short *parts = aparts + (size_t)im * kApartsLd;
    unsigned short *etail = parts + 3 * heads * kGbPad;
    const int estart = hw & ~7;                                 // first pixel of the window that straddles the row end
    const float *s0 = sraw + ((size_t)im * 2 * heads + h) * kSrawLd, *s1 = s0 + (size_t)heads * kSrawLd;
    float sv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = lane + 64 * u;
        sv[u] = t <= hw ? s0[t] + s1[t] : -INFINITY;           // the two channel halves of pass 2
    }
    const float mx = wave_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
    float ex[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ex[u] = lane + 64 * u <= hw ? expf(sv[u] - mx) : 0.0f;
    const float inv = 1.0f / wave_sum((ex[0] + ex[1]) + (ex[2] + ex[3]));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = lane + 64 * u, p = t - 1;
        const float a = ex[u] * inv;
        if (t < KT2p - in_dim) gbuf[((size_t)im * heads + h) * KT2p + in_dim + t] = a;   // [g_h | a_h] row of the o GEMM
        const unsigned int u1 = __float_as_uint(a) & 0xffff0000u;
        const float r1 = a - __uint_as_float(u1);
        const unsigned int u2 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(u2);
        unsigned short q1 = (unsigned short)(u1 >> 16), q2 = (unsigned short)(u2 >> 16),
                       q3 = (unsigned short)(__float_as_uint(r2) >> 16);
        const int pp = p < 0 ? 255 : p;                         // token 0 clears the last padding pixel instead
        if (p < 0) q1 = q2 = q3 = 0;
        parts[(0 * heads + h) * kGbPad + pp] = q1;
        parts[(1 * heads + h) * kGbPad + pp] = q2;
        parts[(2 * heads + h) * kGbPad + pp] = q3;
        // the window that straddles the row end is loaded from pixel hw - 8 in the last image (see
        // k_img_gather_bf): its B fragment holds the probabilities from `estart` on, zero before
        const int e = p - (hw - 8);
        if (p >= 0 && e >= 0 && e < 8) {
            const bool own = p >= estart;
            etail[(0 * heads + h) * 8 + e] = own ? q1 : (unsigned short)0;
            etail[(1 * heads + h) * 8 + e] = own ? q2 : (unsigned short)0;
            etail[(2 * heads + h) * 8 + e] = own ? q3 : (unsigned short)0;
        }
    }
    // padding columns 256..263 of each row are never read (windows end at pixel 255)
}

// ---- pass 3 for bf16 features on the bf16 matrix pipe, still exact to fp32: the probabilities are split
// a = a1 + a2 + a3 into three bf16 parts (8 significant bits each, the split is exact), every product
// a_i * f of two bf16 values is exact in fp32, and v_mfma_f32_16x16x32_bf16 accumulates in fp32 -- so
// the result is an fp32 dot product in a different summation order, at 1/16 of the f32-MFMA cost and
// with no widening VALU work: the 16-B load of 8 pixels IS the A fragment.
//
// The cheap MFMA is spent on a fragment map the memory pipe likes.  Measured with loads only
// (scratch/pattern_bench.hip, 180 MB of 450-B rows): a wave instruction streams at full rate (24.5 us) only
// if ADJACENT LANES read ADJACENT 16-B chunks in groups of four (one 64-B request per quad); the natural
// MFMA map (lane & 15 = channel row, so neighbouring lanes are 450 B apart) runs at half that (45.5 us),
// whatever the length of the per-row runs.  So one MFMA here covers FOUR channel rows x 128 pixels:
//   A row  m = (channel m >> 2, pixel set m & 3)      B column n = (head n >> 2, pixel set n & 3)
//   pixel of (set, K-lane kq, element i) = 128 * step + 32 * kq + 8 * set + i
// lane (m, kq) loads 16 B at that pixel: a quad reads 64 contiguous bytes, 16 lanes read 256.  Only the
// output blocks whose row and column sets agree mean anything (1/4 of the MFMA, which is cheap enough);
// they sit in acc[lane & 3] and are summed over the sets with two quad permutes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NG>                       // 16-channel groups per wave; a work-group covers 64 * NG channels
__global__ __launch_bounds__(256) void k_img_gather_bf(const unsigned short *__restrict__ img, int nimg, int in_dim,
                                                       int hw, int KT2p,
                                                       const unsigned short *__restrict__ aparts,
                                                       float *__restrict__ gbuf)
{
    constexpr int heads = kMaxHeads;
    // [part][head][pixel], zero beyond hw; then [part][head][8] for the window that straddles the row end
    __shared__ __attribute__((aligned(16))) unsigned short parts[3 * heads * kGbPad + 3 * heads * 8];
    unsigned short *etail = parts + 3 * heads * kGbPad;
    const int chunk = 64 * NG, chunks = in_dim / chunk;
    const int im = blockIdx.x / chunks, c0 = (blockIdx.x - im * chunks) * chunk;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int m = lane & 15, kq = lane >> 4, cr = m >> 2, ps = m & 3;
    const int nsteps = hw > 128 ? 2 : 1;
    const bool nat = im != nimg - 1;
    // this lane's pixel window per step: inside the row, straddling its end, or beyond it
    int aoff[2], boff[2];                                       // element offsets: into the row / into `parts`
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int px = 128 * st + 32 * kq + 8 * ps;
        const bool full = px + 8 <= hw, edge = !full && px < hw;
        // a window that sticks out of the row runs on into the next row (contiguous; finite values times
        // the zero padding of `parts`); only the last image of the tensor re-reads its last window instead
        aoff[st] = (full || nat) ? px : hw - 8;
        boff[st] = (edge && !nat) ? -1 : px;
    }
    // group g of this wave = channels c0 + (4 g + wid) * 16 .. +15; tile t, row cr = channel 4 cr + t of them
    const unsigned short *wrow = img + ((size_t)im * in_dim + c0 + wid * 16 + 4 * cr) * hw;
    const size_t gstride = (size_t)64 * hw;
    u32x4 pf[2][4][2];
    auto fetch = [&](int buf, const unsigned short *r) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int st = 0; st < 2; ++st)
                pf[buf][t][st] = *reinterpret_cast<const u4u2 *>(r + (size_t)t * hw + aoff[st < nsteps ? st : 0]);
    };
    fetch(0, wrow);
    {   // the prepared probabilities of this image (k_img_softmax_bf) -> LDS: one round trip
        const u32x4 *src = reinterpret_cast<const u32x4 *>(aparts + (size_t)im * kApartsLd);
        u32x4 *dst = reinterpret_cast<u32x4 *>(parts);
        constexpr int NV = kApartsLd / 8;                       // 16-B vectors
        u32x4 v[(NV + 255) / 256];
#pragma unroll
        for (int i = 0; i < (NV + 255) / 256; ++i) if (tid + 256 * i < NV) v[i] = src[tid + 256 * i];
#pragma unroll
        for (int i = 0; i < (NV + 255) / 256; ++i) if (tid + 256 * i < NV) dst[tid + 256 * i] = v[i];
    }
    __syncthreads();
    const int hh = m >> 2;                                      // head within the group for B columns
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int cur = g & 1;
        if (g + 1 < NG) fetch(cur ^ 1, wrow + (size_t)(g + 1) * gstride);
        f32x4 acc[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st < nsteps) {
#pragma unroll
                for (int hg = 0; hg < 2; ++hg) {
                    bf16x8 bfr[3];
#pragma unroll
                    for (int pt = 0; pt < 3; ++pt) {
                        const int row = pt * heads + 4 * hg + hh;
                        const unsigned short *bp = boff[st] >= 0 ? parts + (size_t)row * kGbPad + boff[st]
                                                                 : etail + row * 8;
                        bfr[pt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp));
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const bf16x8 fa = __builtin_bit_cast(bf16x8, pf[cur][t][st]);
#pragma unroll
                        for (int pt = 0; pt < 3; ++pt)
                            acc[t][hg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, bfr[pt], acc[t][hg], 0, 0, 0);
                    }
                }
            }
        }
        // D[row 4 kq + i][col m]: row = (channel kq, set i), col = (head hh, set ps): keep i == ps, sum the quad
#pragma unroll
        for (int hg = 0; hg < 2; ++hg) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float x = ps == 0 ? acc[t][hg][0] : ps == 1 ? acc[t][hg][1] : ps == 2 ? acc[t][hg][2] : acc[t][hg][3];
                x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                v[t] = x;
            }
            if (ps == 0) {       // lane (kq, hh): channels 4 kq + 0..3 of the group, head 4 hg + hh
                float *dst = gbuf + ((size_t)im * heads + 4 * hg + hh) * KT2p + c0 + (4 * g + wid) * 16 + 4 * kq;
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---- pass 2 for bf16 features on the bf16 matrix pipe (exact to fp32 like k_img_gather_bf: here the
// per-image head weights w_h are the operand that is split into three bf16 parts).  The VALU kernel
// above spends 8 FMAs + 1 widening op per element; this one spends half a v_perm per element:
//   s_h(p) = sum_c w_h(c) f(c,p)   as   D[16 x 16] += A[16 x 32 channels] * B[32 channels x 16 pixels]
//   A row m  = w1 of head m (m < 8) | w2 of head m-8; a second MFMA adds w3 (rows 8-15 zero)
//   B col n  = one pixel; K-lane kq holds channels 8 kq .. 8 kq + 7 of the 32-channel block
// Lane (n, kq) loads, for i = 0..7, the 16 B = pixels 8n .. 8n+7 of channel row (block + 8 kq + i): four
// rows x 256 contiguous bytes per instruction, quads contiguous (see k_img_gather_bf).  The 8 x 8 block
// of (channel, pixel) values a lane then holds is transposed in registers with v_perm_b32 into eight
// B fragments (one per pixel j; column n of MFMA j is pixel 8n + j), each with its own accumulator.
//
// Work unit = (image, half of the channels), 4 waves = (pixel tile of 128) x (channel quarter of the half);
// the two quarters are summed through LDS and the unit writes RAW partial scores to `sraw`
// ([image][half][head][token]; half 0 adds the positional terms and token 0).  The gather kernel adds
// the halves and normalises in its prologue, so no unit waits for a partner.  Measured on the way:
//   * whole images as units (784 on 256 CUs: 3.06 per CU, 4 on some, two resident at a time): 62 us, the
//     same as the VALU kernel -- the launch was two rounds of work-group lifetimes long
//   * removing any ONE of {weight prologue, perm + MFMA work, reduction + output} from this kernel leaves it
//     at 50-54 us, removing all three gives 39 (a load-only copy of its structure runs at 33 in isolation,
//     scratch/pattern_bench.hip): what is left is the serial chain inside each of the 2 x 3 rounds of
//     work-groups, not any one resource
//   * (image, pixel tile) units: 53 us, but 1.8x the L1->L2 requests of a whole-row stream (PMC
//     TCP_TCC_READ_REQ 2.9 M vs 1.6 M): every row has a cache line at the tile boundary that both units
//     fetch, and lanes beyond the row, clamped to offset 0, touched one more line per row
constexpr int kSbPad = 520;             // LDS row stride (elements) of the split weights: 16 rows x 16 B hit all banks
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(256) void k_img_scores_bf(
    const unsigned short *__restrict__ img, const float *__restrict__ we, const float *__restrict__ qkv0,
    int nimg, int in_dim, int hw, int C, int KT1, float scale, float *__restrict__ sraw)
{
    constexpr int heads = kMaxHeads;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *red = sm;                                            // [2 tiles][heads][128]
    unsigned short *wpart = reinterpret_cast<unsigned short *>(red + 2 * heads * 128);   // [24][kSbPad]
    // images in reverse order (the mean pass left the last ones in cache)
    const int half = blockIdx.x & 1, im = nimg - 1 - (blockIdx.x >> 1);
    const int tid = threadIdx.x, lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = wid & 1, q = wid >> 1;                        // pixel tile, channel quarter of this half
    const int n = lane & 15, kq = lane >> 4;
    const float *wim = we + (size_t)im * heads * KT1;
    const unsigned short *f = img + (size_t)im * in_dim * hw;
    // this lane's window of 8 pixels: inside the row, straddling its end (loaded from hw - 8) or beyond
    // Windows that stick out of the row simply run on into the next row (rows are contiguous; those lines
    // are the ones the next load instruction needs anyway) and the surplus columns are discarded.  Only in
    // the last image of the tensor, where that would read past the buffer, they re-read the last window.
    const bool nat = im != nimg - 1;
    const int px = 128 * T + 8 * n;
    const bool full = px + 8 <= hw, edge = !full && px < hw;
    const int aoff = (full || nat) ? px : hw - 8;
    const int own_from = (full || nat) ? px : (edge ? (hw & ~7) : 1 << 30);
    const int ch = in_dim / 2, cq = ch / 2, cbeg = half * ch + q * cq, nkb = cq / 32;
    u32x4 L[2][8];
    auto fetch = [&](u32x4 (&Lb)[8], int kb) {
        const unsigned short *r = f + (size_t)(cbeg + 32 * kb + 8 * kq) * hw + aoff;
#pragma unroll
        for (int i = 0; i < 8; ++i) Lb[i] = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(r + (size_t)i * hw));
    };
    fetch(L[0], 0);
    // head weights of this half -> three bf16 parts in LDS (thread = channel, all loads first)
    for (int c = tid; c < ch; c += 256) {
        float wv[heads];
#pragma unroll
        for (int h = 0; h < heads; ++h) wv[h] = wim[(size_t)h * KT1 + half * ch + c];
#pragma unroll
        for (int h = 0; h < heads; ++h) {
            const unsigned int u1 = __float_as_uint(wv[h]) & 0xffff0000u;
            const float r1 = wv[h] - __uint_as_float(u1);
            const unsigned int u2 = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(u2);
            wpart[(size_t)h * kSbPad + c] = (unsigned short)(u1 >> 16);
            wpart[(size_t)(8 + h) * kSbPad + c] = (unsigned short)(u2 >> 16);
            wpart[(size_t)(16 + h) * kSbPad + c] = (unsigned short)(__float_as_uint(r2) >> 16);
        }
    }
    // token 0: s_h(0) = scale * q_h . k0_h with half 0 (wave w computes heads w and w + 4), 0 with half 1
    {
        const int hd = C / heads;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = wid + 4 * hh;
            const float *qv = qkv0 + (size_t)im * 3 * C + h * hd;
            const float qk = (half == 0 && lane < hd) ? qv[lane] * qv[C + lane] : 0.0f;
            const float s0 = wave_sum(qk);
            if (lane == 0) sraw[(((size_t)im * 2 + half) * heads + h) * kSrawLd] = s0 * scale;
        }
    }
    __syncthreads();
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned short *a1p = wpart + (size_t)n * kSbPad + q * cq + 8 * kq;               // rows 0-15: w1 | w2
    const unsigned short *a2p = wpart + (size_t)(16 + (n & 7)) * kSbPad + q * cq + 8 * kq;   // rows 16-23: w3
    const bool lo = n < 8;
    auto block = [&](const u32x4 (&Lb)[8], int kb) {
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(a1p + 32 * kb));
        u32x4 a2u = *reinterpret_cast<const u32x4 *>(a2p + 32 * kb);
        if (!lo) a2u = u32x4{0u, 0u, 0u, 0u};
        const bf16x8 a2 = __builtin_bit_cast(bf16x8, a2u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int sel = (j & 1) ? 0x07060302u : 0x05040100u;
            u32x4 b;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) b[qd] = __builtin_amdgcn_perm(Lb[2 * qd + 1][j >> 1], Lb[2 * qd][j >> 1], sel);
            const bf16x8 bf = __builtin_bit_cast(bf16x8, b);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bf, acc[j], 0, 0, 0);
        }
    };
    for (int kb = 0; kb < nkb; kb += 2) {                       // nkb is even (validated by the host)
        fetch(L[1], kb + 1);
        block(L[0], kb);
        if (kb + 2 < nkb) fetch(L[0], kb + 2);
        block(L[1], kb + 1);
    }
    // rows 8-15 (lanes 32-63) hold the w2 contribution of heads 0-7: fold onto rows 0-7
    float sv[4][8];                                             // [head 4 kq + i][pixel aoff + j], lanes kq < 2
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i][j] = acc[j][i] + __shfl_xor(acc[j][i], 32, 64);
    // positional terms of this lane's pixels (half 0 adds them), requested before the reduction
    float ev[4][8];
    const bool fin = q == 0 && kq < 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
        if (fin && half == 0 && px < hw) {
            const float *ep = wim + (size_t)(4 * kq + i) * KT1 + in_dim + 1 + aoff;
            e0 = *reinterpret_cast<const f32x4u *>(ep);
            e1 = *reinterpret_cast<const f32x4u *>(ep + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { ev[i][j] = e0[j]; ev[i][4 + j] = e1[j]; }
    }
    // fixed-order sum of the two channel quarters of this half, staged through LDS so that the unit
    // writes its scores as whole rows (scattered 4-B stores straight from the MFMA layout were as
    // many memory requests as the whole feature stream)
    if (q == 1 && kq < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = red + ((size_t)T * heads + 4 * kq + i) * 128 + 8 * n;
            *reinterpret_cast<float4 *>(d) = make_float4(sv[i][0], sv[i][1], sv[i][2], sv[i][3]);
            *reinterpret_cast<float4 *>(d + 4) = make_float4(sv[i][4], sv[i][5], sv[i][6], sv[i][7]);
        }
    }
    __syncthreads();
    float *S = reinterpret_cast<float *>(wpart);                // [heads][256] by pixel; the weights are dead
    if (fin) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *r = red + ((size_t)T * heads + 4 * kq + i) * 128 + 8 * n;
            const float4 t0 = *reinterpret_cast<const float4 *>(r), t1 = *reinterpret_cast<const float4 *>(r + 4);
            const float add[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = aoff + j;
                if (p >= own_from && p < hw) S[(4 * kq + i) * 256 + p] = (sv[i][j] + add[j]) + ev[i][j];
            }
        }
    }
    __syncthreads();
    if (tid < hw) {
#pragma unroll
        for (int h = 0; h < heads; ++h)
            sraw[(((size_t)im * 2 + half) * heads + h) * kSrawLd + 1 + tid] = S[h * 256 + tid];
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
int launch_img_mean16(const void *img, int dt, int nimg, int in_dim, int hw, float *fm, hipStream_t st)
{
    PTX_REQUIRE(in_dim % 8 == 0, "img mean: in_dim=%d must be a multiple of 8", in_dim);
    PTX_REQUIRE(hw >= 8 && (hw >> 3) + ((hw & 7) ? 1 : 0) <= 32, "half-precision image features: hw=%d (supported: 8..255)", hw);
    const int ngroups = nimg * (in_dim / 8);
    const unsigned short *p = static_cast<const unsigned short *>(img);
    if (dt == 1) hipLaunchKernelGGL(k_img_mean16<1>, dim3(cdiv(ngroups, 4)), dim3(256), 0, st, p, ngroups, hw, fm);
    else         hipLaunchKernelGGL(k_img_mean16<2>, dim3(cdiv(ngroups, 4)), dim3(256), 0, st, p, ngroups, hw, fm);
    PTX_LAUNCHED("k_img_mean16");
    return PTX_OK;
}

// bf16 features with the shapes of the path take the matrix-pipe pair (raw scores in `sraw`, softmax in
// the gather prologue); anything else (fp16 storage, unusual in_dim / hw) the VALU / f32-MFMA pair
static bool bf_pair(int dt, int in_dim, int hw, int heads)
{
    static const int off = getenv("PTX_IMG16_VALU") ? 1 : 0;
    return !off && dt == 1 && heads == kMaxHeads && in_dim % 512 == 0 && in_dim / 2 <= kSbPad - 8 && hw > 128 && hw <= 255;
}

int launch_img_scores16(const void *img, int dt, const float *we, const float *qkv0, int nimg, int in_dim,
                        int hw, int heads, int C, int KT1, int KT2p, float scale, float *sraw, float *gbuf,
                        hipStream_t st)
{
    const unsigned short *p = static_cast<const unsigned short *>(img);
    if (bf_pair(dt, in_dim, hw, heads)) {
        const size_t lds = sizeof(float) * 2 * heads * 128 + sizeof(unsigned short) * 24 * kSbPad;
        hipLaunchKernelGGL(k_img_scores_bf, dim3(nimg * 2), dim3(256), lds, st, p, we, qkv0, nimg, in_dim, hw, C, KT1,
                           scale, sraw);
        PTX_LAUNCHED("k_img_scores_bf");
        return PTX_OK;
    }
    PTX_REQUIRE(heads == kMaxHeads && in_dim % (8 * kScoreWaves) == 0 && in_dim / kScoreWaves <= 64,
                "img scores: heads=%d in_dim=%d unsupported", heads, in_dim);
    PTX_REQUIRE(hw >= 4 && (hw >> 2) + ((hw & 3) ? 1 : 0) <= 64, "img scores: hw=%d (supported: 4..256 pixels)", hw);
    const size_t lds = sizeof(float) * ((size_t)(kScoreWaves / 2) * heads * 256 + (size_t)heads * (hw + 1));
    PTX_REQUIRE(lds <= 64 * 1024, "img scores: %zu B of LDS", lds);
    if (dt == 1)
        hipLaunchKernelGGL((k_img_scores16<kMaxHeads, 1>), dim3(nimg), dim3(kScoreWaves * 64), lds, st, p, we, qkv0,
                           in_dim, hw, C, KT1, KT2p, scale, gbuf);
    else
        hipLaunchKernelGGL((k_img_scores16<kMaxHeads, 2>), dim3(nimg), dim3(kScoreWaves * 64), lds, st, p, we, qkv0,
                           in_dim, hw, C, KT1, KT2p, scale, gbuf);
    PTX_LAUNCHED("k_img_scores16");
    return PTX_OK;
}

size_t img16_aparts_bytes(int nimg) { return (size_t)nimg * kApartsLd * sizeof(unsigned short); }

int launch_img_gather16(const void *img, int dt, int nimg, int in_dim, int hw, int heads, int KT2p,
                        const float *sraw, unsigned short *aparts, float *gbuf, hipStream_t st)
{
    PTX_REQUIRE(in_dim % kGatherCh == 0 && heads <= kMaxHeads, "img gather: in_dim=%d heads=%d", in_dim, heads);
    const unsigned short *p = static_cast<const unsigned short *>(img);
    if (bf_pair(dt, in_dim, hw, heads)) {
        hipLaunchKernelGGL(k_img_softmax_bf, dim3(nimg), dim3(512), 0, st, sraw, in_dim, hw, KT2p, gbuf, aparts);
        PTX_LAUNCHED("k_img_softmax_bf");
        static const int ng = getenv("PTX_GATHER_NG") ? atoi(getenv("PTX_GATHER_NG")) : 4;
#define PTX_GN(N_) case N_: hipLaunchKernelGGL(k_img_gather_bf<N_>, dim3(nimg * (in_dim / (64 * N_))), dim3(256), 0, st, p, nimg, in_dim, hw, KT2p, aparts, gbuf); break;
        switch (ng) { PTX_GN(1) PTX_GN(2) PTX_GN(8) default: PTX_GN(4) }
#undef PTX_GN
        PTX_LAUNCHED("k_img_gather_bf");
        return PTX_OK;
    }
    const int hwp = (hw + 3) & ~3;
    const size_t lds = sizeof(float) * (size_t)heads * hwp;
    const dim3 grid(nimg * (in_dim / kGatherCh));
    if (dt == 1) hipLaunchKernelGGL(k_img_gather16<1>, grid, dim3(256), lds, st, p, in_dim, hw, heads, KT2p, gbuf);
    else         hipLaunchKernelGGL(k_img_gather16<2>, grid, dim3(256), lds, st, p, in_dim, hw, heads, KT2p, gbuf);
    PTX_LAUNCHED("k_img_gather16");
    return PTX_OK;
}

}  // namespace ptx
