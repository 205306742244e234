// ProxyAttention (PRE:230-250) of one (scene, head, branch) in ONE work-group and one launch, head_dim 32:
//   stage A   proxy as query : PV = softmax_n((P scale) K^T) V           queries = Lp proxies, keys = n cluster tokens
//   stage B   proxy as key   : O  = softmax_L(mask((Q scale) P^T)) PV    queries = n tokens,   keys = Lp proxies
// r02 ran these as two launches of k_attn32 (attn.hip) on the fp32 matrix instruction with PV making a round trip through
// HBM: 28 us at 4 scenes per GPU, 113 us at 32 (0.12 / 0.25 of the fp32 matrix peak).  Here
//   * both contractions run on the bf16 pipe through the three-way operand split of split3.h (six
//     v_mfma_f32_32x32x16_bf16 per 16 k: an fp32 product in another summation order, 3/8 of the matrix cycles);
//   * the projected proxies P (keys of stage B, queries of stage A) and PV^T live in LDS as pre-split bf16 planes, so
//     the only per-tile split work is the 16 probabilities a lane holds anyway;
//   * K and V of up to 256 keys are split ONCE (each wave its own key tile, as the MFMA fragments it loaded) into LDS
//     planes; in stage A a wave owns a (query tile, key slice) pair and runs its online soft-max over the slice without
//     any barrier -- so the two waves of a SIMD drift apart and one's VALU work (soft-max, splitting the probabilities)
//     runs under the other's matrix instructions; the slices of a query tile are merged once at the end and PV^T is
//     written in the lane order stage B's MFMA wants, over the bytes K / V occupied.  (First version, r03: one key tile
//     per WAVE held in registers for all query tiles, partials merged per query tile behind a barrier: 32 us per launch
//     at 4 scenes, 56 us at 32 -- the barriers kept all eight waves in the same phase, matrix and VALU time added.)
//   * the S^T = K Q^T orientation of attn.hip is kept: a lane owns one query column and 16 key rows, so the softmax
//     reductions are in-lane plus one xor-32 exchange and P never moves between lanes before the second contraction.
// Larger n (the reference's gs = 12: 691 tokens) stage their keys in chunks of 256 with the running soft-max carried
// across chunks; the next chunk's fragments are in flight in registers while the current one is consumed.
#include <cstdlib>

#include "common.h"
#include "split3.h"

namespace ptx {

constexpr int kFaWaves = 8;
constexpr float kLog2e = 1.4426950408889634f;

// key (0..31 inside a tile) held in accumulator register r of half hh: the C / D layout of the 32x32 MFMA
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
// position of key kappa (0..31) inside its tile in the PV^T rows: the 8 keys (step t, k-block hh, j) of a lane's B / A
// fragment are contiguous -- kappa = (j & 3) + 8 (2 t + (j >> 2)) + 4 hh  <->  pos = 16 t + 8 hh + j
__device__ __forceinline__ int pv_pos(int kappa)
{
    const int g = kappa >> 3, hh = (kappa >> 2) & 1;
    return 16 * (g >> 1) + 8 * hh + (kappa & 3) + 4 * (g & 1);
}

// all-reduce over the two halves of a wave (lane ^ 32) on the VALU: v_permlane32_swap leaves {lo, lo} and {hi, hi}
// (__shfl_xor is a ds_bpermute: an LDS round trip in the middle of the soft-max chain)
__device__ __forceinline__ float half_max(float x)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

constexpr int kFaChunk = 32 * kFaWaves;         // keys staged per pass of stage A: one key tile per wave

// LDS geometry, fixed (so that every fragment address is one per-lane base + an immediate): planes of 256 rows x 64 B for
// the projected proxies and the staged K, rows of 256 elements + 16 B for V^T / PV^T (rows 4 banks apart: 16-byte reads
// of sixteen rows are conflict-free)
constexpr int kFaRowPlane = kFaChunk * XROW;                // 16384
constexpr int kFaTStride = kFaChunk * 2 + 16;               // 528
constexpr int kFaTPlane = 32 * kFaTStride;                  // 16896
constexpr int kFaBiasOff = 3 * kFaRowPlane;                 // key bias of stage B (256 floats)
constexpr int kFaRegOff = kFaBiasOff + kFaChunk * 4;        // {K, V^T} during stage A, {PV^T, merge slots} afterwards
constexpr int kFaSlotOff = kFaRegOff + 3 * kFaTPlane;
constexpr int kFaLds = kFaRegOff + 3 * kFaRowPlane + 3 * kFaTPlane;
static_assert(kFaSlotOff + kFaWaves * 18 * 64 * 4 <= kFaLds && kFaLds <= 160 * 1024, "LDS carve of k_proxy_attn");

template <int NP>
__device__ __forceinline__ void mfma_one_chain(const bf16x8 (&a)[2][3], const bf16x8 (&b)[2][3], f32x16 &c)
{
    c = mfma_parts<NP>(a[0], b[0], c);
    c = mfma_parts<NP>(a[1], b[1], c);
}

// One soft-max step of a 32 x 32 score tile held as sc[r] (this lane's query, 16 of the 32 keys; -inf = no key): running
// maximum mb (already times c1) and sum l, rescale factor of the running output, probabilities split for the MFMA
template <int NP>
__device__ __forceinline__ float softmax_tile(f32x16 &sc, float c1, float &mb, float &l, bf16x8 (&pb)[2][3])
{
    float tmax = fmaxf(sc[0], sc[1]);
#pragma unroll
    for (int r = 2; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
    tmax = half_max(tmax);                                          // finite: every tile has a valid key
    const float mb_new = fmaxf(mb, tmax * c1);
    const float alpha = __builtin_amdgcn_exp2f(mb - mb_new);        // first tile: exp2(-inf) = 0
    float psum = 0.0f, e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { e[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], c1, -mb_new)); psum += e[r]; }
    l = fmaf(l, alpha, half_sum(psum));
    mb = mb_new;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = e[8 * s + j];
        frag_parts<NP>(x, pb[s]);
    }
    return alpha;
}

// Wave roles in stage A: NQ query tiles (proxies) x NS key slices; wave w works on query tile w / NS with the key tiles
// [sl * TPS, (sl + 1) * TPS) of every staged chunk, sl = w % NS.  NS = 8 / NQ: all eight waves are busy for 1, 2, 4 or 8
// query tiles, seven for 7 (the benchmark's 196 image proxies), six for 3.
//
// Where the time goes (r03, s_memtime stamps per phase + PMC, 4 scenes x 196 proxies: 27 us per launch, 60 k cycles per
// wave): a (score, soft-max, second contraction) step of one wave takes ~2900 cycles while its SIMD partner runs the same
// program -- matrix time (24 MFMAs x 32 cycles) and VALU time (~200 instructions x 4.3 cycles) of the two waves of a SIMD
// ADD (VALU busy 62 % + matrix busy 36 % of the cycles): an in-order wave stalls on the soft-max until its score MFMAs
// have drained, and the partner's VALU work does not slip under another wave's matrix instructions.  Tried and
// measured, all SLOWER or equal: two interleaved accumulator chains per contraction (27.5 us: the chain was not the
// stall); a barrier-paced ping-pong in which one wave group runs its matrix segment (second contraction of tile t +
// scores of tile t + 1) while the other runs its VALU segment (34 us: every phase takes ~1900 cycles whichever
// segment a wave is in); the first version's per-query-tile merges behind barriers (32 us).  What would help is an
// in-wave software pipeline with the soft-max instructions placed BETWEEN the matrix instructions of the neighbouring
// tiles in program order (<= 5 per MFMA slot) -- not done.
template <int NP>       // 3: split operands (fp32-equivalent); 1: plain bf16 operands (compute_dtype = 1)
__global__ __launch_bounds__(kFaWaves * 64) void k_proxy_attn(FAttnBatch ab)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FAttnProb p = ab.p[blockIdx.y];
    const int b = blockIdx.x / ab.heads, h = blockIdx.x - b * ab.heads;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int n = ab.n, C = ab.C, Lp = p.Lp;
    const int NQ = (Lp + 31) >> 5, LpPad = NQ * 32, NKT = (n + 31) >> 5;
    char *Pk = smem;
    float *kbias = reinterpret_cast<float *>(smem + kFaBiasOff);
    char *Kp = smem + kFaRegOff, *Vt = Kp + 3 * kFaRowPlane;        // stage A
    char *PVt = smem + kFaRegOff;                                   // after stage A (same bytes)
    float *slots = reinterpret_cast<float *>(smem + kFaSlotOff);
    const float *qkv = p.qkv + (size_t)b * n * 3 * C + h * 32;
    const float c1 = ab.scale * kLog2e;         // soft-max exponents in base 2: exp(scale (s - m)) = exp2(c1 s - c1 m)
    // per-lane bases of the fragment reads: rows of a row-major plane (two 16-byte pieces, swizzled), rows of a transposed one
    const int rowA = li * XROW + xswz(li, hh), rowB = li * XROW + xswz(li, 2 + hh);     // + tile * 2048 + plane * kFaRowPlane
    const int trow = li * kFaTStride + 16 * hh;                                         // + tile * 64 + 32 s + plane * kFaTPlane
    auto read_rows = [&](const char *base, int tile, bf16x8 (&f)[2][3]) {
#pragma unroll
        for (int pt3 = 0; pt3 < NP; ++pt3) {
            f[0][pt3] = *reinterpret_cast<const bf16x8 *>(base + rowA + tile * (32 * XROW) + pt3 * kFaRowPlane);
            f[1][pt3] = *reinterpret_cast<const bf16x8 *>(base + rowB + tile * (32 * XROW) + pt3 * kFaRowPlane);
        }
    };
    auto read_trans = [&](const char *base, int tile, bf16x8 (&f)[2][3]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pt3 = 0; pt3 < NP; ++pt3)
                f[s][pt3] = *reinterpret_cast<const bf16x8 *>(base + trow + tile * 64 + 32 * s + pt3 * kFaTPlane);
    };

    // K / V fragments of key tile kt as the MFMAs take them: kf = K[key li][dims 16 s + 8 hh ..], vf = V[key(t, hh, j)][dim li]
    auto load_kv = [&](int kt, float (&kf)[16], float (&vf)[16]) {
        const int key = min(32 * kt + li, n - 1);
        const float4 *kr = reinterpret_cast<const float4 *>(qkv + (size_t)key * 3 * C + C + 8 * hh);
        const float4 k0 = kr[0], k1 = kr[1], k2 = kr[4], k3 = kr[5];
        kf[0] = k0.x; kf[1] = k0.y; kf[2] = k0.z; kf[3] = k0.w; kf[4] = k1.x; kf[5] = k1.y; kf[6] = k1.z; kf[7] = k1.w;
        kf[8] = k2.x; kf[9] = k2.y; kf[10] = k2.z; kf[11] = k2.w; kf[12] = k3.x; kf[13] = k3.y; kf[14] = k3.z; kf[15] = k3.w;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 32 * kt + (j & 3) + 8 * (2 * t + (j >> 2)) + 4 * hh;
                vf[8 * t + j] = kk < n ? qkv[(size_t)kk * 3 * C + 2 * C + li] : 0.0f;
            }
    };
    // ... split and stored as local tile `lt` of the staged chunk
    auto stash_kv = [&](int lt, const float (&kf)[16], const float (&vf)[16]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float x[8], y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { x[i] = kf[8 * s + i]; y[i] = vf[8 * s + i]; }
            bf16x8 fk[3], fv[3];
            frag_parts<NP>(x, fk);
            frag_parts<NP>(y, fv);
#pragma unroll
            for (int pt3 = 0; pt3 < NP; ++pt3) {
                *reinterpret_cast<bf16x8 *>(Kp + (s ? rowB : rowA) + lt * (32 * XROW) + pt3 * kFaRowPlane) = fk[pt3];
                *reinterpret_cast<bf16x8 *>(Vt + trow + lt * 64 + 32 * s + pt3 * kFaTPlane) = fv[pt3];
            }
        }
    };
    // the query rows of stage B's round `rnd` of this wave (tokens 32 (wv + 8 rnd) + li), requested long before they are split
    auto load_q = [&](int rnd, float4 (&q)[4]) {
        const int tok = min(32 * (wv + kFaWaves * rnd) + li, n - 1);
        const float4 *qr = reinterpret_cast<const float4 *>(qkv + (size_t)tok * 3 * C + 8 * hh);
        q[0] = qr[0]; q[1] = qr[1]; q[2] = qr[4]; q[3] = qr[5];
    };
    float kf[16], vf[16];
    if (wv < NKT) load_kv(wv, kf, vf);                      // requested first: in flight under the proxy staging

    // ---- the projected proxies of this (scene, head) -> three bf16 planes in LDS; key bias of stage B
    {
        const float *pt = p.pt + (size_t)b * Lp * C + h * 32;
        for (int e = tid; e < LpPad * 8; e += kFaWaves * 64) {
            const int row = e >> 3, kq = (e & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < Lp) v = *reinterpret_cast<const float4 *>(pt + (size_t)row * C + kq);
            stash_parts<NP>(Pk + row * XROW + xswz(row, kq >> 3) + (kq & 4) * 2, kFaRowPlane, v);
        }
        const float masked = -1e9f / ab.scale;              // masked_fill(-1e9) of the SCALED scores (PRE:247), in raw units
        for (int k = tid; k < LpPad; k += kFaWaves * 64) {
            float kb = k >= Lp ? -INFINITY : 0.0f;          // 0: valid key; -inf: beyond the proxies
            if (k < Lp && p.mask != nullptr && p.mask[(size_t)b * Lp + k] == 0) kb = masked;
            kbias[k] = kb;
        }
    }

    // ---- stage A: PV = softmax_n((P scale) K^T) V, no mask (PRE:232-238)
    const int NS = kFaWaves / min(NQ, kFaWaves), TPS = kFaWaves / NS;
    const int qtA = wv / NS, slA = wv - qtA * NS;
    const bool activeA = qtA < NQ;
    float mb_run = -INFINITY, l_run = 0.0f;         // running maximum (times c1) and sum of this wave's (query tile, key slice)
    f32x16 oA;
#pragma unroll
    for (int i = 0; i < 16; ++i) oA[i] = 0.0f;
    bf16x8 qb[2][3], ka[2][3], va[2][3], pb[2][3];
    f32x16 sc;
    for (int c0 = 0; c0 < NKT; c0 += kFaWaves) {            // chunks of 8 key tiles staged through LDS
        if (c0 > 0) {
            // (a register prefetch of the next chunk's fragments under the current one costs 32 live registers in a kernel
            //  that has none to spare: larger scenes pay one exposed load per 256 keys instead)
            load_kv(min(c0 + wv, NKT - 1), kf, vf);         // unconditional: the old values are dead for the allocator too
            __syncthreads();                                // every wave is done with the previous chunk
        }
        if (c0 + wv < NKT) stash_kv(wv, kf, vf);
        __syncthreads();
        if (c0 == 0 && activeA) read_rows(Pk, qtA, qb);
        const int NT = min(kFaWaves, NKT - c0);                                  // tiles of this chunk
        const int lt0 = slA * TPS, lt1 = activeA ? min(lt0 + TPS, NT) : 0;
        // fragment reads run one phase ahead of their MFMAs: V^T of tile lt is requested before its scores are computed, K of
        // tile lt + 1 (into the same registers, dead by then) before the second contraction of tile lt
        if (lt0 < lt1) read_rows(Kp, lt0, ka);
        for (int lt = lt0; lt < lt1; ++lt) {
            read_trans(Vt, lt, va);
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = 0.0f;
            mfma_one_chain<NP>(ka, qb, sc);
            if (lt + 1 < lt1) read_rows(Kp, lt + 1, ka);
            if (32 * (c0 + lt) + 32 > n) {                  // the scene's last, partial key tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * (c0 + lt) + acc_row(r, hh) >= n) sc[r] = -INFINITY;
            }
            const float alpha = softmax_tile<NP>(sc, c1, mb_run, l_run, pb);
#pragma unroll
            for (int i = 0; i < 16; ++i) oA[i] *= alpha;
            mfma_one_chain<NP>(va, pb, oA);
        }
    }
    float4 qraw[4];
    if (wv < NKT) load_q(0, qraw);                          // stage B's first query rows: in flight under the merge
    __syncthreads();                                        // K / V^T are dead: their bytes become PV^T and the merge slots
    if (activeA && slA > 0) {
        float *sl = slots + (size_t)wv * 18 * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) sl[r * 64] = oA[r];
        sl[16 * 64] = mb_run;
        sl[17 * 64] = l_run;
    }
    if (NS > 1) __syncthreads();
    if (activeA && slA == 0) {
        // merge the key slices of this query tile in fixed order, normalise, store PV^T in stage B's fragment order
        float m = mb_run;
        for (int w = 1; w < NS; ++w) m = fmaxf(m, slots[((size_t)(wv + w) * 18 + 16) * 64 + lane]);
        const float f0 = __builtin_amdgcn_exp2f(mb_run - m);
        l_run *= f0;
#pragma unroll
        for (int i = 0; i < 16; ++i) oA[i] *= f0;
        for (int w = 1; w < NS; ++w) {
            const float *sw = slots + (size_t)(wv + w) * 18 * 64 + lane;
            const float f = __builtin_amdgcn_exp2f(sw[16 * 64] - m);        // a slice without keys: exp2(-inf) = 0
            l_run = fmaf(sw[17 * 64], f, l_run);
#pragma unroll
            for (int r = 0; r < 16; ++r) oA[r] = fmaf(sw[r * 64], f, oA[r]);
        }
        const float inv = 1.0f / l_run;
        const int pos = 32 * qtA + pv_pos(li);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            unsigned q1, q2, q3;
            split3_pair(oA[r] * inv, oA[r + 1] * inv, q1, q2, q3);
            char *d0 = PVt + acc_row(r, hh) * kFaTStride + pos * 2, *d1 = PVt + acc_row(r + 1, hh) * kFaTStride + pos * 2;
            *reinterpret_cast<unsigned short *>(d0) = (unsigned short)(q1 & 0xffffu);
            *reinterpret_cast<unsigned short *>(d1) = (unsigned short)(q1 >> 16);
            if (NP == 3) {
                *reinterpret_cast<unsigned short *>(d0 + kFaTPlane) = (unsigned short)(q2 & 0xffffu);
                *reinterpret_cast<unsigned short *>(d1 + kFaTPlane) = (unsigned short)(q2 >> 16);
                *reinterpret_cast<unsigned short *>(d0 + 2 * kFaTPlane) = (unsigned short)(q3 & 0xffffu);
                *reinterpret_cast<unsigned short *>(d1 + 2 * kFaTPlane) = (unsigned short)(q3 >> 16);
            }
        }
    }
    __syncthreads();

    // ---- stage B: O = softmax_L(mask((Q scale) P^T)) PV (PRE:241-250); a wave owns whole query tiles (rounds of eight)
    const int rounds = (NKT + kFaWaves - 1) / kFaWaves;
    for (int rnd = 0; rnd < rounds; ++rnd) {
        const int qt = wv + kFaWaves * rnd;
        const bool has = qt < NKT;
        const int tok = 32 * qt + li;
        if (has) {
            const float x0[8] = {qraw[0].x, qraw[0].y, qraw[0].z, qraw[0].w, qraw[1].x, qraw[1].y, qraw[1].z, qraw[1].w};
            const float x1[8] = {qraw[2].x, qraw[2].y, qraw[2].z, qraw[2].w, qraw[3].x, qraw[3].y, qraw[3].z, qraw[3].w};
            frag_parts<NP>(x0, qb[0]);
            frag_parts<NP>(x1, qb[1]);
        }
        if (qt + kFaWaves < NKT) load_q(rnd + 1, qraw);
        float mb = -INFINITY, l = 0.0f;
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.0f;
        auto bias_tile = [&](int j) {                       // padded text tokens / keys beyond the proxies
            float kb[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t4 = *reinterpret_cast<const float4 *>(kbias + 32 * j + 8 * g + 4 * hh);
                kb[4 * g] = t4.x; kb[4 * g + 1] = t4.y; kb[4 * g + 2] = t4.z; kb[4 * g + 3] = t4.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = kb[r] == 0.0f ? sc[r] : kb[r];
        };
        if (has) read_rows(Pk, 0, ka);
        for (int j = 0; has && j < NQ; ++j) {
            read_trans(PVt, j, va);
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = 0.0f;
            mfma_one_chain<NP>(ka, qb, sc);
            if (j + 1 < NQ) read_rows(Pk, j + 1, ka);
            bias_tile(j);
            const float alpha = softmax_tile<NP>(sc, c1, mb, l, pb);
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= alpha;
            mfma_one_chain<NP>(va, pb, o);
        }
        if (has && tok < n) {
            const float inv = 1.0f / l;
            float *dst = p.out + ((size_t)b * n + tok) * C + h * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(dst + 8 * g + 4 * hh) =
                    make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        }
    }
}

bool fused_attn_supported(const FAttnBatch &ab)
{
    static const int off = getenv("PTX_ATTN_FUSED") ? atoi(getenv("PTX_ATTN_FUSED")) == 0 : 0;
    if (off || ab.hd != 32 || ab.C % 4 != 0 || ab.n < 1) return false;
    int lpmax = 0;
    for (int g = 0; g < ab.nb; ++g) lpmax = ab.p[g].Lp > lpmax ? ab.p[g].Lp : lpmax;
    return lpmax >= 1 && lpmax <= kFaChunk;
}

int launch_proxy_attn(const FAttnBatch &ab_in, hipStream_t st)
{
    FAttnBatch ab = ab_in;
    PTX_REQUIRE(ab.nb >= 1 && ab.nb <= 2 && ab.hd == 32, "fused attention: nb=%d hd=%d", ab.nb, ab.hd);
    int lpmax = 0;
    for (int g = 0; g < ab.nb; ++g) {
        PTX_REQUIRE(ab.p[g].qkv && ab.p[g].pt && ab.p[g].out && ab.p[g].Lp >= 1, "fused attention: bad group %d", g);
        lpmax = ab.p[g].Lp > lpmax ? ab.p[g].Lp : lpmax;
    }
    PTX_REQUIRE(lpmax <= 32 * kFaWaves, "fused attention: at most %d proxies (got %d)", 32 * kFaWaves, lpmax);
    const int lds = kFaLds;
    const dim3 grid(ab.B * ab.heads, ab.nb), block(kFaWaves * 64);
    if (ab.compute_dtype == 1) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_proxy_attn<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(k_proxy_attn<1>, grid, block, lds, st, ab);
    } else {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_proxy_attn<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(k_proxy_attn<3>, grid, block, lds, st, ab);
    }
    PTX_LAUNCHED("k_proxy_attn");
    return PTX_OK;
}

}  // namespace ptx
