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
// Calls with few scenes (4 scenes x 8 heads x 2 branches = 64 work-groups on 256 CUs): the proxies of a (scene, head,
// branch) are dealt to up to four work-groups (FAttnBatch::split, at least two proxy tiles each), every one takes its
// slice through both stages -- stage A is complete per proxy, stage B is a partial soft-max over the slice -- and the last
// one to arrive merges the partials (23 -> 17.7 us alone at 4 scenes x 196 proxies, 58 -> 39 us at n = 691).
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

// agent-scope relaxed stores / loads (global_store / global_load ... sc1): visible across the XCDs' L2s without a fence
__device__ __forceinline__ void st_agent(float *p, float a, float b, float c, float d)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_agent(float *p, float a, float b)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
// (loads by inline assembly: the compiler does not count them -- every value is passed through wait_agent() before use)
__device__ __forceinline__ f32x4v ld4_agent(const float *p)
{
    f32x4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x2v ld2_agent(const float *p)
{
    f32x2v v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_agent(f32x4v &a, f32x4v &b, f32x4v &c, f32x4v &d, f32x2v &e, f32x2v &f, f32x2v &g, f32x2v &h)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "memory");
}

constexpr int kFaChunk = 32 * kFaWaves;         // keys staged per pass of stage A: one key tile per wave

// LDS geometry, fixed (so that every fragment address is one per-lane base + an immediate): planes of 256 rows x 64 B for
// the projected proxies and the staged K, rows of 256 elements + 16 B for V^T / PV^T (rows 4 banks apart: 16-byte reads
// of sixteen rows are conflict-free)
constexpr int kFaRowPlane = kFaChunk * XROW;                // 16384
constexpr int kFaTStride = kFaChunk * 2 + 16;               // 528
constexpr int kFaTPlane = 32 * kFaTStride;                  // 16896
constexpr int kFaBiasOff = 3 * kFaRowPlane;                 // key bias of stage B (256 floats)
constexpr int kFaFlagOff = kFaBiasOff + kFaChunk * 4;       // per key tile of stage B: does any key carry a bias (8 ints)
constexpr int kFaRegOff = kFaFlagOff + 32;                  // {K, V^T} during stage A, {PV^T, merge slots} afterwards
constexpr int kFaSlotOff = kFaRegOff + 3 * kFaTPlane;
constexpr int kFaLds = kFaRegOff + 3 * kFaRowPlane + 3 * kFaTPlane;
static_assert(kFaSlotOff + kFaWaves * 18 * 64 * 4 <= kFaLds && kFaLds <= 160 * 1024, "LDS carve of k_proxy_attn");

template <int NP>
__device__ __forceinline__ void mfma_one_chain(const u32x4 (&a)[2][3], const u32x4 (&b)[2][3], f32x16 &c)
{
    c = mfma_parts<NP>(a[0], b[0], c);
    c = mfma_parts<NP>(a[1], b[1], c);
}

// One soft-max step of a 32 x 32 score tile held as sc[r] (this lane's query, 16 of the 32 keys; -inf = no key): running
// maximum mb (already times c1) and sum l, rescale factor of the running output, probabilities split for the MFMA
template <int NP>
__device__ __forceinline__ float softmax_tile(f32x16 &sc, float c1, float &mb, float &l, u32x4 (&pb)[2][3])
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
// Where the time goes (r03, s_memtime stamps of one work-group, 4 scenes x 196 proxies, 57 k cycles = 24 us per launch):
// start-up 7.6 k (requests 1.5-3 k, proxies split 3.1-5 k, K / V split 5.4-7.5 k, barrier), stage A 21 k (eight steps per
// wave, two waves per SIMD: 1330 cycles per step and wave), merge + PV^T 2.5 k, stage B 19.5 k (seven steps, 1390 each).
// A step's matrix time is 24 MFMAs x 32 cycles = 768; with the MFMAs removed the launch takes 20.2 us, with the soft-max
// arithmetic removed 18.8 us, with both in 24.2 us: the two waves of a SIMD hide a little more than half of the matrix time
// under each other's VALU work, the VALU instruction stream is what a step waits for.  What that stream cost, and what
// did NOT help (all measured on one box against the same build, scratch/lab_ab.sh):
//   * fragments carried as bf16x8 values were taken apart and re-packed around every conditional reload (24 v_lshrrev +
//     24 v_perm per step): raw u32x4 registers (split3.h); the masked_fill selects (16 v_cmp + 16 v_cndmask + 14 hazard
//     nops per step) run only for key tiles that have a bias (per-tile flags); the start-up's loads are all requested up
//     front, branch-free, with 32-bit offsets (the proxy staging loop waited with vmcnt(0) per trip; sixteen conditional V
//     loads were sixteen basic blocks with a 64-bit multiply each): 26.1 -> 23.0 us, 31.1 -> 28.6 us at 32 scenes
//   * hand-packed soft-max arithmetic (v_pk_fma / v_pk_mul / v_pk_add: 189 -> 154 VALU instructions per step): 24.0 us;
//     v_pk_add_f32 for the split residuals: 24.9; v_max3_f32 by inline assembly (no canonicalising v_max x, x): 23.7 --
//     fewer instructions, all slower; the compiler's own SLP pairing costs 2 % (-fno-slp-vectorize in the Makefile)
//   * two interleaved accumulator chains per contraction: 27.5 vs 27 us; a barrier-paced ping-pong of matrix and VALU
//     segments between the two wave groups: 34 us; per-query-tile merges behind barriers (first version): 32 us
//   * an in-wave software pipeline of stage B (block j issues the MFMAs of PV(j - 1) and S(j + 1) as two chains in one
//     basic block with the VALU work of soft-max(j), sched_group_barrier places 9 VALU behind every MFMA; 249 VGPRs, no
//     spills, parity green): 25.35 vs 25.39 us -- the extra register traffic costs what the overlap gains.
template <int NP>       // 3: split operands (fp32-equivalent); 1: plain bf16 operands (compute_dtype = 1)
__global__ __launch_bounds__(kFaWaves * 64) void k_proxy_attn(FAttnBatch ab)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FAttnProb p = ab.p[blockIdx.y];
    const int b = blockIdx.x / ab.heads, h = blockIdx.x - b * ab.heads;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int n = ab.n, C = ab.C, NKT = (n + 31) >> 5;
    // proxy split: this work-group takes the proxy tiles [t0, t1) of the branch through both stages as if they were all
    // (a slice of one tile is all fixed cost: branches with one or two proxy tiles -- the text branch -- stay whole)
    const int NQall = (p.Lp + 31) >> 5, S = min(ab.split, (NQall + 1) >> 1), sp = blockIdx.z;
    if (sp >= S) return;
    const int t0 = sp * NQall / S, t1 = (sp + 1) * NQall / S;
    const int Lp = min(p.Lp, 32 * t1) - 32 * t0, NQ = t1 - t0, LpPad = NQ * 32;
    char *Pk = smem;
    float *kbias = reinterpret_cast<float *>(smem + kFaBiasOff);
    int *tflag = reinterpret_cast<int *>(smem + kFaFlagOff);
    char *Kp = smem + kFaRegOff, *Vt = Kp + 3 * kFaRowPlane;        // stage A
    char *PVt = smem + kFaRegOff;                                   // after stage A (same bytes)
    float *slots = reinterpret_cast<float *>(smem + kFaSlotOff);
    const int C3 = 3 * C;
    const float *qkv = p.qkv + (size_t)b * n * C3 + h * 32;
    const float c1 = ab.scale * kLog2e;         // soft-max exponents in base 2: exp(scale (s - m)) = exp2(c1 s - c1 m)
    // per-lane bases of the fragment reads: rows of a row-major plane (two 16-byte pieces, swizzled), rows of a transposed one
    const int rowA = li * XROW + xswz(li, hh), rowB = li * XROW + xswz(li, 2 + hh);     // + tile * 2048 + plane * kFaRowPlane
    const int trow = li * kFaTStride + 16 * hh;                                         // + tile * 64 + 32 s + plane * kFaTPlane
    auto read_rows = [&](const char *base, int tile, u32x4 (&f)[2][3]) {
#pragma unroll
        for (int pt3 = 0; pt3 < NP; ++pt3) {
            f[0][pt3] = *reinterpret_cast<const u32x4 *>(base + rowA + tile * (32 * XROW) + pt3 * kFaRowPlane);
            f[1][pt3] = *reinterpret_cast<const u32x4 *>(base + rowB + tile * (32 * XROW) + pt3 * kFaRowPlane);
        }
    };
    auto read_trans = [&](const char *base, int tile, u32x4 (&f)[2][3]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pt3 = 0; pt3 < NP; ++pt3)
                f[s][pt3] = *reinterpret_cast<const u32x4 *>(base + trow + tile * 64 + 32 * s + pt3 * kFaTPlane);
    };

    // K / V fragments of key tile kt as the MFMAs take them: kf = K[key li][dims 16 s + 8 hh ..], vf = V[key(t, hh, j)][dim li]
    auto load_kv = [&](int kt, float (&kf)[16], float (&vf)[16]) {
        // (offsets are 32-bit and formed by additions from one product per lane: a 64-bit multiply per load is a dozen
        //  quarter-rate instructions, 2 to 3 thousand cycles before the last of these loads was even issued)
        const int key = min(32 * kt + li, n - 1);
        const float4 *kr = reinterpret_cast<const float4 *>(qkv + (key * C3 + C + 8 * hh));
        const float4 k0 = kr[0], k1 = kr[1], k2 = kr[4], k3 = kr[5];
        kf[0] = k0.x; kf[1] = k0.y; kf[2] = k0.z; kf[3] = k0.w; kf[4] = k1.x; kf[5] = k1.y; kf[6] = k1.z; kf[7] = k1.w;
        kf[8] = k2.x; kf[9] = k2.y; kf[10] = k2.z; kf[11] = k2.w; kf[12] = k3.x; kf[13] = k3.y; kf[14] = k3.z; kf[15] = k3.w;
        const int v0 = (32 * kt + 4 * hh) * C3 + 2 * C + li, vmax = (n - 1) * C3 + 2 * C + li;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                vf[8 * t + j] = qkv[min(v0 + ((j & 3) + 8 * (2 * t + (j >> 2))) * C3, vmax)];   // clamped: no branch per load
            }
    };
    // ... split and stored as local tile `lt` of the staged chunk
    auto stash_kv = [&](int lt, int kt, const float (&kf)[16], const float (&vf)[16]) {
        const int kv0 = 32 * kt + 4 * hh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float x[8], y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[i] = kf[8 * s + i];
                y[i] = kv0 + (i & 3) + 8 * (2 * s + (i >> 2)) < n ? vf[8 * s + i] : 0.0f;       // rows beyond the scene: zero
            }
            u32x4 fk[3], fv[3];
            frag_parts<NP>(x, fk);
            frag_parts<NP>(y, fv);
#pragma unroll
            for (int pt3 = 0; pt3 < NP; ++pt3) {
                *reinterpret_cast<u32x4 *>(Kp + (s ? rowB : rowA) + lt * (32 * XROW) + pt3 * kFaRowPlane) = fk[pt3];
                *reinterpret_cast<u32x4 *>(Vt + trow + lt * 64 + 32 * s + pt3 * kFaTPlane) = fv[pt3];
            }
        }
    };
    // the query rows of stage B's round `rnd` of this wave (tokens 32 (wv + 8 rnd) + li), requested long before they are split
    auto load_q = [&](int rnd, float4 (&q)[4]) {
        const int tok = min(32 * (wv + kFaWaves * rnd) + li, n - 1);
        const float4 *qr = reinterpret_cast<const float4 *>(qkv + (tok * C3 + 8 * hh));
        q[0] = qr[0]; q[1] = qr[1]; q[2] = qr[4]; q[3] = qr[5];
    };
    // ---- prologue: every global load of the start-up is requested up front, branch-free (clamped rows, masked when they
    // are split): the projected proxies of this (scene, head), this wave's K / V tile, the key mask
    const float *pt = p.pt + ((size_t)b * p.Lp + 32 * t0) * C + h * 32;
    constexpr int kTrips = kFaChunk * 8 / (kFaWaves * 64);
    float4 pv[kTrips];
#pragma unroll
    for (int i = 0; i < kTrips; ++i) {
        const int e = tid + i * kFaWaves * 64;
        pv[i] = *reinterpret_cast<const float4 *>(pt + (min(e >> 3, Lp - 1) * C + (e & 7) * 4));
    }
    float kf[16], vf[16];
    load_kv(min(wv, NKT - 1), kf, vf);
    int mk = 1;
    if (p.mask != nullptr && tid < kFaChunk) mk = p.mask[(size_t)b * p.Lp + 32 * t0 + min(tid, Lp - 1)];     // (wave-uniform branch)

    // the proxies -> three bf16 planes in LDS
#pragma unroll
    for (int i = 0; i < kTrips; ++i) {
        const int e = tid + i * kFaWaves * 64, row = e >> 3, kq = (e & 7) * 4;
        if (row < LpPad)
            stash_parts<NP>(Pk + row * XROW + xswz(row, kq >> 3) + (kq & 4) * 2, kFaRowPlane,
                            row < Lp ? pv[i] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    // key bias of stage B: 0 = valid key, -inf = beyond the proxies, masked_fill(-1e9) of the SCALED scores (PRE:247) in
    // raw units; and per key tile whether any key has one (most tiles have none: their 16 selects per lane are skipped)
    if (tid < kFaChunk) {                                   // waves 0..3: two key tiles each
        float kb = tid >= Lp ? -INFINITY : 0.0f;
        if (tid < Lp && mk == 0) kb = -1e9f / ab.scale;
        kbias[tid] = kb;
        const unsigned long long nz = __ballot(kb != 0.0f);
        if (lane == 0) {
            tflag[2 * wv] = (unsigned)nz != 0u;
            tflag[2 * wv + 1] = (unsigned)(nz >> 32) != 0u;
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
    u32x4 qb[2][3], ka[2][3], va[2][3], pb[2][3];
    f32x16 sc;
    for (int c0 = 0; c0 < NKT; c0 += kFaWaves) {            // chunks of 8 key tiles staged through LDS
        if (c0 > 0) {
            // (a register prefetch of the next chunk's fragments under the current one costs 32 live registers in a kernel
            //  that has none to spare: larger scenes pay one exposed load per 256 keys instead)
            load_kv(min(c0 + wv, NKT - 1), kf, vf);         // unconditional: the old values are dead for the allocator too
            __syncthreads();                                // every wave is done with the previous chunk
        }
        if (c0 + wv < NKT) stash_kv(wv, c0 + wv, kf, vf);
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
    const unsigned bflags = (unsigned)__ballot(tflag[lane & 7] != 0) & 0xffu;      // bit j: key tile j has a bias somewhere
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
            if ((bflags >> j) & 1u) bias_tile(j);
            const float alpha = softmax_tile<NP>(sc, c1, mb, l, pb);
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= alpha;
            mfma_one_chain<NP>(va, pb, o);
        }
        if (has && tok < n) {
            if (S == 1) {
                const float inv = 1.0f / l;
                float *dst = p.out + ((size_t)b * n + tok) * C + h * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(dst + 8 * g + 4 * hh) =
                        make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            } else {                                        // this slice's share: un-normalised, with its maximum and sum
                float *po = ab.part + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ab.split + sp) * n * kFaPartRow;
#pragma unroll
                for (int g = 0; g < 4; ++g) st_agent(po + (size_t)tok * 32 + 8 * g + 4 * hh, o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
                if (hh == 0) st_agent(po + (size_t)n * 32 + 2 * tok, mb, l);
            }
        }
    }
    if (S == 1) return;
    // ---- the last slice of this (scene, head, branch) to arrive merges them (split soft-max, slices in index order).
    // The partials travel as agent-scope stores and loads (sc1: written through / read past the XCD's L2) around a relaxed
    // ticket -- NOT behind release / acquire fences: a fence is a write-back (buffer_wbl2) plus an invalidate (buffer_inv)
    // of the whole L2 per wave, measured 41 instead of 23 us at 4 scenes and 264 instead of 29 us at 32.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *tk = ab.tickets + blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0) tflag[0] = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tflag[0] != S - 1) return;
    if (tid == 0) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left zero for the next launch
    const float *gp = ab.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ab.split * n * kFaPartRow;
    // (every load of up to four (token, four outputs) items per thread is requested before the first is used: they come from
    //  the memory side, one round trip instead of one per item -- 13-19 k cycles of the first version's 46 k)
    constexpr int kItems = 4;
    for (int i0 = tid; i0 < n * 8; i0 += kItems * kFaWaves * 64) {
        f32x4v v[kItems][kFaMaxSplit];
        f32x2v ml[kItems][kFaMaxSplit];
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int idx = min(i0 + it * kFaWaves * 64, n * 8 - 1), tok = idx >> 3, d4 = (idx & 7) * 4;
#pragma unroll
            for (int s = 0; s < kFaMaxSplit; ++s) {
                const float *ps = gp + (size_t)min(s, S - 1) * n * kFaPartRow;     // (slices that do not exist: re-read the last)
                v[it][s] = ld4_agent(ps + (size_t)tok * 32 + d4);
                ml[it][s] = ld2_agent(ps + (size_t)n * 32 + 2 * tok);
            }
        }
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            wait_agent(v[it][0], v[it][1], v[it][2], v[it][3], ml[it][0], ml[it][1], ml[it][2], ml[it][3]);
            const int idx = i0 + it * kFaWaves * 64, tok = idx >> 3, d4 = (idx & 7) * 4;
            float m = -INFINITY;
#pragma unroll
            for (int s = 0; s < kFaMaxSplit; ++s) m = fmaxf(m, s < S ? ml[it][s][0] : -INFINITY);
            float lsum = 0.0f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < kFaMaxSplit; ++s) {
                const float w = s < S ? __builtin_amdgcn_exp2f(ml[it][s][0] - m) : 0.0f;
                lsum = fmaf(ml[it][s][1], w, lsum);
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[d] = fmaf(v[it][s][d], w, acc[d]);
            }
            const float inv = 1.0f / lsum;
            if (idx < n * 8)
                *reinterpret_cast<float4 *>(p.out + ((size_t)b * n + tok) * C + h * 32 + d4) =
                    make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
        }
    }
}

bool fused_attn_supported(const FAttnBatch &ab)
{
    if (ab.hd != 32 || ab.C % 4 != 0 || ab.n < 1) return false;
    int lpmax = 0;
    for (int g = 0; g < ab.nb; ++g) lpmax = ab.p[g].Lp > lpmax ? ab.p[g].Lp : lpmax;
    return lpmax >= 1 && lpmax <= kFaChunk;
}

// Slices of the proxies per (scene, head, branch): with few pairs in a call one work-group each leaves most of the chip idle
// (4 scenes x 8 heads x 2 branches = 64 of 256 CUs); up to four slices fill it.
int fattn_split_for(int B, int heads)
{
    const long groups = (long)B * heads * 2;
    const long s = 256 / (groups > 0 ? groups : 1);
    return s < 1 ? 1 : s > kFaMaxSplit ? kFaMaxSplit : (int)s;
}

int launch_proxy_attn(const FAttnBatch &ab_in, hipStream_t st)
{
    FAttnBatch ab = ab_in;
    if (ab.split == 0) ab.split = 1;                        // callers that do not know about the proxy split
    PTX_REQUIRE(ab.nb >= 1 && ab.nb <= 2 && ab.hd == 32, "fused attention: nb=%d hd=%d", ab.nb, ab.hd);
    int lpmax = 0;
    for (int g = 0; g < ab.nb; ++g) {
        PTX_REQUIRE(ab.p[g].qkv && ab.p[g].pt && ab.p[g].out && ab.p[g].Lp >= 1, "fused attention: bad group %d", g);
        lpmax = ab.p[g].Lp > lpmax ? ab.p[g].Lp : lpmax;
    }
    PTX_REQUIRE(lpmax <= 32 * kFaWaves, "fused attention: at most %d proxies (got %d)", 32 * kFaWaves, lpmax);
    const int lds = kFaLds;
    PTX_REQUIRE(ab.split >= 1 && ab.split <= kFaMaxSplit && (ab.split == 1 || (ab.part && ab.tickets)),
                "fused attention: split=%d needs the partial buffers", ab.split);
    // (work-groups are dealt to the XCDs round-robin by linear id, x fastest: with B * heads a multiple of eight every slice and
    //  both branches of a (scene, head) run on XCD h % 8 and share its L2 for K, V and the partials)
    const dim3 grid(ab.B * ab.heads, ab.nb, ab.split), block(kFaWaves * 64);
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
