// Clustering and apply kernels of the preshape path (gfx950, wave64).
//
//   k_minmax        PRE:37-38   per-scene bbox (HBM-bound streaming read, 12 B/point)
//   k_ball_query    PRE:56,65   pytorch3d.ops.ball_query: first K in index order, r = 3
//                               (+ PRE:41-48 grid centres when GRID)
//   k_slot_net      PRE:87-107 / 126-142 OffsetNetwork / SimplifiedPointNet, one wave per cluster
//   k_select        PRE:352-420 padding-count ordering, FPS, keep list, gathers, ownership tags
//   k_tile_count / k_affine     PRE:459-467, 472-525 affine + last-writer scatter + ordered drop
//
// Build with -ffp-contract=off: squared distances decide integer results and must be
// ((dx*dx)+dy*dy)+dz*dz exactly; they additionally use __f*_rn intrinsics.
#include <cstdlib>
#include <hip/hip_ext.h>

#include "common.h"

namespace ptx {

// ------------------------------------------------------------------------------ min / max
// mm_enc[b][0..2] = ~ord(min_d)   (so that atomicMax over a zeroed word yields the minimum)
// mm_enc[b][3..5] =  ord(max_d)
__global__ __launch_bounds__(256) void k_minmax(ScenePts points, int N,
                                                uint32_t *__restrict__ mm_enc)
{
    __shared__ float red[4][6];
    minmax_block<false>(points.p[blockIdx.y], N, mm_enc, blockIdx.y, blockIdx.x, gridDim.x, red);      // common.h
}

int launch_minmax(const ScenePts &points, int B, int N, uint32_t *mm_enc, hipStream_t st)
{
    hipLaunchKernelGGL(k_minmax, dim3(minmax_chunks(N), B), dim3(256), 0, st, points, N, mm_enc);
    PTX_LAUNCHED("k_minmax");
    return PTX_OK;
}

// ------------------------------------------------------------------------------ ball query
// One wave per centre; lanes test 64 consecutive points at a time, a ballot + prefix popcount
// keeps hits in index order, the wave stops as soon as it has K hits.  All waves of a launch
// walk the same short prefix of the point array, which stays in L2.
constexpr int kBqUnroll = 4;   // 256 points per outer step

// PRE:48: (min + margin) + lin * ((max - min) - 2*margin), python operator order; meshgrid 'ij' (PRE:44)
__device__ __forceinline__ void grid_centre(const uint32_t *__restrict__ mm_enc, const float *__restrict__ lin, int gs,
                                            float margin, int b, int m, float (&mn)[3], float (&mx)[3], float (&c)[3])
{
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn[d] = ord2f(~mm_enc[b * 6 + d]); mx[d] = ord2f(mm_enc[b * 6 + 3 + d]); }
    const int ijk[3] = {m / (gs * gs), (m / gs) % gs, m % gs};
    const float two_margin = __fmul_rn(2.0f, margin);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float base = __fadd_rn(mn[d], margin);
        const float span = __fsub_rn(__fsub_rn(mx[d], mn[d]), two_margin);
        c[d] = __fadd_rn(base, __fmul_rn(lin[ijk[d]], span));
    }
}

// One wave scans the scene for the first K points with dist2 < r2 around (cx,cy,cz), in index order.  Hit number
// `pos` goes to oi[pos] (point index, may be null) and oc[3 pos ..] (xyz; global or wave-private LDS); slots
// beyond the hit count are padded with -1 / 0.0 (masked_gather, PRE:664-671).  Returns the hit count (<= K).
//
// r06 (VERDICT r05 "next" #5, north_star's "LDS-staged cluster tiles"): a centre that fills its K slots within a few hundred points
// -- every benchmark distribution: P_max 2 677 at cfg2, < 1 000 in rooms -- is served best by the wave walking the scene on its
// own (the prefix stays in L2, nothing is synchronised).  A centre that NEVER fills (sparse or clumped clouds, large extents)
// reads all N points, and a wave on its own does that as ~N / 256 dependent round trips: 1.13 ms for four cfg2 scenes of a
// two-blob cloud against 31 us at the benchmark distribution (profiles/r06_cluster_regimes.txt).  So: the first kBqPrivate points
// wave-private as before; if any wave of the work-group is still short of K hits then, the WORK-GROUP stages the rest of the
// scene through LDS in tiles of kBqTile points -- all 256 threads request the next tile with 16-byte loads while the unfinished
// waves test the current one out of LDS (stride-3 dword reads: conflict-free), one barrier + vote per tile, until every wave has
// its K hits or the scene ends.  Same ballot + prefix-popcount bookkeeping on the same points in the same order: bit-identical.
constexpr int kBqPrivate = 4096;          // points every wave scans on its own first
constexpr int kBqTile = 512;              // points per staged tile (6 KB; two buffers per work-group: with 1024-point tiles the 27 KB of
                                          // LDS capped k_cluster at 5 work-groups per CU and cost the benchmark shape 7 %)
constexpr int kBqTileFloats = kBqTile * 3;

struct BqHits { int32_t *oi; float *oc; int K; int count; unsigned long long lt; };

__device__ __forceinline__ void bq_group(BqHits &h, int j, bool valid, float px, float py, float pz, float cx, float cy, float cz,
                                         float r2)
{
    const float d2 = dist2_nofma(cx, cy, cz, px, py, pz);
    const bool hit = valid && (d2 < r2);                        // strict <
    const unsigned long long mask = __ballot(hit);
    const int pos = h.count + __popcll(mask & h.lt);
    if (hit && pos < h.K) {
        if (h.oi) h.oi[pos] = j;
        h.oc[pos * 3] = px; h.oc[pos * 3 + 1] = py; h.oc[pos * 3 + 2] = pz;
    }
    h.count += __popcll(mask);
}

// `tiles`: 2 * kBqTileFloats floats of LDS shared by the work-group, or null = wave-private scan only.  With tiles != null EVERY
// thread of the (256-thread) work-group must make this call with the same p and N (barriers inside).
__device__ __forceinline__ int bq_scan(const float *__restrict__ p, int N, int K, float cx, float cy, float cz, float r2,
                                       int32_t *oi, float *oc, float *tiles = nullptr)
{
    const int lane = lane_id();
    BqHits h{oi, oc, K, 0, (1ull << lane) - 1ull};
    // the staged part starts where p + 3 * start is 16-byte aligned (p is a (N,3) fp32 tensor or a row of a (B,N,3) one)
    int n1 = N;
    if (tiles != nullptr) {
        // p + 3 (kBqPrivate + d) floats is 16-byte aligned for d = (p's offset in floats) mod 4:  fo + 3 d = 0 (mod 4)  <=>  d = fo
        const int start = kBqPrivate + (int)((reinterpret_cast<uintptr_t>(p) >> 2) & 3u);
        n1 = min(N, start);
    }
    for (int base = 0; base < n1 && h.count < K; base += 64 * kBqUnroll) {
        float px[kBqUnroll], py[kBqUnroll], pz[kBqUnroll];
#pragma unroll
        for (int u = 0; u < kBqUnroll; ++u) {
            const int j = base + u * 64 + lane;
            if (j < n1) { px[u] = p[(size_t)j * 3]; py[u] = p[(size_t)j * 3 + 1]; pz[u] = p[(size_t)j * 3 + 2]; }
            else        { px[u] = py[u] = pz[u] = 0.0f; }
        }
#pragma unroll
        for (int u = 0; u < kBqUnroll; ++u) {
            if (h.count < K) {                                 // wave-uniform
                const int j = base + u * 64 + lane;
                bq_group(h, j, j < n1, px[u], py[u], pz[u], cx, cy, cz, r2);
            }
        }
    }
    if (tiles != nullptr && n1 < N) {                           // work-group uniform (same p, N in every thread)
        const int tid = threadIdx.x;
        const float *src = p + (size_t)n1 * 3;                  // 16-byte aligned
        const int nfl = (N - n1) * 3;                           // floats left
        constexpr int kF4 = kBqTileFloats / 4, kPer = (kF4 + 255) / 256;      // 384 float4 per tile: two per thread, the second for half of them
        float4 nx[kPer];
        auto fetch = [&](int t) {
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                if (tid + 256 * i >= kF4) continue;
                const int f = t * kBqTileFloats + 4 * (tid + 256 * i);
                if (f + 3 < nfl) nx[i] = *reinterpret_cast<const float4 *>(src + f);
                else {                                          // the ragged end: floats beyond the scene are 3e38 -- such a "point" is at an
                    constexpr float kFar = 3.0e38f;             // infinite squared distance from every centre (inf < r2 is false)
                    nx[i].x = f < nfl ? src[f] : kFar; nx[i].y = f + 1 < nfl ? src[f + 1] : kFar;
                    nx[i].z = f + 2 < nfl ? src[f + 2] : kFar; nx[i].w = kFar;
                }
            }
        };
        auto stash = [&](int t) {
#pragma unroll
            for (int i = 0; i < kPer; ++i)
                if (tid + 256 * i < kF4) *reinterpret_cast<float4 *>(tiles + (t & 1) * kBqTileFloats + 4 * (tid + 256 * i)) = nx[i];
        };
        const int ntile = (N - n1 + kBqTile - 1) / kBqTile;
        bool any = __syncthreads_or(h.count < K);              // (also: nobody still reads `tiles` from an earlier call)
        if (any) {
            fetch(0);
            stash(0);
            __syncthreads();
            for (int t = 0; t < ntile; ++t) {
                if (t + 1 < ntile) fetch(t + 1);
                if (h.count < K) {                              // wave-uniform
                    // four groups of 64 points at a time: all twelve LDS reads first, four ballots, and the hit bookkeeping only
                    // when one of them is non-empty -- in the regime this path exists for (centres that never fill) nearly every
                    // group is empty, and the loop is 3 reads + 8 VALU + a compare per 64 points (the generic loop above it: ~40
                    // instructions, r06 ISA count; the pass is bound by exactly this instruction stream: M N distance tests)
                    const float *tl = tiles + (t & 1) * kBqTileFloats;
                    const int tbase = n1 + t * kBqTile;
#pragma unroll 1
                    for (int g0 = 0; g0 < kBqTile / 64; g0 += 4) {
                        float x[4], y[4], z[4];
                        unsigned long long mk[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int q = (g0 + u) * 64 + lane;
                            x[u] = tl[q * 3]; y[u] = tl[q * 3 + 1]; z[u] = tl[q * 3 + 2];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) mk[u] = __ballot(dist2_nofma(cx, cy, cz, x[u], y[u], z[u]) < r2);
                        if ((mk[0] | mk[1] | mk[2] | mk[3]) != 0ull) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (mk[u] != 0ull && h.count < K) {
                                    const int pos = h.count + __popcll(mk[u] & h.lt);
                                    if (((mk[u] >> lane) & 1ull) && pos < K) {
                                        if (h.oi) h.oi[pos] = tbase + (g0 + u) * 64 + lane;
                                        h.oc[pos * 3] = x[u]; h.oc[pos * 3 + 1] = y[u]; h.oc[pos * 3 + 2] = z[u];
                                    }
                                    h.count += __popcll(mk[u]);
                                }
                            }
                            if (h.count >= K) break;
                        }
                    }
                }
                if (t + 1 < ntile) stash(t + 1);
                any = __syncthreads_or(h.count < K);           // tile t + 1 visible, tile t free; anybody still short?
                if (!any) break;
            }
        }
    }
    int count = h.count;
    if (count > K) count = K;
    for (int k = count + lane; k < K; k += 64) {               // masked_gather padding (PRE:664-671)
        if (oi) oi[k] = -1;
        oc[k * 3] = 0.0f; oc[k * 3 + 1] = 0.0f; oc[k * 3 + 2] = 0.0f;
    }
    return count;
}

template <bool GRID>
__global__ __launch_bounds__(256) void k_ball_query(
    const float *__restrict__ centers, const uint32_t *__restrict__ mm_enc,
    const float *__restrict__ lin, int gs, float margin, float *__restrict__ minmax_out,
    float *__restrict__ centers_out, ScenePts points, int BM, int M, int N, int K,
    float radius, int32_t *__restrict__ idx, float *__restrict__ cluster,
    int32_t *__restrict__ pad_count)
{
    __shared__ __attribute__((aligned(16))) float s_tiles[2 * kBqTileFloats];
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    // the work-group stages tiles only when its four waves scan the SAME scene (always, unless M is no multiple of 4: then the
    // work-groups that straddle two scenes, and a ragged last one, keep the wave-private scan)
    const bool coop = (int)blockIdx.x * 4 + 3 < BM && ((int)blockIdx.x * 4) / M == ((int)blockIdx.x * 4 + 3) / M;
    if (w >= BM) return;
    const int b = w / M, m = w - b * M;
    float cx, cy, cz;
    if (GRID) {
        float mn[3], mx[3], c[3];
        grid_centre(mm_enc, lin, gs, margin, b, m, mn, mx, c);
        cx = c[0]; cy = c[1]; cz = c[2];
        if (lane < 3) centers_out[(size_t)w * 3 + lane] = c[lane];
        if (m == 0 && lane < 3) { minmax_out[b * 6 + lane] = mn[lane]; minmax_out[b * 6 + 3 + lane] = mx[lane]; }
    } else {
        cx = centers[(size_t)w * 3 + 0]; cy = centers[(size_t)w * 3 + 1]; cz = centers[(size_t)w * 3 + 2];
    }
    const float r2 = __fmul_rn(radius, radius);
    const int count = bq_scan(points.p[b], N, K, cx, cy, cz, r2, idx + (size_t)w * K, cluster + (size_t)w * K * 3,
                              coop ? s_tiles : nullptr);
    if (pad_count != nullptr && lane == 0) pad_count[w] = K - count;
}

int launch_ball_query(const float *centers, const uint32_t *mm_enc, const float *lin, int gs,
                      float margin, float *minmax_out, float *centers_out, const ScenePts &points,
                      int B, int M, int N, int K, float radius, int32_t *idx, float *cluster,
                      int32_t *pad_count, hipStream_t st)
{
    const int BM = B * M;
    const dim3 grid(cdiv(BM, 4)), block(256);
    if (centers == nullptr) {
        hipLaunchKernelGGL(k_ball_query<true>, grid, block, 0, st, nullptr, mm_enc, lin, gs, margin,
                           minmax_out, centers_out, points, BM, M, N, K, radius, idx, cluster, pad_count);
    } else {
        hipLaunchKernelGGL(k_ball_query<false>, grid, block, 0, st, centers, nullptr, nullptr, 0, 0.0f,
                           nullptr, nullptr, points, BM, M, N, K, radius, idx, cluster, pad_count);
    }
    PTX_LAUNCHED("k_ball_query");
    return PTX_OK;
}

// ------------------------------------------------------------------------------ slot networks
// Shared front of OffsetNetwork (PRE:87-107) and SimplifiedPointNet (PRE:126-142):
//   x_k = [rel_k (zeroed on padded slots), p_k]  (6)      PRE:93-99
//   h_k = relu(alpha * (W x_k + b) + beta)       (256)    Conv2d 1x1 + eval BatchNorm2d + ReLU
// One wave per cluster, lane owns channels lane, lane+64, lane+128, lane+192; the K slot inputs
// are produced by lanes 0..K-1 and broadcast through v_readlane (SGPR operands of the FMAs).
// MODE 0: mean_K -> 256->3 map -> tanh*margin -> add -> clamp  (PRE:59-62, 102-103)
// MODE 1: max_K  -> point_proxy (+ LayerNorm1 + per-slot bias for both ProxyBlocks, PRE:274, 212-217)
struct SlotNetArgs {
    const float *ab;                  // (2,256) alpha, beta
    const float *conv_w, *conv_b;     // (256,6), (256)
    const float *center, *cluster;    // (BM,3), (BM,K,3)
    int BM, Mper, K;
    // MODE 0
    const float *map_w; const float *minmax; float margin; float *centers_out; float *offsets_out;
    // MODE 1
    float *proxy; const float *n1w[2]; const float *n1b[2]; const float *posb[2]; float *xin[2];
    float ln_eps;
    const int32_t *ksrc; int Msrc;                  // MODE 1: cluster rows gathered through the selection when ksrc != null
    int center_src;                                 // MODE 1 with ksrc: the centre is an un-gathered row as well (early proxies)
    uint32_t *head_flag; uint32_t head_seq;         // the first thread of the launch stores head_seq (api.hip, "gates": the stream being
                                                    // in order, k_select in front of this launch has completed by then), or null
};

// pooled hidden features of one cluster: lane k (< K) holds slot k's input x[6]; returns acc[q] = channel
// lane + 64 q of  sum_k / max_k relu(alpha (W x_k + b) + beta)
template <bool MAXPOOL, int Q>     // Q = hidden width / 64 channels per lane (4: the reference's 256; 8: 512)
__device__ __forceinline__ void slot_pool(const float *__restrict__ conv_w, const float *__restrict__ conv_b,
                                          const float *__restrict__ ab, int K, const float (&x)[6], float (&acc)[Q])
{
    // Two channels per instruction (v_pk_fma_f32 / v_pk_max_f32 / v_pk_add_f32: per element the same IEEE operations in
    // the same order as the scalar form, so every result is bit-identical): this loop is 2/3 of k_cluster's instructions and
    // the launch is VALU-issue-bound
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int W = 64 * Q, P = Q / 2;
    static_assert(Q % 2 == 0, "channels per lane come in pairs");
    const int lane = lane_id();
    f32x2 wt[P][6], bs[P], al[P], be[P], ac[P];
#pragma unroll
    for (int pq = 0; pq < P; ++pq)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = lane + 64 * (2 * pq + e);
#pragma unroll
            for (int i = 0; i < 6; ++i) wt[pq][i][e] = conv_w[c * 6 + i];
            bs[pq][e] = conv_b[c]; al[pq][e] = ab[c]; be[pq][e] = ab[W + c];
            ac[pq][e] = MAXPOOL ? -INFINITY : 0.0f;
        }
    const f32x2 zero = {0.0f, 0.0f};
    for (int k = 0; k < K; ++k) {
        f32x2 s[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[i]), k));
            s[i] = f32x2{v, v};
        }
#pragma unroll
        for (int pq = 0; pq < P; ++pq) {
            f32x2 h = bs[pq];
#pragma unroll
            for (int i = 0; i < 6; ++i) h = __builtin_elementwise_fma(wt[pq][i], s[i], h);
            h = __builtin_elementwise_max(__builtin_elementwise_fma(h, al[pq], be[pq]), zero);
            ac[pq] = MAXPOOL ? __builtin_elementwise_max(ac[pq], h) : ac[pq] + h;
        }
    }
#pragma unroll
    for (int pq = 0; pq < P; ++pq) { acc[2 * pq] = ac[pq][0]; acc[2 * pq + 1] = ac[pq][1]; }
}

// slot k's input from its xyz and the centre: [rel (zeroed on padded slots), p]   PRE:93-99 / 131-137
__device__ __forceinline__ void slot_input(float px, float py, float pz, float cx, float cy, float cz, float (&x)[6])
{
    const bool pad = (px == 0.0f) && (py == 0.0f) && (pz == 0.0f);     // PRE:94 / PRE:132
    x[0] = pad ? 0.0f : px - cx; x[1] = pad ? 0.0f : py - cy; x[2] = pad ? 0.0f : pz - cz;
    x[3] = px; x[4] = py; x[5] = pz;
}

// OffsetNetwork tail: mean_K -> 256->3 map -> tanh*margin -> add -> clamp (PRE:59-62, 102-103).  Lanes 0..2 return
// component `lane` of the new centre in `nc` and of the offset in `off`.
template <int Q>
__device__ __forceinline__ void offset_tail(const float (&acc)[Q], const float *__restrict__ map_w, int K, float margin,
                                            float cen, float mn, float mx, float &nc, float &off)
{
    constexpr int W = 64 * Q;
    const int lane = lane_id();
    const float invK = 1.0f / (float)K;
    float o[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const float hm = acc[q] * invK;
        const int c = lane + 64 * q;
#pragma unroll
        for (int j = 0; j < 3; ++j) o[j] = fmaf(map_w[j * W + c], hm, o[j]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = wave_sum(o[j]);
    const float raw = lane == 0 ? o[0] : (lane == 1 ? o[1] : o[2]);
    off = tanhf(raw) * margin;                                 // PRE:59
    nc = fmaxf(fminf(cen + off, mx), mn);                      // PRE:61-62
}

template <int MODE, int Q>
__global__ __launch_bounds__(256) void k_slot_net(SlotNetArgs a)
{
    constexpr int W = 64 * Q;
    const int lane = lane_id();
    if (a.head_flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.head_flag, a.head_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= a.BM) return;
    // MODE 1 may read its cluster through the selection (row ksrc[w] of the un-gathered array, written by k_select next
    // to the kept centres) instead of a gathered copy: the gathers are then off the chain that leads to the proxy blocks
    int src = w;
    if (MODE == 1 && a.ksrc != nullptr) src = (w / a.Mper) * a.Msrc + a.ksrc[w];
    // MODE 1: everything the epilogue needs from memory is requested NOW (LayerNorm weights, the slot-bias rows): on a
    // chip that holds one wave per SIMD for this launch nothing else hides those round trips
    float e_w[2][Q], e_b[2][Q], e_p[2][Q];
    if (MODE == 1) {
        const int j = w % a.Mper;                                                  // kept-slot position (Q9)
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            if (a.xin[br] == nullptr) continue;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int c = lane + 64 * q;
                e_w[br][q] = a.n1w[br][c]; e_b[br][q] = a.n1b[br][c]; e_p[br][q] = a.posb[br] ? a.posb[br][(size_t)j * W + c] : 0.0f;
            }
        }
    }
    const size_t cw = (MODE == 1 && a.center_src) ? (size_t)src : (size_t)w;
    const float cx = a.center[cw * 3], cy = a.center[cw * 3 + 1], cz = a.center[cw * 3 + 2];
    // lane k (< K) prepares slot k
    float x[6] = {0, 0, 0, 0, 0, 0};
    if (lane < a.K) {
        const float *pk = a.cluster + ((size_t)src * a.K + lane) * 3;
        slot_input(pk[0], pk[1], pk[2], cx, cy, cz, x);
    }
    float acc[Q];
    slot_pool<MODE == 1, Q>(a.conv_w, a.conv_b, a.ab, a.K, x, acc);
    if (MODE == 0) {
        const int b = w / a.Mper;
        const int l3 = lane < 3 ? lane : 0;
        const float cen = lane == 0 ? cx : (lane == 1 ? cy : cz);
        float nc, off;
        offset_tail<Q>(acc, a.map_w, a.K, a.margin, cen, a.minmax[b * 6 + l3], a.minmax[b * 6 + 3 + l3], nc, off);
        if (lane < 3) {
            a.centers_out[(size_t)w * 3 + lane] = nc;
            if (a.offsets_out) a.offsets_out[(size_t)w * 3 + lane] = off;
        }
    } else {
        float *pr = a.proxy + (size_t)w * W;
#pragma unroll
        for (int q = 0; q < Q; ++q) pr[lane + 64 * q] = acc[q];
        // LayerNorm1 (PRE:274) + per-slot bias (PRE:215-217) for the text / image ProxyBlock
        float sum = 0.0f;
#pragma unroll
        for (int q = 0; q < Q; ++q) sum += acc[q];
        float mean = wave_sum(sum) * (1.0f / W);
        float var = 0.0f;
#pragma unroll
        for (int q = 0; q < Q; ++q) { const float d = acc[q] - mean; var = fmaf(d, d, var); }
        var = wave_sum(var) * (1.0f / W);
        const float rstd = 1.0f / sqrtf(var + a.ln_eps);
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            if (a.xin[br] == nullptr) continue;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int c = lane + 64 * q;
                const float xn = (acc[q] - mean) * rstd * e_w[br][q] + e_b[br][q];
                a.xin[br][(size_t)w * W + c] = xn + e_p[br][q];
            }
        }
    }
}

// ------------------------------------------------------------------------------ fused clustering
// get_point_cluster (PRE:53-67) per centre in ONE launch: grid centre -> ball query #1 (gathered xyz only, Q3) ->
// OffsetNetwork + tanh*margin + clamp -> ball query #2.  Every step of a centre depends only on that centre, so
// one wave carries it from the bounding box to its final cluster; the K slots of query #1 never leave the CU
// (wave-private LDS), and three launches with their cold starts become one.  Arithmetic is exactly that of the
// separate kernels (same device functions).
struct ClusterArgs {
    const uint32_t *mm_enc; const float *lin; int gs; float margin, radius;
    ScenePts points; int BM, M, N, K;
    const float *ab, *conv_w, *conv_b, *map_w;         // OffsetNetwork
    const float *centers_override;                     // test hook or null
    float *minmax_out, *centers0, *cluster1, *offsets; // optional outputs (debug / stage parity)
    float *centers; int32_t *idx2; float *cluster2; int32_t *pad_count;
};

__global__ __launch_bounds__(256) void k_cluster(ClusterArgs a)
{
    __shared__ float s_slots[4][64 * 3];
    __shared__ __attribute__((aligned(16))) float s_tiles[2 * kBqTileFloats];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    // LDS-staged tiles for centres that do not fill within kBqPrivate points: only when the four waves scan the same scene (bq_scan)
    const bool coop = (int)blockIdx.x * 4 + 3 < a.BM && ((int)blockIdx.x * 4) / a.M == ((int)blockIdx.x * 4 + 3) / a.M;
    float *tiles = coop ? s_tiles : nullptr;
    if (w >= a.BM) return;
    const int b = w / a.M, m = w - b * a.M;
    float mn[3], mx[3], c[3];
    grid_centre(a.mm_enc, a.lin, a.gs, a.margin, b, m, mn, mx, c);
    if (a.centers0 && lane < 3) a.centers0[(size_t)w * 3 + lane] = c[lane];
    if (a.minmax_out && m == 0 && lane < 3) { a.minmax_out[b * 6 + lane] = mn[lane]; a.minmax_out[b * 6 + 3 + lane] = mx[lane]; }
    const float r2 = __fmul_rn(a.radius, a.radius);
    const float *__restrict__ p = a.points.p[b];
    float *slots = s_slots[wv];
    bq_scan(p, a.N, a.K, c[0], c[1], c[2], r2, nullptr, slots, tiles);          // PRE:56
    float x[6] = {0, 0, 0, 0, 0, 0};
    if (lane < a.K) {
        const float px = slots[lane * 3], py = slots[lane * 3 + 1], pz = slots[lane * 3 + 2];
        slot_input(px, py, pz, c[0], c[1], c[2], x);
        if (a.cluster1) {
            float *o = a.cluster1 + ((size_t)w * a.K + lane) * 3;
            o[0] = px; o[1] = py; o[2] = pz;
        }
    }
    constexpr int Q = kSlotHidden / 64;
    float acc[Q];
    slot_pool<false, Q>(a.conv_w, a.conv_b, a.ab, a.K, x, acc);
    const float cen = lane == 0 ? c[0] : (lane == 1 ? c[1] : c[2]);
    const float lo = lane == 0 ? mn[0] : (lane == 1 ? mn[1] : mn[2]);
    const float hi = lane == 0 ? mx[0] : (lane == 1 ? mx[1] : mx[2]);
    float nc, off;
    offset_tail<Q>(acc, a.map_w, a.K, a.margin, cen, lo, hi, nc, off);
    if (a.offsets && lane < 3) a.offsets[(size_t)w * 3 + lane] = off;
    if (a.centers_override && lane < 3) nc = a.centers_override[(size_t)w * 3 + lane];
    if (lane < 3) a.centers[(size_t)w * 3 + lane] = nc;
    const float cx = PTX_LANE_F(nc, 0), cy = PTX_LANE_F(nc, 1), cz = PTX_LANE_F(nc, 2);
    const int count = bq_scan(p, a.N, a.K, cx, cy, cz, r2, a.idx2 + (size_t)w * a.K, a.cluster2 + (size_t)w * a.K * 3, tiles);   // PRE:65
    if (lane == 0) a.pad_count[w] = a.K - count;
}

// (r03, measured and removed: two neighbouring centres per wave sharing the wave's point loads -- half the load instructions
//  and dependent round trips per centre, bit-identical results -- is SLOWER at every shape: 128 -> 145 us at 32 scenes,
//  79 -> 98 us for the reference's gs = 12 batch of six, 42 -> 56 us at 4 scenes.  The launch is bound by the VALU work of
//  the offset network (1 950 instructions per wave, SQ_ACTIVE_INST_VALU 43 % of the kernel at 2.6 resident waves per SIMD
//  beside the pooling pass), which two centres in one wave only serialise.)
int launch_cluster(const PtxShape &s, const uint32_t *mm_enc, const float *lin, const ScenePts &points,
                   const float *off_ab, const PtxSlotMlp &mlp, const float *map_w, const float *centers_override,
                   float *minmax_out, float *centers0, float *cluster1, float *offsets, float *centers,
                   int32_t *idx2, float *cluster2, int32_t *pad_count, hipStream_t st, hipEvent_t done)
{
    const int M = s.grid_size * s.grid_size * s.grid_size;
    ClusterArgs a{mm_enc, lin, s.grid_size, s.margin, s.radius, points, s.B * M, M, s.N, s.K,
                  off_ab, mlp.conv_w, mlp.conv_b, map_w, centers_override,
                  minmax_out, centers0, cluster1, offsets, centers, idx2, cluster2, pad_count};
    // `done` (the streams that fork off behind the clusters wait for it) rides on the kernel's own completion signal, like
    // k_select's: an event record is a packet of its own between this kernel and the next one of the stream
    if (done != nullptr) hipExtLaunchKernelGGL(k_cluster, dim3(cdiv(s.B * M, 4)), dim3(256), 0, st, nullptr, done, 0, a);
    else hipLaunchKernelGGL(k_cluster, dim3(cdiv(s.B * M, 4)), dim3(256), 0, st, a);
    PTX_LAUNCHED("k_cluster");
    return PTX_OK;
}

int launch_offset_net(const float *ab, const PtxSlotMlp &mlp, const float *map_w,
                      const float *centers_in, const float *cluster, const float *minmax,
                      int BM, int M, int K, float margin, float *centers_out, float *offsets_out,
                      hipStream_t st)
{
    SlotNetArgs a{};
    a.ab = ab; a.conv_w = mlp.conv_w; a.conv_b = mlp.conv_b; a.center = centers_in; a.cluster = cluster;
    a.BM = BM; a.Mper = M; a.K = K; a.map_w = map_w; a.minmax = minmax; a.margin = margin;
    a.centers_out = centers_out; a.offsets_out = offsets_out;
    hipLaunchKernelGGL((k_slot_net<0, kSlotHidden / 64>), dim3(cdiv(BM, 4)), dim3(256), 0, st, a);
    PTX_LAUNCHED("k_slot_net<offset>");
    return PTX_OK;
}

int launch_pointnet(const float *ab, const PtxSlotMlp &mlp, const float *kcenter,
                    const float *kcluster, int BM, int Mk, int K, int width, float *point_proxy,
                    const PtxBlock *blk_t, const PtxBlock *blk_i, const float *posb_t,
                    const float *posb_i, float *xin_t, float *xin_i, float ln_eps,
                    const int32_t *ksrc, int Msrc, hipStream_t st, uint32_t *head_flag, uint32_t head_seq, bool center_src)
{
    // ksrc given: kcluster is the UN-gathered (B,Msrc,K,3) array and kept cluster j of scene b reads row ksrc[b][j]
    // (kcenter is the gathered (B,Mk,3) array unless center_src says it is un-gathered too)
    SlotNetArgs a{};
    a.center_src = center_src && ksrc != nullptr ? 1 : 0;
    a.ab = ab; a.conv_w = mlp.conv_w; a.conv_b = mlp.conv_b; a.center = kcenter; a.cluster = kcluster;
    a.BM = BM; a.Mper = Mk; a.K = K; a.proxy = point_proxy; a.ln_eps = ln_eps;
    a.ksrc = ksrc; a.Msrc = Msrc; a.head_flag = head_flag; a.head_seq = head_seq;
    if (blk_t && xin_t) { a.n1w[0] = blk_t->norm1_w; a.n1b[0] = blk_t->norm1_b; a.posb[0] = posb_t; a.xin[0] = xin_t; }
    if (blk_i && xin_i) { a.n1w[1] = blk_i->norm1_w; a.n1b[1] = blk_i->norm1_b; a.posb[1] = posb_i; a.xin[1] = xin_i; }
    PTX_REQUIRE(width == 256 || width == 512, "pointnet: width=%d (supported: 256, 512)", width);
    if (width == 256) hipLaunchKernelGGL((k_slot_net<1, 4>), dim3(cdiv(BM, 4)), dim3(256), 0, st, a);
    else              hipLaunchKernelGGL((k_slot_net<1, 8>), dim3(cdiv(BM, 4)), dim3(256), 0, st, a);
    PTX_LAUNCHED("k_slot_net<pointnet>");
    return PTX_OK;
}

// ------------------------------------------------------------------------------ kept rows out of the early tables
// (the early-proxy path of ptx_forward, api.hip: point proxies, LayerNorm rows and their qkv projections are computed for the Mt
//  clusters that enter the farthest point sampling, beside it; afterwards the kept rows are a gather + the per-slot bias term)
struct QkvGatherArgs { const float *pp_all, *g[2], *tb[2]; const int32_t *ksrc; int M, Mk, C; float *pp, *qkv[2]; };
__global__ __launch_bounds__(512) void k_qkv_gather(QkvGatherArgs a)
{
    const int row = blockIdx.x, b = row / a.Mk, j = row - b * a.Mk;
    const size_t src = (size_t)b * a.M + a.ksrc[row];
    const int c4 = a.C / 4, n4 = 3 * c4;
    for (int i = threadIdx.x; i < c4 + 2 * n4; i += blockDim.x) {          // C = 256: 448 threads, one 16-byte piece each
        if (i < c4) {
            reinterpret_cast<float4 *>(a.pp + (size_t)row * a.C)[i] = reinterpret_cast<const float4 *>(a.pp_all + src * a.C)[i];
        } else {
            const int br = (i - c4) / n4, k = (i - c4) - br * n4;
            const float4 gv = reinterpret_cast<const float4 *>(a.g[br] + src * 3 * a.C)[k];
            const float4 tv = reinterpret_cast<const float4 *>(a.tb[br] + (size_t)j * 3 * a.C)[k];
            reinterpret_cast<float4 *>(a.qkv[br] + (size_t)row * 3 * a.C)[k] = make_float4(gv.x + tv.x, gv.y + tv.y, gv.z + tv.z, gv.w + tv.w);
        }
    }
}
int launch_qkv_gather(const float *pp_all, const float *const g[2], const float *const tb[2], const int32_t *ksrc, int B, int M, int Mk,
                      int C, float *point_proxy, float *const qkv[2], hipStream_t st)
{
    PTX_REQUIRE(pp_all && g[0] && g[1] && tb[0] && tb[1] && ksrc && point_proxy && qkv[0] && qkv[1] && C % 4 == 0, "qkv gather: bad arguments");
    QkvGatherArgs a{pp_all, {g[0], g[1]}, {tb[0], tb[1]}, ksrc, M, Mk, C, point_proxy, {qkv[0], qkv[1]}};
    const int pieces = C / 4 * 7;
    hipLaunchKernelGGL(k_qkv_gather, dim3(B * Mk), dim3(pieces <= 512 ? (pieces + 63) / 64 * 64 : 512), 0, st, a);
    PTX_LAUNCHED("k_qkv_gather");
    return PTX_OK;
}

// ------------------------------------------------------------------------------ cluster selection
// One work-group per scene, everything after the loads lives in LDS / registers.
//  1. stable counting sort of clusters by padding count (keys 0..K), first Mt   PRE:372-385
//  2. FPS over the re-ordered centres, Kd = Mt - Mk picks, start 0, first max   PRE:393
//  3. keep = ascending positions not picked, first Mk                           PRE:395-408
//  4. gathers + drop_idx                                                        PRE:411-418
//  5. ownership / drop tags for pt_replace and remove_points_by_index          PRE:478-495, 516-523
struct SelectArgs {
    const int32_t *idx; const float *centers; const float *cluster; const int32_t *pad_count;
    const int32_t *order_override;
    int32_t *order, *picks, *keep; float *kcenter, *kcluster; int32_t *kidx, *drop_idx; uint32_t *tag;
    int M, K, Mt, Mk, Kd, N;
    int32_t *ksrc;           // (B,Mk) row of kept cluster j in the un-gathered arrays (= order[keep[j]]), or null
    uint32_t *mm_clear;      // forward only: the encoded bounding boxes (B,6), last read before this launch, are
                             // zeroed here so that the next call finds them clean (no memset launch per call)
    GateRef tail;            // flag != null: the launch ends with the wait for another stream's word (api.hip, "cgate"): the kernel
                             // behind it on this stream starts when both the selection and that stream are done -- no barrier packet
};

// Step 1 of the selection for one scene (one work-group of 256 threads): the stable counting sort of the clusters by padding count
// (keys 0..K), first Mt (PRE:372-385), into s_order (LDS); s_hist: 64 ints of LDS.  Shared by k_select and k_order.
__device__ __forceinline__ void select_order(const int32_t *__restrict__ pc, const int32_t *__restrict__ order_override, int M, int Mt,
                                             int *s_order, int *s_hist)
{
    const int T = blockDim.x, tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    if (order_override != nullptr) {
        for (int t = tid; t < Mt; t += T) s_order[t] = order_override[t];
        __syncthreads();
        return;
    }
    if (tid < 64) s_hist[tid] = 0;
    __syncthreads();
    for (int m = tid; m < M; m += T) atomicAdd(&s_hist[pc[m]], 1);
    __syncthreads();
    if (wid == 0) {
        // lane v holds the next free position of bucket v (K + 1 <= 64 buckets)
        int cnt = s_hist[lane];
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
        int base = incl - cnt;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int m0 = 0; m0 < M; m0 += 64) {
            const int m = m0 + lane;
            const bool valid = m < M;
            const int c = valid ? pc[m] : -1;
            unsigned long long remaining = __ballot(valid);
            while (remaining) {
                const int leader = __ffsll((long long)remaining) - 1;
                const int v = __shfl(c, leader, 64);
                const bool mine = valid && c == v;
                const unsigned long long match = __ballot(mine);
                const int start = __shfl(base, v, 64);
                const int pos = start + __popcll(match & lt);
                if (mine && pos < Mt) s_order[pos] = m;
                if (lane == v) base += __popcll(match);
                remaining &= ~match;
            }
        }
    }
    __syncthreads();
}

// The ordering alone (early proxies, api.hip: the third stream needs WHICH Mt clusters enter the sampling -- not their selection --
// while k_select is still picking; the same code on the same counts gives the same order)
__global__ __launch_bounds__(256) void k_order(const int32_t *__restrict__ pad_count, const int32_t *__restrict__ order_override, int M,
                                               int Mt, int32_t *__restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_order = reinterpret_cast<int *>(smem), *s_hist = s_order + Mt;
    const int b = blockIdx.x;
    select_order(pad_count + (size_t)b * M, order_override ? order_override + (size_t)b * Mt : nullptr, M, Mt, s_order, s_hist);
    for (int t = threadIdx.x; t < Mt; t += blockDim.x) order[(size_t)b * Mt + t] = s_order[t];
}
int launch_order(const PtxShape &s, const int32_t *pad_count, const int32_t *order_override, int32_t *order, hipStream_t st)
{
    const int M = s.grid_size * s.grid_size * s.grid_size;
    hipLaunchKernelGGL(k_order, dim3(s.B), dim3(256), sizeof(int) * ((size_t)s.Mt + 64), st, pad_count, order_override, M, s.Mt, order);
    PTX_LAUNCHED("k_order");
    return PTX_OK;
}

// P points per lane of the FPS; ONE: a single wave holds all points (64 P >= Mt) and picks without any exchange or barrier
// (small Mt: the benchmark shape's 359 centres), otherwise the four waves share them (256 P >= Mt)
template <int P, bool ONE>
__global__ __launch_bounds__(256) void k_select(SelectArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = blockDim.x, tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const int b = blockIdx.x;
    const int M = a.M, Mt = a.Mt, Mk = a.Mk, Kd = a.Kd;
    // LDS carve
    int *s_order = reinterpret_cast<int *>(smem);                 // Mt
    float *sx = reinterpret_cast<float *>(s_order + Mt);          // Mt each
    float *sy = sx + Mt, *sz = sy + Mt;
    int *s_flag = reinterpret_cast<int *>(sz + Mt);               // Mt: 1 = picked by FPS
    int *s_keep = s_flag + Mt;                                    // Mk
    int *s_picks = s_keep + Mk;                                   // Kd
    int *s_hist = s_picks + (Kd > 0 ? Kd : 1);                    // 64
    // [2][4] FPS candidates of the four waves, (~index, distance bits), 16-byte aligned (select_lds_bytes leaves the room)
    uint2 *s_cand = reinterpret_cast<uint2 *>(smem + ((reinterpret_cast<unsigned char *>(s_hist + 64) - smem + 15) & ~(size_t)15));

    const int32_t *pc = a.pad_count + (size_t)b * M;
    // ---- 1. ordering
    select_order(pc, a.order_override ? a.order_override + (size_t)b * Mt : nullptr, M, Mt, s_order, s_hist);
    for (int t = tid; t < Mt; t += T) {
        const int src = s_order[t];
        a.order[(size_t)b * Mt + t] = src;
        const float *c = a.centers + ((size_t)b * M + src) * 3;
        sx[t] = c[0]; sy[t] = c[1]; sz[t] = c[2];
        s_flag[t] = 0;
    }
    __syncthreads();

    // ---- 2. farthest point sampling: sequential in k.  The four waves (one per SIMD) each keep a quarter of
    // the points and their running minimum distances in registers: thread g holds points g*P .. g*P+P-1 (blocked:
    // thread order == index order).  Per pick: P distance updates, a DPP row-rotate max and one ballot per wave,
    // the four wave candidates exchanged through LDS with one barrier, and every wave takes the same decision.
    // First-max tie-break (PRE:613): in-lane strict '>', across lanes the lowest set ballot bit, across waves the
    // lowest wave.  (One wave with all points, no barrier: 0.37 us per pick at Mt = 359 but 1.04 us at the
    // shipped configuration's Mt = 1210 -- VALU-bound on one SIMD, 540 us for its 519 picks.  Carrying the winner's
    // coordinates through the candidate exchange instead of re-reading sx[last]: slower, 233 vs 215 us at Mt = 1210 --
    // the per-lane select of the candidate's registers and three more LDS words cost more than the saved read.
    // Eight waves with P = 3: slower as well, 273 vs 211 us -- two waves per SIMD at the barrier; the pick is bound by
    // its synchronisation chain, not by the distance updates.
    //  r04: ONE wave with twenty points per lane (packed updates, no barrier, no exchange) at Mt = 1 210: slower too, cfg4 at 6
    //  scenes 0.533 vs 0.486 ms per step -- 160 dependent-issue VALU instructions per pick cost more than the barrier they save.
    //  r04: the exchange WITHOUT the barrier -- every wave stores its candidate with the pick number's low byte as a tag and polls
    //  the four words until all carry it: picks identical, 0.50 instead of 0.38 us per pick (cfg4, one scene: 0.401 vs 0.338 ms per
    //  step; cfg5 1.355 vs 1.152 ms): a poll is an LDS round trip and most picks need two of them; s_barrier is the cheap part.)
    const int kn = Kd < Mt ? Kd : Mt;                                            // PRE:595
    {
        // the whole scene waits on this latency-bound loop while bandwidth-bound kernels of the image branch
        // share the CU: give it issue priority
        __builtin_amdgcn_s_setprio(3);
        // Minimum distances are >= +0: their bit patterns order like the floats, so the argmax runs on unsigned
        // integers (v_max_u32 takes the DPP operand directly; a float max needs two more canonicalising ops per step).
        // Padding lanes hold distance 0: they tie with picked points only and lose by index (first max, PRE:613).
        // From six points per lane on (Mt > 1280) the points sit in PAIRS: the packed fp32 instructions (v_pk_add / v_pk_mul,
        // IEEE round-to-nearest like the scalar ones, never fused) update two distances at once -- 1.38 -> 1.18 ms per step
        // at the 2 868 centres of cfg5; with five points per lane the sixth, empty slot costs more than packing saves.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr bool PK = P >= 6;
        constexpr int PP = PK ? (P + 1) / 2 : P;               // register slots per coordinate
        constexpr int NE = PK ? 2 * PP : P;                    // distance slots
        f32x2 px[PP], py[PP], pz[PP]; uint32_t mind[NE];
        const int fid = ONE ? lane : tid;                      // position of this lane among the lanes that hold points
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int t = fid * P + i;
            const bool real = i < P && t < Mt;
            const int j = PK ? i >> 1 : i, e = PK ? i & 1 : 0;
            px[j][e] = real ? sx[t] : 0.0f; py[j][e] = real ? sy[t] : 0.0f; pz[j][e] = real ? sz[t] : 0.0f;
            mind[i] = real ? 0x7f800000u : 0u;          // +inf; padding: 0, never the FIRST maximum
        }
        int last = 0;
        if (tid == 0 && Kd > 0) s_picks[0] = 0;
        for (int k = 1; k < (ONE && wid != 0 ? 0 : kn); ++k) {
            const float lx = sx[last], ly = sy[last], lz = sz[last];
            uint32_t bv = 0u; int bi = 0;
            if (PK) {
                const f32x2 lx2 = {lx, lx}, ly2 = {ly, ly}, lz2 = {lz, lz};
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    // ((dx*dx)+dy*dy)+dz*dz per element, one rounding per operation (built with -ffp-contract=off)
                    const f32x2 dx = lx2 - px[j], dy = ly2 - py[j], dz = lz2 - pz[j];
                    const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int i = 2 * j + e;
                        const uint32_t u = min(mind[i], __float_as_uint(d2[e]));     // PRE:609 (non-negative floats: bit order)
                        mind[i] = u;
                        if (u > bv) { bv = u; bi = i; }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const float d2 = dist2_nofma(lx, ly, lz, px[i][0], py[i][0], pz[i][0]);
                    const uint32_t u = min(mind[i], __float_as_uint(d2));            // PRE:609
                    mind[i] = u;
                    if (u > bv) { bv = u; bi = i; }
                }
            }
            uint32_t g = bv;
            g = max(g, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x128, 0xf, 0xf, false));    // row_ror 8
            g = max(g, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x124, 0xf, 0xf, false));    // 4
            g = max(g, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x122, 0xf, 0xf, false));    // 2
            g = max(g, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g, 0x121, 0xf, 0xf, false));    // 1
            const uint32_t gmax = max(max((uint32_t)__builtin_amdgcn_readlane((int)g, 0), (uint32_t)__builtin_amdgcn_readlane((int)g, 16)),
                                      max((uint32_t)__builtin_amdgcn_readlane((int)g, 32), (uint32_t)__builtin_amdgcn_readlane((int)g, 48)));
            const unsigned long long who = __ballot(bv == gmax);
            const int leader = __ffsll((long long)who) - 1;
            const int cand = ((ONE ? 0 : wid * 64) + leader) * P + __builtin_amdgcn_readlane(bi, leader);
            if (ONE) {
                last = cand;
                if (lane == 0) s_picks[k] = last;
                continue;
            }
            const int par = (k & 1) * 4;                         // double-buffered: one barrier per pick
            // one 64-bit word per wave: distance bits above the complemented index, so that a plain unsigned maximum
            // is "largest distance, lowest index" -- and both halves of every word are wanted by the comparison itself
            // (with separate compares hipcc reads the distances first and fetches the winner's index afterwards)
            if (lane == 0) s_cand[par + wid] = make_uint2(~(uint32_t)cand, gmax);
            __syncthreads();
            const unsigned long long *cw = reinterpret_cast<const unsigned long long *>(s_cand + par);
            const unsigned long long bw = max(max(cw[0], cw[1]), max(cw[2], cw[3]));
            last = (int)~(uint32_t)bw;
            if (tid == 0) s_picks[k] = last;
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    for (int k = tid; k < Kd; k += T) {
        const int pk = k < kn ? s_picks[k] : -1;
        a.picks[(size_t)b * Kd + k] = pk;
        if (pk >= 0) s_flag[pk] = 1;            // duplicates write the same value
    }
    __syncthreads();

    // ---- 3. keep list: ascending positions that were not picked, first Mk (PRE:400-406)
    if (wid == 0) {
        const unsigned long long lt = (1ull << lane) - 1ull;
        int n = 0;
        for (int t0 = 0; t0 < Mt && n < Mk; t0 += 64) {
            const int t = t0 + lane;
            const bool kp = t < Mt && s_flag[t] == 0;
            const unsigned long long mask = __ballot(kp);
            const int pos = n + __popcll(mask & lt);
            if (kp && pos < Mk) s_keep[pos] = t;
            n += __popcll(mask);
        }
    }
    __syncthreads();

    // ---- 4. kept centres; the slot gathers, drop list and tags are spread over the chip by
    // k_select_slots (they are dependent-load chains: one work-group per scene made them the
    // longest part of this kernel once the image branch loads the memory system)
    if (a.mm_clear != nullptr && tid < 6) a.mm_clear[b * 6 + tid] = 0u;
    for (int j = tid; j < Mk; j += T) {
        const int t = s_keep[j];
        a.keep[(size_t)b * Mk + j] = t;
        if (a.ksrc != nullptr) a.ksrc[(size_t)b * Mk + j] = s_order[t];
        a.kcenter[((size_t)b * Mk + j) * 3] = sx[t];
        a.kcenter[((size_t)b * Mk + j) * 3 + 1] = sy[t];
        a.kcenter[((size_t)b * Mk + j) * 3 + 2] = sz[t];
    }
    if (a.tail.flag != nullptr && b == 0 && tid == 0) gate_wait(a.tail);
}

// One thread per slot of the kept clusters (gather xyz + idx, ownership tag) and of the dropped
// clusters (drop list, drop bit).  Ownership = last writer in flat (m,k) order = atomicMax of
// 1 + flat slot; the drop bit is OR-ed in concurrently, so an owner that finds the
// bit already set repeats its max above the bit: every interleaving ends at
// bit31 | max(owner slots).
__global__ __launch_bounds__(256) void k_select_slots(SelectArgs a)
{
    const int b = blockIdx.y;
    const int M = a.M, K = a.K, Mt = a.Mt, Mk = a.Mk, Kd = a.Kd;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int kn = Kd < Mt ? Kd : Mt;
    uint32_t *tag = a.tag ? a.tag + (size_t)b * a.N : nullptr;
    const int32_t *order = a.order + (size_t)b * Mt;
    if (e < Mk * K) {
        const int j = e / K, k = e - j * K;
        const int src = order[a.keep[(size_t)b * Mk + j]];
        const int id = a.idx[((size_t)b * M + src) * K + k];
        if (a.kidx) a.kidx[(size_t)b * Mk * K + e] = id;
        if (a.kcluster) {
            const float *cp = a.cluster + (((size_t)b * M + src) * K + k) * 3;
            float *op = a.kcluster + ((size_t)b * Mk * K + e) * 3;
            op[0] = cp[0]; op[1] = cp[1]; op[2] = cp[2];
        }
        if (tag && id >= 0) {
            const uint32_t v = (uint32_t)(e + 1);
            const uint32_t old = atomicMax(&tag[id], v);
            if (old & 0x80000000u) atomicMax(&tag[id], 0x80000000u | v);
        }
    } else if (e < (Mk + Kd) * K) {
        const int ed = e - Mk * K;
        const int kk = ed / K, k = ed - kk * K;
        const int pk = kk < kn ? a.picks[(size_t)b * Kd + kk] : -1;
        int id = -1;
        if (pk >= 0) id = a.idx[((size_t)b * M + order[pk]) * K + k];
        if (a.drop_idx) a.drop_idx[(size_t)b * Kd * K + ed] = id;
        if (tag && id >= 0) atomicOr(&tag[id], 0x80000000u);
    }
}

static size_t select_lds_bytes(const PtxShape &s)
{
    const int Kd = s.Mt - s.Mk;
    return sizeof(int) * ((size_t)s.Mt * 5 + s.Mk + (Kd > 0 ? Kd : 1) + 64 + 16 + 16);
}

static SelectArgs select_args(const PtxShape &s, const int32_t *idx, const float *centers, const float *cluster,
                              const int32_t *pad_count, const int32_t *order_override, int32_t *order,
                              int32_t *picks, int32_t *keep, float *kcenter, float *kcluster, int32_t *kidx,
                              int32_t *drop_idx, uint32_t *tag, uint32_t *mm_clear)
{
    return SelectArgs{idx, centers, cluster, pad_count, order_override, order, picks, keep, kcenter,
                      kcluster, kidx, drop_idx, tag, s.grid_size * s.grid_size * s.grid_size, s.K, s.Mt,
                      s.Mk, s.Mt - s.Mk, s.N, nullptr, mm_clear, GateRef{}};
}

// ordering + FPS + keep list + kept centres (one work-group per scene)
int launch_select_order(const PtxShape &s, const float *centers, const int32_t *pad_count,
                        const int32_t *order_override, int32_t *order, int32_t *picks, int32_t *keep,
                        float *kcenter, int32_t *ksrc, uint32_t *mm_clear, hipStream_t st, bool critical,
                        hipEvent_t done, const GateRef *tail)
{
    SelectArgs a = select_args(s, nullptr, centers, nullptr, pad_count, order_override, order, picks, keep, kcenter,
                               nullptr, nullptr, nullptr, nullptr, mm_clear);
    a.ksrc = ksrc;
    if (tail != nullptr) a.tail = *tail;
    size_t lds = select_lds_bytes(s);
    PTX_REQUIRE(lds <= 160 * 1024, "select: Mt=%d needs %zu B of LDS (> 160 KiB)", s.Mt, lds);
    // r04: where the step waits for this kernel (`critical`: the clustering chain owns the caller's stream -- the shipped gs = 12
    // configuration, cfg1, cfg5) the one work-group per scene asks for (nearly) the whole LDS of its CU, so that no work-group of
    // the image chain's passes on the other stream can become its neighbour and bring 8 or 16 more waves to the CU's four SIMDs:
    // the picks are a chain of dependent instructions and run slower with neighbours.  cfg4 at 6 scenes per GPU: 11.38k / 11.37k ->
    // 11.82k / 11.81k scenes/s (+3.8 %), one scene and cfg1 neutral.  Not at the benchmark shape: there the image chain is the
    // critical one and gives up four CUs for it (k_img_pool 57.5 -> 62 us, step -0.6 %; profiles/r04_select_alone_ab.txt).
    // (End of r04: up to 32 scenes per call instead of 8 -- cfg5 at 16 scenes +2 %, 12 +1.3 %, cfg4 at 9
    // +3 %, the rest within 1 %: profiles/r04_select_alone_cap_ab.txt.)
    const bool alone = critical;
    if (alone && s.B <= 32 && lds < 150 * 1024) lds = 150 * 1024;
    // one wave only when the step waits for this kernel: beside a longer image chain the four-wave form finishes early
    // enough and leaves the point-proxy / qkv kernels later, i.e. less of them under the pooling pass
    const bool one = critical && s.Mt <= 384;
    const int per = cdiv(s.Mt, one ? 64 : 256);
    const dim3 grid(s.B), block(256);
#define PTX_SEL(P_)                                                                          \
    do {                                                                                     \
        if (lds > 64 * 1024) {                                                               \
            PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select<P_, false>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            if (one && P_ <= 6)                                                              \
                PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select<(P_ <= 6 ? P_ : 1), true>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        }                                                                                    \
        /* `done` rides on the kernel's own completion signal: a separate event record is one more packet (~6 us) */ \
        if (one && P_ <= 6) hipExtLaunchKernelGGL((k_select<(P_ <= 6 ? P_ : 1), true>), grid, block, lds, st, nullptr, done, 0, a); \
        else hipExtLaunchKernelGGL((k_select<P_, false>), grid, block, lds, st, nullptr, done, 0, a); \
    } while (0)
    if (per <= 1) PTX_SEL(1);
    else if (per <= 2) PTX_SEL(2);
    else if (per <= 3) PTX_SEL(3);
    else if (per <= 4) PTX_SEL(4);
    else if (per <= 5) PTX_SEL(5);
    else if (per <= 6) PTX_SEL(6);
    else if (per <= 8) PTX_SEL(8);
    else if (per <= 12) PTX_SEL(12);
    else if (per <= 16) PTX_SEL(16);
    else { set_error("select: Mt=%d too large (max %d)", s.Mt, 256 * 16); return PTX_EINVAL; }
#undef PTX_SEL
    PTX_LAUNCHED("k_select");
    return PTX_OK;
}

// ownership / drop tags of every slot (+ the gathered copies kidx / kcluster / drop_idx where asked for), chip-wide
int launch_select_slots(const PtxShape &s, const int32_t *idx, const float *cluster, const int32_t *order,
                        const int32_t *picks, const int32_t *keep, float *kcluster, int32_t *kidx,
                        int32_t *drop_idx, uint32_t *tag, hipStream_t st)
{
    SelectArgs a = select_args(s, idx, nullptr, cluster, nullptr, nullptr, const_cast<int32_t *>(order),
                               const_cast<int32_t *>(picks), const_cast<int32_t *>(keep), nullptr, kcluster, kidx,
                               drop_idx, tag, nullptr);
    hipLaunchKernelGGL(k_select_slots, dim3(cdiv(s.Mt * s.K, 256), s.B), dim3(256), 0, st, a);
    PTX_LAUNCHED("k_select_slots");
    return PTX_OK;
}

int launch_select(const PtxShape &s, const int32_t *idx, const float *centers, const float *cluster,
                  const int32_t *pad_count, const int32_t *order_override, int32_t *order,
                  int32_t *picks, int32_t *keep, float *kcenter, float *kcluster, int32_t *kidx,
                  int32_t *drop_idx, uint32_t *tag, hipStream_t st)
{
    PTX_TRY(launch_select_order(s, centers, pad_count, order_override, order, picks, keep, kcenter, nullptr, nullptr, st, true, nullptr));
    // r05: the gathers chip-wide without the tag atomics (~50k device-scope atomics per scene: 72 us of the training step's
    // forward at 6 scenes), the tags by the forward path's k_tags (LDS-owned ranges, plain stores; it writes EVERY word of the tag
    // rows, so the caller's buffer need not be cleared): 72 -> 14 + 32 us
    PTX_TRY(launch_select_slots(s, idx, cluster, order, picks, keep, kcluster, kidx, drop_idx, nullptr, st));
    if (tag == nullptr) return PTX_OK;
    return launch_tags(s, idx, order, picks, nullptr, tag, nullptr, nullptr, nullptr, st, keep);
}

// ------------------------------------------------------------------------------ apply
// tag word per point: bit 31 = dropped (PRE:516-523), low 31 bits = 1 + owning flat (m,k) slot
// (last writer in flat order, PRE:495 single-thread semantics), 0 = untouched.
__global__ __launch_bounds__(256) void k_tile_count(const uint32_t *__restrict__ tag, int N,
                                                    int32_t *__restrict__ tile_counts, int32_t *scene_acc,
                                                    int32_t *counts)
{
    const int b = blockIdx.y, tile = blockIdx.x;
    const uint32_t *tg = tag + (size_t)b * N;
    int c = 0;
#pragma unroll
    for (int r = 0; r < kTilePts / 256; ++r) {
        const int n = tile * kTilePts + r * 256 + threadIdx.x;
        if (n < N) c += (tg[n] >> 31) == 0;
    }
    c = wave_sum(c);
    __shared__ int red[4];
    if (lane_id() == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = red[0] + red[1] + red[2] + red[3];
        tile_counts[b * gridDim.x + tile] = total;
        if (scene_acc != nullptr) {
            // Surviving points per scene, published as soon as the drop tags are final so that the host can size
            // its output views (PRE:467 masked_select lengths) long before the forward has drained: the last tile
            // of a scene to arrive (ticket) stores the sum with system scope -- `counts` is normally pinned host
            // memory that the caller polls -- and leaves the two accumulators zero for the next call.
            __hip_atomic_fetch_add(scene_acc + 2 * b, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ticket = __hip_atomic_fetch_add(scene_acc + 2 * b + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == (int)gridDim.x - 1) {
                const int sum = __hip_atomic_exchange(scene_acc + 2 * b, 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(scene_acc + 2 * b + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(counts + b, sum, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Ownership / drop tags AND the per-tile survivor counts of one range of points in one work-group, without a single
// global atomic (forward path; k_select_slots + k_tile_count are the stage-API form).  The work-group owns the tag words of
// kTagRange consecutive points in LDS, scans every slot of its scene -- kept clusters: LDS max of 1 + flat slot (last
// writer in flat order), then, after a barrier, dropped clusters: LDS or of bit 31 -- and writes the range
// out with plain stores together with its tiles' counts.  (The ~50k device-scope atomics of k_select_slots ran next to
// the kernels the step waits for and stretched them: 23 us of the 401 us step at the reference's gs = 12 shape.)
// TILES 2048-point tiles per work-group: 16 (32768 points, 128 KB of LDS) when there are many scenes, 4 when a single
// scene would otherwise sit on four work-groups (each of them scans every slot of the scene)
struct TagArgs {
    const int32_t *idx, *order, *picks, *ksrc;  // ksrc (B,Mk): source row of kept cluster j (= order[keep[j]]); null: read through keep
    const int32_t *keep;                        // (B,Mk) rank of kept cluster j in the order (stage API: k_select did not write ksrc)
    uint32_t *tag; int32_t *tile_counts, *scene_acc, *counts;
    int M, K, Mt, Mk, Kd, N, ntiles;
};

template <int TILES>
__global__ __launch_bounds__(1024) void k_tags(TagArgs a)
{
    constexpr int kTagRange = TILES * kTilePts;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_tag[];      // kTagRange words + 16 tile counters
    int *s_cnt = reinterpret_cast<int *>(s_tag + kTagRange);
    const int b = blockIdx.y, base = blockIdx.x * kTagRange, tid = threadIdx.x;
    const int M = a.M, K = a.K, Mt = a.Mt, Mk = a.Mk, Kd = a.Kd;
    const int kn = Kd < Mt ? Kd : Mt;
    for (int i = tid * 4; i < kTagRange; i += 1024 * 4) *reinterpret_cast<uint4 *>(s_tag + i) = make_uint4(0u, 0u, 0u, 0u);
    if (tid < TILES) s_cnt[tid] = 0;
    __syncthreads();
    const int32_t *idx = a.idx + (size_t)b * M * K;
    const int32_t *order = a.order + (size_t)b * Mt;
    // (four slots per thread and trip: the two dependent loads of a slot are the cost of this scan, not its arithmetic)
    for (int e0 = tid; e0 < Mk * K; e0 += 4 * 1024) {           // owners
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 1024;
            const int ec = min(e, Mk * K - 1), j = ec / K, k = ec - j * K;
            const int src = a.ksrc != nullptr ? a.ksrc[(size_t)b * Mk + j] : order[a.keep[(size_t)b * Mk + j]];
            id[u] = e < Mk * K ? idx[(size_t)src * K + k] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned r = (unsigned)(id[u] - base);        // id = -1 (padding) is out of range too
            if (r < (unsigned)kTagRange) atomicMax(&s_tag[r], (uint32_t)(e0 + u * 1024 + 1));
        }
    }
    __syncthreads();
    for (int e0 = tid; e0 < Kd * K; e0 += 4 * 1024) {           // drops
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ed = e0 + u * 1024;
            const int ec = min(ed, Kd * K - 1), kk = ec / K, k = ec - kk * K;
            const int pk = (ed < Kd * K && kk < kn) ? a.picks[(size_t)b * Kd + kk] : -1;
            id[u] = pk >= 0 ? idx[(size_t)order[pk] * K + k] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned r = (unsigned)(id[u] - base);
            if (r < (unsigned)kTagRange) atomicOr(&s_tag[r], 0x80000000u);
        }
    }
    __syncthreads();
    // write-out (16-B stores when the scene's tag row allows it) + survivors per tile
    uint32_t *tg = a.tag + (size_t)b * a.N;
    const int lim = min(kTagRange, a.N - base);                 // valid words of this range
    const bool vec = ((a.N & 3) == 0);
    for (int i0 = 0; i0 < lim; i0 += 1024 * 4) {               // wave-uniform trip count: the wave reduction needs all lanes
        const int i = i0 + tid * 4;
        int c = 0;
        if (i < lim) {
            const uint4 v = *reinterpret_cast<const uint4 *>(s_tag + i);
            if (vec) {                                          // then lim is a multiple of 4 as well
                *reinterpret_cast<uint4 *>(tg + base + i) = v;
                c = (int)((v.x >> 31) == 0) + (int)((v.y >> 31) == 0) + (int)((v.z >> 31) == 0) + (int)((v.w >> 31) == 0);
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i + q < lim) { tg[base + i + q] = w[q]; c += (int)((w[q] >> 31) == 0); }
            }
        }
        // the 64 lanes of a wave cover 256 consecutive points: one tile (2048 points) per 8 waves and pass
        c = wave_sum(c);
        if (lane_id() == 0 && i < kTagRange) atomicAdd(&s_cnt[i / kTilePts], c);
    }
    __syncthreads();
    const int t0 = blockIdx.x * TILES;
    if (a.tile_counts != nullptr && tid < TILES && t0 + tid < a.ntiles) a.tile_counts[b * a.ntiles + t0 + tid] = s_cnt[tid];
    if (tid == 0 && a.scene_acc != nullptr) {
        int total = 0;
        for (int t = 0; t < TILES; ++t) total += s_cnt[t];
        // the last range of a scene to arrive publishes the scene's count (see k_tile_count)
        __hip_atomic_fetch_add(a.scene_acc + 2 * b, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ticket = __hip_atomic_fetch_add(a.scene_acc + 2 * b + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (int)gridDim.x - 1) {
            const int sum = __hip_atomic_exchange(a.scene_acc + 2 * b, 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.scene_acc + 2 * b + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.counts + b, sum, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int launch_tags(const PtxShape &s, const int32_t *idx, const int32_t *order, const int32_t *picks, const int32_t *ksrc,
                uint32_t *tag, int32_t *tile_counts, int32_t *counts, int32_t *scene_acc, hipStream_t st, const int32_t *keep)
{
    const int M = s.grid_size * s.grid_size * s.grid_size;
    TagArgs a{idx, order, picks, ksrc, keep, tag, tile_counts, scene_acc, counts, M, s.K, s.Mt, s.Mk, s.Mt - s.Mk, s.N,
              cdiv(s.N, kTilePts)};
    const bool small = (long)s.B * cdiv(s.N, 16 * kTilePts) < 16;
    if (small) {
        const size_t lds = sizeof(uint32_t) * 4 * kTilePts + 16 * sizeof(int);
        hipLaunchKernelGGL(k_tags<4>, dim3(cdiv(s.N, 4 * kTilePts), s.B), dim3(1024), lds, st, a);
    } else {
        const size_t lds = sizeof(uint32_t) * 16 * kTilePts + 16 * sizeof(int);
        // (per launch, like k_select: the attribute is per device, a process may drive several)
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tags<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_tags<16>, dim3(cdiv(s.N, 16 * kTilePts), s.B), dim3(1024), lds, st, a);
    }
    PTX_LAUNCHED("k_tags");
    return PTX_OK;
}

// stand-alone form of the per-scene sum (stage API without accumulators)
__global__ __launch_bounds__(64) void k_scene_counts(const int32_t *tile_counts, int ntiles, int32_t *counts)
{
    const int b = blockIdx.x;
    int c = 0;
    for (int t = threadIdx.x; t < ntiles; t += 64) c += tile_counts[b * ntiles + t];
    c = wave_sum(c);
    if (threadIdx.x == 0) {
        __hip_atomic_store(counts + b, c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// scene_acc (B,2) int32, zero on entry, left zero: per-scene sum + arrival ticket; null = separate k_scene_counts launch
int launch_tile_count(const uint32_t *tag, int B, int N, int32_t *tile_counts, int32_t *counts, int32_t *scene_acc,
                      hipStream_t st)
{
    const int ntiles = cdiv(N, kTilePts);
    hipLaunchKernelGGL(k_tile_count, dim3(ntiles, B), dim3(256), 0, st, tag, N, tile_counts,
                       counts ? scene_acc : nullptr, counts);
    PTX_LAUNCHED("k_tile_count");
    if (counts && scene_acc == nullptr) {
        hipLaunchKernelGGL(k_scene_counts, dim3(B), dim3(64), 0, st, tile_counts, ntiles, counts);
        PTX_LAUNCHED("k_scene_counts");
    }
    return PTX_OK;
}

struct AffineArgs {
    ScenePts points; uint32_t *tag; const float *kcenter, *translate, *transform;
    float *out; int32_t *counts; const int32_t *tile_counts; int N, Mk, K;
    int clear_tag;      // forward only: this is the last reader of the tags; leave them zero for the next call
    const uint32_t *poison;     // forward with stream gates: nonzero when a gate timed out (common.h, gate_wait) -- the outputs are NaN then
};

// TABLE: the scene's (centre, transform, translate) rows are staged in LDS (Mk x 15 floats) and every global load of the
// work-group -- table, tags, points, the tile counts in front of this tile -- is requested before the first is waited
// for: ONE round trip to memory.  (r03; before, a thread walked its eight points one after the other, each a tag load, a
// wait, fifteen dependent gather loads, a wait: sixteen round trips in sequence were the whole 9 us of the launch.)
// Without TABLE (more than 1024 kept clusters: the rows do not fit) the per-point gathers stay.
template <bool COMPACT, bool TABLE>
__global__ __launch_bounds__(256) void k_affine(AffineArgs a)
{
    extern __shared__ float s_tab[];                        // TABLE: [3 Mk centres | 9 Mk transforms | 3 Mk translations]
    const int b = blockIdx.y, tile = blockIdx.x, ntiles = gridDim.x;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    uint32_t *tg = a.tag + (size_t)b * a.N;
    const float *__restrict__ pts = a.points.p[b];
    float *out = a.out + (size_t)b * a.N * 3;
    constexpr int R = kTilePts / 256;
    __shared__ int s_cnt[R][4];
    __shared__ int s_base;
    __shared__ uint32_t s_poison;
    const float *kc = a.kcenter + (size_t)b * a.Mk * 3, *kT = a.transform + (size_t)b * a.Mk * 9, *kt = a.translate + (size_t)b * a.Mk * 3;
    // ---- requests: the counts of the tiles in front (one per thread), the table, then tags and points
    int acc = 0;
    if (COMPACT) acc = a.tile_counts[b * ntiles + min(tid, ntiles - 1)];
    // (ONE thread reads the word and the work-group takes its value behind the barrier below: the poisoned path leaves the kernel
    //  early and must be taken by every thread or by none, also when the word is stored while this work-group runs)
    if (tid == 0) s_poison = a.poison != nullptr ? __hip_atomic_load(a.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    constexpr int kBatch = 8;
    const int c3 = 3 * a.Mk, c12 = 12 * a.Mk, c15 = 15 * a.Mk;
    auto tab_src = [&](int i) {                             // (the address is selected, not the load: no branch per request)
        const float *p = i < c3 ? kc + i : i < c12 ? kT + (i - c3) : kt + (i - c12);
        return *p;
    };
    constexpr int kAhead = 4;                               // batches requested up front: 8192 floats = 546 clusters
    float tb[kAhead][kBatch];
    if (TABLE) {
#pragma unroll
        for (int h = 0; h < kAhead; ++h)
            if (h * kBatch * 256 < c15) {                   // (work-group uniform)
#pragma unroll
                for (int u = 0; u < kBatch; ++u) tb[h][u] = tab_src(min((h * kBatch + u) * 256 + tid, c15 - 1));
            }
    }
    uint32_t tgv[R]; float v[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int nc = min(tile * kTilePts + r * 256 + tid, a.N - 1);
        tgv[r] = tg[nc];
        v[r][0] = pts[(size_t)nc * 3]; v[r][1] = pts[(size_t)nc * 3 + 1]; v[r][2] = pts[(size_t)nc * 3 + 2];
    }
    if (TABLE) {
#pragma unroll
        for (int h = 0; h < kAhead; ++h)
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int i = (h * kBatch + u) * 256 + tid;
                if (i < c15) s_tab[i] = tb[h][u];
            }
        for (int i0 = kAhead * kBatch * 256; i0 < c15; i0 += kBatch * 256) {      // more than 546 kept clusters
            float t2[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) t2[u] = tab_src(min(i0 + u * 256 + tid, c15 - 1));
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int i = i0 + u * 256 + tid;
                if (i < c15) s_tab[i] = t2[u];
            }
        }
    }
    int base = 0;
    if (COMPACT) {
        acc = tid < tile ? acc : 0;
        for (int t = tid + 256; t < tile; t += 256) acc += a.tile_counts[b * ntiles + t];      // more than 256 tiles per scene
        acc = wave_sum(acc);
        if (lane == 0) s_cnt[0][wid] = acc;
    }
    __syncthreads();                                        // table (and count partials) visible
    if (s_poison != 0u) {
        // A stream gate of this forward timed out: the chains were not ordered, so the tags, the tile counts and the transform rows
        // may be ANYTHING (r05: a garbage tile count sent the compacted stores out of bounds -- a GPU memory fault instead of the
        // reported error).  Nothing index-like is used: every point's row becomes NaN in place, the tags are left for the
        // re-initialisation that follows the error (module: ws_dirty; C callers: ptx_workspace_init, proxyt.h).
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = tile * kTilePts + r * 256 + tid;
            if (n < a.N) { out[(size_t)n * 3] = out[(size_t)n * 3 + 1] = out[(size_t)n * 3 + 2] = __uint_as_float(0x7fc00000u); }
        }
        return;
    }
    if (COMPACT) {
        base = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
        __syncthreads();                                    // s_cnt[0] is re-used below
    }
    (void)s_base;
    bool keep[R]; unsigned long long bal[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = tile * kTilePts + r * 256 + tid;
        keep[r] = false;
        if (n < a.N) {
            const uint32_t t = tgv[r];
            if (a.clear_tag && t != 0u) tg[n] = 0u;
            float x = v[r][0], y = v[r][1], z = v[r][2];
            keep[r] = !COMPACT || (t >> 31) == 0;
            const uint32_t own = t & 0x7fffffffu;
            if (own != 0 && keep[r]) {
                const int j = (int)(own - 1) / a.K;
                float c[3], T[9], tr[3];
                if (TABLE) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) { c[d] = s_tab[3 * j + d]; tr[d] = s_tab[c12 + 3 * j + d]; }
#pragma unroll
                    for (int d = 0; d < 9; ++d) T[d] = s_tab[c3 + 9 * j + d];
                } else {
#pragma unroll
                    for (int d = 0; d < 3; ++d) { c[d] = kc[3 * j + d]; tr[d] = kt[3 * j + d]; }
#pragma unroll
                    for (int d = 0; d < 9; ++d) T[d] = kT[9 * j + d];
                }
                const float dx = x - c[0], dy = y - c[1], dz = z - c[2];
                // (T (p-c)^T)^T + c + t     PRE:462
                const float rx = fmaf(T[2], dz, fmaf(T[1], dy, T[0] * dx));
                const float ry = fmaf(T[5], dz, fmaf(T[4], dy, T[3] * dx));
                const float rz = fmaf(T[8], dz, fmaf(T[7], dy, T[6] * dx));
                x = (rx + c[0]) + tr[0]; y = (ry + c[1]) + tr[1]; z = (rz + c[2]) + tr[2];
            }
            v[r][0] = x; v[r][1] = y; v[r][2] = z;
        }
        bal[r] = __ballot(keep[r]);
        if (COMPACT && lane == 0) s_cnt[r][wid] = __popcll(bal[r]);
    }
    if (!COMPACT) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = tile * kTilePts + r * 256 + tid;
            if (n < a.N) { out[(size_t)n * 3] = v[r][0]; out[(size_t)n * 3 + 1] = v[r][1]; out[(size_t)n * 3 + 2] = v[r][2]; }
        }
        return;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    int run = base;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int before = 0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) before += ww < wid ? s_cnt[r][ww] : 0;
        if (keep[r]) {
            const size_t pos = (size_t)(run + before + __popcll(bal[r] & lt));
            out[pos * 3] = v[r][0]; out[pos * 3 + 1] = v[r][1]; out[pos * 3 + 2] = v[r][2];
        }
        run += s_cnt[r][0] + s_cnt[r][1] + s_cnt[r][2] + s_cnt[r][3];
    }
}

int launch_affine(const PtxShape &s, const ScenePts &points, uint32_t *tag, const float *kcenter,
                  const float *translate, const float *transform, float *out, int32_t *counts,
                  const int32_t *tile_counts, bool compact, bool clear_tag, hipStream_t st, const uint32_t *poison)
{
    AffineArgs a{points, tag, kcenter, translate, transform, out, counts, tile_counts, s.N, s.Mk, s.K, clear_tag ? 1 : 0, poison};
    const dim3 grid(cdiv(s.N, kTilePts), s.B), block(256);
    const size_t tab = (size_t)s.Mk * 15 * sizeof(float);
    if (tab <= 62 * 1024) {
        if (compact) hipLaunchKernelGGL((k_affine<true, true>), grid, block, tab, st, a);
        else         hipLaunchKernelGGL((k_affine<false, true>), grid, block, tab, st, a);
    } else {
        if (compact) hipLaunchKernelGGL((k_affine<true, false>), grid, block, 0, st, a);
        else         hipLaunchKernelGGL((k_affine<false, false>), grid, block, 0, st, a);
    }
    PTX_LAUNCHED("k_affine");
    return PTX_OK;
}

}  // namespace ptx
