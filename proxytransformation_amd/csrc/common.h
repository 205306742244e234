// Internal helpers shared by the gfx950 translation units of libproxyt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/proxyt.h"

namespace ptx {

void set_error(const char *fmt, ...);
// per-kernel timing (api.hip): a launcher that can attach events to its kernel's own dispatch packet takes the armed pair
bool timing_ext_take(hipEvent_t *start, hipEvent_t *stop);

#define PTX_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            ::ptx::set_error(__VA_ARGS__);                       \
            return PTX_EINVAL;                                   \
        }                                                        \
    } while (0)

#define PTX_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            ::ptx::set_error("%s failed: %s", #expr, hipGetErrorString(e_));       \
            return PTX_ELAUNCH;                                                    \
        }                                                                          \
    } while (0)

#define PTX_LAUNCHED(name)                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ::ptx::set_error("launch %s failed: %s", name, hipGetErrorString(e_)); \
            return PTX_ELAUNCH;                                                    \
        }                                                                          \
    } while (0)

#define PTX_TRY(expr)                 \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != PTX_OK) return rc_;\
    } while (0)

constexpr int kWave = 64;
constexpr int kSlotHidden = 256;   // OffsetNetwork width (PRE:31); SimplifiedPointNet is as wide as embed_dim (PRE:302: 256)
constexpr int kTilePts = 2048;     // points per compaction tile

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// All-lanes wave reductions without LDS traffic: 4 DPP row rotations leave every lane with its
// 16-lane row result, v_readlane of one lane per row combines the 4 rows on the scalar side.
#define PTX_ROR_F(v, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, false))
#define PTX_ROR_I(v, n) __builtin_amdgcn_update_dpp(0, (v), 0x120 + (n), 0xf, 0xf, false)
#define PTX_LANE_F(v, l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (l)))
__device__ __forceinline__ float wave_sum(float v) {
    v += PTX_ROR_F(v, 8); v += PTX_ROR_F(v, 4); v += PTX_ROR_F(v, 2); v += PTX_ROR_F(v, 1);
    return (PTX_LANE_F(v, 0) + PTX_LANE_F(v, 16)) + (PTX_LANE_F(v, 32) + PTX_LANE_F(v, 48));
}
__device__ __forceinline__ int wave_sum(int v) {
    v += PTX_ROR_I(v, 8); v += PTX_ROR_I(v, 4); v += PTX_ROR_I(v, 2); v += PTX_ROR_I(v, 1);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, PTX_ROR_F(v, 8)); v = fmaxf(v, PTX_ROR_F(v, 4)); v = fmaxf(v, PTX_ROR_F(v, 2)); v = fmaxf(v, PTX_ROR_F(v, 1));
    return fmaxf(fmaxf(PTX_LANE_F(v, 0), PTX_LANE_F(v, 16)), fmaxf(PTX_LANE_F(v, 32), PTX_LANE_F(v, 48)));
}
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, PTX_ROR_F(v, 8)); v = fminf(v, PTX_ROR_F(v, 4)); v = fminf(v, PTX_ROR_F(v, 2)); v = fminf(v, PTX_ROR_F(v, 1));
    return fminf(fminf(PTX_LANE_F(v, 0), PTX_LANE_F(v, 16)), fminf(PTX_LANE_F(v, 32), PTX_LANE_F(v, 48)));
}
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_max(v); }
// order-preserving float <-> uint32 (larger float <=> larger uint)
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// squared distance exactly as pytorch3d's CPU loops accumulate it: ((dx*dx)+dy*dy)+dz*dz,
// one rounding per operation, never contracted into an FMA.
__device__ __forceinline__ float dist2_nofma(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// GELU(erf) of the eval path: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute) with the hardware
// reciprocal and exponential -- a third of libm erff's instructions (16 values per lane in fc1's epilogue); the error in
// GELU is <= |x| * 1e-7, against the 5e-5 bar of the float stages.  (Train mode keeps erff: its backward differentiates it.)
__device__ __forceinline__ float gelu_erf(float x)
{
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfz = 1.0f - p * t * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erfz, x));
}

// ---- stream gates (api.hip): ordering two HIP streams through a device word --------------------------------------
// A kernel of stream X waits until a kernel of stream Y has stored the forward's sequence number into `flag`.  The wait is
// bounded in WALL-CLOCK time (s_memrealtime ticks, 100 MHz on gfx950): when the bound runs out the waiter does not let go
// silently -- it stores a nonzero word into `err` (device-mapped pinned host memory the host checks: ptx_context_check, the
// next ptx_forward) and into `poison` (device memory: k_affine turns the outputs of that forward into NaN), and traps
// if asked to (PTX_GATE_TRAP=1: the queue is torn down, the process aborts).
struct GateRef {
    const uint32_t *flag; uint32_t seq;
    uint32_t *err;              // [host, pinned] sticky error word: 0x80000000 | site << 24 | low 24 bits of seq
    uint32_t *poison;           // device word read by k_affine
    uint64_t ticks;             // bound in s_memrealtime ticks
    uint32_t site;              // 1 fork, 2 join, 3 probe (fork direction), 4 probe (join direction)
    int trap;
};
__device__ __forceinline__ void gate_wait(const GateRef &g)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned it = 0;; ++it) {
        if ((int32_t)(__hip_atomic_load(g.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.seq) >= 0) return;
        if (it < 64) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(64);
        if ((it & 7u) == 7u && __builtin_amdgcn_s_memrealtime() - t0 > g.ticks) break;
    }
    if (g.poison) __hip_atomic_store(g.poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g.err) __hip_atomic_store(g.err, 0x80000000u | (g.site << 24) | (g.seq & 0xffffffu), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (g.trap) __builtin_trap();
}

// Per-scene base pointers of the point clouds, passed by value as a kernel argument (see make_scene_pts)
constexpr int kMaxScenes = 32;
struct ScenePts { const float *p[kMaxScenes]; };
// ---- per-scene bounding boxes (PRE:37-38) -----------------------------------------------------------------------------
// mm_enc[b][0..2] = ~ord(min_d)   (so that atomicMax over a zeroed word yields the minimum), mm_enc[b][3..5] = ord(max_d).
// One work-group reduces chunk `chunk` of `nchunks` of scene b's points and folds its result into the scene's six words
// with RETURNING atomics (the caller may publish a ticket afterwards: the atomics have been performed by then).
// 4 points = 12 floats = 3 float4; a thread's quads are all requested before the first is reduced (clamped: a repeated
// quad does not change a minimum).  As a loop with one quad per trip the pass was four dependent round trips next to the
// mean pass, which keeps the memory system loaded: 20 us for 4.8 MB (r03).
template <bool WAIT>     // WAIT: the six atomics are RETURNING and waited for (a ticket follows); else fire and forget
__device__ __forceinline__ void minmax_block(const float *__restrict__ p, int N, uint32_t *__restrict__ mm_enc, int b,
                                             int chunk, int nchunks, float (*red)[6])
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int tid = chunk * 256 + threadIdx.x;
    const int nth = nchunks * 256;
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    if (vec) {
        const float4 *p4 = reinterpret_cast<const float4 *>(p);
        const int nq = N >> 2;
        constexpr int U = 4;
        for (int q0 = tid; q0 < nq; q0 += nth * U) {         // one trip with minmax_chunks()'s decomposition
            float4 v[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = min(q0 + u * nth, nq - 1);
                v[u][0] = p4[3 * q]; v[u][1] = p4[3 * q + 1]; v[u][2] = p4[3 * q + 2];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 a = v[u][0], c = v[u][1], e = v[u][2];
                // a = x0 y0 z0 x1 | c = y1 z1 x2 y2 | e = z2 x3 y3 z3
                const float xs[4] = {a.x, a.w, c.z, e.y}, ys[4] = {a.y, c.x, c.w, e.z}, zs[4] = {a.z, c.y, e.x, e.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[0] = fminf(lo[0], xs[i]); hi[0] = fmaxf(hi[0], xs[i]);
                    lo[1] = fminf(lo[1], ys[i]); hi[1] = fmaxf(hi[1], ys[i]);
                    lo[2] = fminf(lo[2], zs[i]); hi[2] = fmaxf(hi[2], zs[i]);
                }
            }
        }
    } else {
        for (int i = tid; i < N; i += nth) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v = p[(size_t)i * 3 + d];
                lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = wave_min(lo[d]); hi[d] = wave_max(hi[d]); }
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { red[w][d] = lo[d]; red[w][3 + d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        float v = red[0][d];
        for (int i = 1; i < 4; ++i) v = d < 3 ? fminf(v, red[i][d]) : fmaxf(v, red[i][d]);
        uint32_t old = 0u;
        if (d < 3) { if (v != INFINITY) old = atomicMax(&mm_enc[b * 6 + d], ~f2ord(v)); }
        else       { if (v != -INFINITY) old = atomicMax(&mm_enc[b * 6 + d], f2ord(v)); }
        if (WAIT) asm volatile("" :: "v"(old));             // the returned value is waited for: the atomic has been performed
    }
}
inline int minmax_chunks(int N) { int c = (N + 256 * 16 - 1) / (256 * 16); return c < 1 ? 1 : (c > 256 ? 256 : c); }

// The fork of the two chains (api.hip, "gates"): the clustering stream waits for `gate` to reach gate_seq; the first thread of the
// image chain's first launch stores it (the stream being in order, everything the caller enqueued before the forward has
// completed by then).  (r04 also built the bounding boxes into the first work-groups of that launch, the fork's word stored when
// they were final: the mean launch took 36 instead of 32 us on the chain the step waits for, 18.95k -> 18.55k scenes/s --
// profiles/r04_minmax_fuse_ab.txt; removed in r05.)
__device__ __forceinline__ void mean_prologue(uint32_t *gate, uint32_t gate_seq)
{
    if (gate != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(gate, gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- parameter-only tables (ptx_prepare) ------------------------------------
struct PrepLayout {
    // all offsets in floats from the start of `prep`
    size_t off_ab, enc_ab;          // (2,256), (2,C): alpha, beta of the folded eval BatchNorm2d
    size_t ttn_ab, itn_ab;          // (2,3), (2,9) BatchNorm1d alpha/beta
    size_t posb_t, posb_i;          // (Mk,C) per-slot bias tables (PRE:212-215)
    size_t w3, b3;                  // (3C,in_dim), (3C): [q | k0 | v0] of token 0 from the image mean
    size_t t1;                      // (heads, KT1, hd): scale * [WkWc | K-proj of (bc+pos_i)]
    size_t t2;                      // (heads, hd, KT2p): [WvWc | V-proj of (bc+pos_i)] (transposed)
    // LayerNorm folds (GemmProb::lnp_in): norm_img -> proxy_proj of the image block; norm2 -> fc1 of both blocks
    size_t ppg_w, ppg_s, ppg_c;     // (C,C), (C), (C)
    size_t fc1g_w[2], fc1g_s[2], fc1g_c[2];   // (hidden,C), (hidden), (hidden) for the text / image block
    size_t mlp_w1p[2], mlp_w2p[2];  // fused Mlp (mlp.hip): fc1g_w / fc2_w as three bf16 planes in MFMA fragment order
    size_t qkvb[2];                 // (Mk,3C): slot-bias rows through the qkv projection + its bias (the early-proxy path, api.hip)
    size_t total;                   // floats
    int KT1, KT2p, hd;
};
PrepLayout prep_layout(const PtxShape &s);

// ---- per-call scratch ---------------------------------------------------------
struct WsLayout {
    // byte offsets from the start of the workspace
    size_t zero_begin, zero_bytes;  // region that every forward finds zero and leaves zero (cleared once, by
                                    // ptx_workspace_init; the kernels that read these words last re-zero them)
    size_t mm_enc;                  // (B,6) uint32 encoded min / max
    size_t scene_acc;               // (B,2) int32 survivor-count accumulator + arrival ticket
    size_t tag;                     // (B,N) uint32
    size_t fa_ticket;               // (2, B*heads) int32 arrival tickets of the split fused attention
    size_t mlp_ticket;              // (2, row tiles) int32 arrival tickets of the fused Mlp
    size_t mm_ticket;               // one int32: arrival ticket of the bounding-box work-groups inside the mean launch
    size_t minmax, centers0, cluster1, offsets, centers, idx2, cluster2, pad_count;
    size_t order, picks, keep, ksrc, kcenter, kcluster, kidx, drop_idx, tile_counts;
    size_t point_proxy, x_in[2];    // x_in: LN1(x)+slot bias per branch (B*Mk,C)
    size_t pp_all, xln_all[2], g_all[2];   // early-proxy path: point proxies / LN1 rows (B*Mt,C) and their qkv rows (B*Mt,3C) of the Mt
    size_t order_e;                        // clusters that enter the sampling, in its order (order_e: the ordering, computed beside k_select)
    size_t fm, qkv0, we, pool, gbuf, obuf, cbuf, img_proxy;
    size_t qkv[2], pt[2], pv[2], ao[2], x1[2], xn2[2], hbuf[2], x2[2], guide[2], head[2];
    size_t lnp_img, lnp_x1[2];      // LayerNorm partials (rows, C/32, 2) of c_proj's / proj's output
    size_t fa_part;                 // (2, B*heads, split, Mk, 34) partial results of the split fused attention
    size_t mlp_part;                // (2, row tiles, 4, 32, 256) partial fc2 sums of the fused Mlp
    size_t total;
};
WsLayout ws_layout(const PtxShape &s);

int validate_shape(const PtxShape &s);

// ---- grouped NT GEMM (gemm.hip) --------------------------------------------------
enum : int { EPI_NONE = 0, EPI_GELU = 1 };
struct GemmProb {
    const float *A; const float *W; float *C;
    const float *bias;      // (N) or null
    const float *res;       // residual (R,N) ld = ldres, or null
    const float *rs;        // per-row scalar, element r*rs_stride, or null
    const float *ad;        // addend (R,N) ld = ldad, multiplied by rs
    int R, N, K, lda, ldw, ldc, ldres, rs_stride, ldad, epi;
    // A operand produced on the fly from the per-tile results of k_img_pool (imgpool.hip) instead of read from
    // `A` (null then): row r, column k = [ G0[r][k] c0 + G1[r][k] c1  (k < kg) | E[r][k - kg] scale(k - kg) ]
    // with the split-softmax factors of row r computed from ML[r] = (m0, l0, m1, l1, s(0)); the row scalar
    // `rs` is then a_h(0) = ct (gemm.hip, k_gemm32<SK, 1>)
    const float *pg, *pe, *pml; int ldg, gslab, lde, ldml, kg;
    // LayerNorm folded across a GEMM -> GEMM seam (no LayerNorm launch, no normalised copy of the rows):
    //   producer (lnp_out != null): besides C, every 32-column tile writes the partial (sum, sum of squares) of its
    //     rows' FINAL values to lnp_out[(row * ceil(N/32) + tile) * 2 + {0,1}];
    //   consumer (lnp_in != null): A holds the RAW rows of the producer (width ln_C), W = W' diag(gamma), and the
    //     epilogue applies  y = rstd_r (acc - mean_r s_n) + c_n  with s_n = sum_k W[n][k], c_n = W' beta + bias
    //     (mean_r, rstd_r from the ln_parts partials of row r; `bias` must be null) -- LN(x) W'^T + b, reassociated.
    float *lnp_out;
    const float *lnp_in, *ln_s, *ln_c; int ln_parts, ln_C; float ln_eps;
    // Chained product (latency-regime kernel, 32-column tiles): the work-group of column tile t < chain_tiles multiplies
    // its finished 32 x 32 tile Y (rows x 32 columns) by a second weight table,  C2_t[r][m] = sum_k Y[r][k] W2_t[m][k]
    // (m < n2, K = 32), W2_t = w2 + t * w2_stride (row-major (n2,32)), C2_t = c2 + t * c2_stride -- the attention pool's
    // per-head [w_h | e_h] = q_h T1_h^T straight out of the q columns of the qkv0 GEMM, without a launch of its own.
    const float *w2; float *c2; int n2, ldc2, chain_tiles; long w2_stride, c2_stride;
};
constexpr int kMaxGroups = 8;
struct GemmBatch {
    GemmProb p[kMaxGroups]; int n; int rotate;
    // join folded into this launch (api.hip, "gates"): work-group (0,0,0) ends with gate_wait(tail_gate), so the LAUNCH completes
    // -- and the next kernel of its stream starts -- only when the other stream has signalled; flag == null: no wait
    GateRef tail_gate;
};
int launch_gemm(const GemmBatch &gb, hipStream_t st, int compute_dtype = 0);     // 1: plain bf16 operands where supported
int gemm_policy(int min_tiles_128);      // ptx_gemm_policy

struct LnProb { const float *x; float *y; const float *w; const float *b; const float *add; int R, add_rows; };
struct LnBatch { LnProb p[2]; int n; int C; float eps; };
int launch_ln_rows(const LnBatch &lb, hipStream_t st);

struct HeadProb {
    const float *x; const float *nw; const float *nb;   // rows (R,C), LayerNorm weight/bias
    const float *hw; const float *hb;                   // Linear (nout,C), (nout)
    const float *ab;                                    // (2,nout) BatchNorm1d alpha, beta
    float *out; float *guide; int R, nout;
};
struct HeadBatch { HeadProb p[2]; int n; int C; float eps; };
int launch_heads(const HeadBatch &hb, hipStream_t st);

// ---- proxy attention (attn.hip) -----------------------------------------------------
struct AttnProb {
    const float *Q, *K, *V; float *O; const uint8_t *mask;   // mask (B,nk), 1 = valid, or null
    int nq, nk, ldq, ldk, ldv, ldo;
    long sQ, sK, sV, sO;                                     // per-scene strides (elements)
};
struct AttnBatch { AttnProb p[2]; int n; int B, heads, hd; float scale; };
int launch_attn32(const AttnBatch &ab, hipStream_t st);

// fused ProxyAttention of one (scene, head, branch) per work-group (fattn.hip): qkv (B*n, 3C) rows [q | k | v], pt (B*Lp, C)
// projected proxies, mask (B,Lp) uint8 (1 = valid) or null, out (B*n, C)
struct FAttnProb { const float *qkv, *pt; const uint8_t *mask; float *out; int Lp; };
struct FAttnBatch {
    FAttnProb p[2]; int nb, B, heads, hd, n, C; float scale; int compute_dtype;
    // proxy split (calls with few (scene, head) pairs): `split` work-groups share a (scene, head, branch), each takes a
    // slice of the proxies through both stages; partial results (un-normalised output, running maximum, sum per token) go
    // through `part`, the last one to arrive (ticket) merges.  tickets: (nb, B*heads) int32, zero on entry, left zero.
    int split; float *part; int *tickets;
};
int fattn_split_for(int B, int heads);                      // a pure function of the shape: the workspace is sized with it
constexpr int kFaMaxSplit = 4, kFaPartRow = 34;             // floats per (split, token): 32 outputs + maximum + sum
bool fused_attn_supported(const FAttnBatch &ab);
int launch_proxy_attn(const FAttnBatch &ab, hipStream_t st);

// ---- fused Mlp of a ProxyBlock (mlp.hip) -----------------------------------------------------------------------
struct MlpProb {
    const float *x1, *lnp;          // rows (R,256) and their LayerNorm partials (R,8,2) from the proj GEMM
    const void *w1p, *w2p;          // weight planes (k_prep_planes) of fc1 (with norm2's gamma folded in) and fc2
    const float *fc1_s, *fc1_c, *b2;
    float *x2; int R;
    // the block's output head, applied by the work-group that finishes a row tile (k_heads' arithmetic; head_out = NULL: not)
    const float *nw, *nb, *hw, *hb, *ab; float *head_out, *guide; int nout;
};
struct MlpBatch { MlpProb p[2]; int n; float ln_eps; float *part; int *tickets; int compute_dtype; };
bool mlp_fused_supported(int C, int hidden, int R, int compute_dtype);
size_t mlp_part_bytes(int R);
size_t mlp_ticket_bytes(int R);
int launch_mlp(const MlpBatch &mb, hipStream_t st);
int launch_prep_planes(const float *W, int rows, int K, void *out, hipStream_t st);

// Per-scene base pointers of the point clouds, passed by value as a kernel argument: the caller's
// list of (N,3) tensors is used in place (the reference stacks them into a copy, PRE:426-427; the
// path never writes to its input, so no copy is needed).
int make_scene_pts(const float *stacked, const float *const *list, int B, int N, ScenePts *out);

// ---- clustering / apply (cluster.hip) ---------------------------------------------------
int launch_minmax(const ScenePts &points, int B, int N, uint32_t *mm_enc, hipStream_t st);
int launch_ball_query(const float *centers, const uint32_t *mm_enc, const float *lin, int gs,
                      float margin, float *minmax_out, float *centers_out, const ScenePts &points,
                      int B, int M, int N, int K, float radius, int32_t *idx, float *cluster,
                      int32_t *pad_count, hipStream_t st);
int launch_offset_net(const float *ab, const PtxSlotMlp &mlp, const float *map_w,
                      const float *centers_in, const float *cluster, const float *minmax,
                      int BM, int M, int K, float margin, float *centers_out, float *offsets_out,
                      hipStream_t st);
int launch_pointnet(const float *ab, const PtxSlotMlp &mlp, const float *kcenter,
                    const float *kcluster, int BM, int Mk, int K, int width, float *point_proxy,
                    const PtxBlock *blk_t, const PtxBlock *blk_i, const float *posb_t,
                    const float *posb_i, float *xin_t, float *xin_i, float ln_eps,
                    const int32_t *ksrc, int Msrc, hipStream_t st, uint32_t *head_flag = nullptr, uint32_t head_seq = 0,
                    bool center_src = false);
int launch_order(const PtxShape &s, const int32_t *pad_count, const int32_t *order_override, int32_t *order, hipStream_t st);
// kept rows out of the early tables: point_proxy[row] = pp_all[src], qkv[i][row] = g[i][src] + tb[i][j]  (src = b * M + ksrc[row]; M rows per scene)
int launch_qkv_gather(const float *pp_all, const float *const g[2], const float *const tb[2], const int32_t *ksrc, int B, int M, int Mk,
                      int C, float *point_proxy, float *const qkv[2], hipStream_t st);
int launch_cluster(const PtxShape &s, const uint32_t *mm_enc, const float *lin, const ScenePts &points,
                   const float *off_ab, const PtxSlotMlp &mlp, const float *map_w, const float *centers_override,
                   float *minmax_out, float *centers0, float *cluster1, float *offsets, float *centers,
                   int32_t *idx2, float *cluster2, int32_t *pad_count, hipStream_t st, hipEvent_t done = nullptr);
int launch_select(const PtxShape &s, const int32_t *idx, const float *centers, const float *cluster,
                  const int32_t *pad_count, const int32_t *order_override, int32_t *order,
                  int32_t *picks, int32_t *keep, float *kcenter, float *kcluster, int32_t *kidx,
                  int32_t *drop_idx, uint32_t *tag, hipStream_t st);
int launch_select_order(const PtxShape &s, const float *centers, const int32_t *pad_count,
                        const int32_t *order_override, int32_t *order, int32_t *picks, int32_t *keep,
                        float *kcenter, int32_t *ksrc, uint32_t *mm_clear, hipStream_t st, bool critical,
                        hipEvent_t done, const GateRef *tail = nullptr);
int launch_tags(const PtxShape &s, const int32_t *idx, const int32_t *order, const int32_t *picks, const int32_t *ksrc,
                uint32_t *tag, int32_t *tile_counts, int32_t *counts, int32_t *scene_acc, hipStream_t st,
                const int32_t *keep = nullptr);
int launch_select_slots(const PtxShape &s, const int32_t *idx, const float *cluster, const int32_t *order,
                        const int32_t *picks, const int32_t *keep, float *kcluster, int32_t *kidx,
                        int32_t *drop_idx, uint32_t *tag, hipStream_t st);
// train_fused.hip: LayerNorm rows of the training path for other translation units (the folded attention pool's c_proj + norm_img tail)
int launch_t_ln_fwd(const float *x, const float *w, const float *b, int R, int C, float eps, float *y, float *stats, hipStream_t st);
int t_ln_chunks(int R);                                  // row chunks of launch_t_ln_bwd's column partials
// dx = LayerNorm backward of dy; part[(chunk * 2 + k) * C + c]: k = 0 the chunk's sum of dy * xhat (dgamma), k = 1 of dy (dbeta)
int launch_t_ln_bwd(const float *x, const float *stats, const float *w, const float *dy, int R, int C, float *dx, float *part, hipStream_t st);
int launch_tile_count(const uint32_t *tag, int B, int N, int32_t *tile_counts, int32_t *counts, int32_t *scene_acc,
                      hipStream_t st);
int launch_affine(const PtxShape &s, const ScenePts &points, uint32_t *tag, const float *kcenter,
                  const float *translate, const float *transform, float *out, int32_t *counts,
                  const int32_t *tile_counts, bool compact, bool clear_tag, hipStream_t st,
                  const uint32_t *poison = nullptr);

// ---- image proxy (imgproxy.hip) ------------------------------------------------------------
int launch_img_mean(const float *img, int nimg, int in_dim, int hw, float *fm, hipStream_t st,
                    uint32_t *gate = nullptr, uint32_t gate_seq = 0);
int launch_img_scores(const float *img, const float *we, const float *qkv0, int nimg, int in_dim,
                      int hw, int heads, int C, int KT1, int KT2p, float scale, float *gbuf,
                      hipStream_t st);
int launch_img_gather(const float *img, int nimg, int in_dim, int hw, int heads, int KT2p,
                      float *gbuf, hipStream_t st);

int launch_img_mean16(const void *img, int dt, int nimg, int in_dim, int hw, float *fm, hipStream_t st,
                      uint32_t *gate = nullptr, uint32_t gate_seq = 0);
int launch_img_scores16(const void *img, int dt, const float *we, const float *qkv0, int nimg, int in_dim,
                        int hw, int heads, int C, int KT1, int KT2p, float scale, float *gbuf, hipStream_t st);
int launch_img_gather16(const void *img, int dt, int nimg, int in_dim, int hw, int heads, int KT2p, float *gbuf,
                        hipStream_t st);
// imgpool.hip: single-pass attention pooling of bf16 / fp16 features (scores + softmax numerators + weighted sums)
bool img_pool_supported(int dt, int in_dim, int hw, int heads);
size_t img_pool_bytes(int nimg, int in_dim, int EW);
void img_pool_layout(float *scratch, int nimg, int in_dim, int EW, float **Gs, float **E, float **ML);
int launch_img_pool(const void *img, int dt, const float *we, const float *qkv0, int nimg, int in_dim, int hw, int C,
                    int KT1, int EW, float scale, float *Gs, float *E, float *ML, hipStream_t st);

// imgpool32.hip (r05): the same single pass for fp32 features -- a unit = (image, 128 pixels) = a PAIR of the 16-bit kernel's units,
// results in the same (Gs, E, ML) layout
bool img_pool32_supported(int dt, int in_dim, int hw, int heads);
int launch_img_pool32(const float *img, const float *we, const float *qkv0, int nimg, int in_dim, int hw, int C, int KT1, int EW,
                      float scale, float *Gs, float *E, float *ML, hipStream_t st);

// ---- prep (prep.hip) ----------------------------------------------------------------------------
int run_prepare(const PtxShape &s, const PtxWeights &w, float *prep, hipStream_t st);

}  // namespace ptx
