// Train-mode operator set of the preshape path: the differentiable float half of
// ProxyTransformationNormReverse.forward (PRE:424-469) in train mode -- batch-statistics BatchNorm (PRE:74, 114,
// 329-330), Dropout (PRE:189-191, timm Mlp) and DropPath (PRE:268) -- as forward / backward kernel pairs behind the
// C ABI (include/proxyt.h, "train-mode operators").  The host side (proxytransformation_amd/train.py) chains them with
// torch.autograd.Function nodes: torch keeps the graph and owns the buffers, every arithmetic step is one of the kernels
// below.  The index half (ball query, FPS, selection, tags) is shared with the eval path and is not differentiable,
// exactly as in the reference (pytorch3d returns integer indices).
//
// These kernels favour exactness and generality over speed (fp32 everywhere, fixed summation orders so that two runs
// give the same bits); the eval path's tuned kernels are untouched.
#include <cstdlib>

#include "common.h"

namespace ptx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------ strided batched GEMM
// C[z][m][n] (+)= alpha * sum_k A[z][m][k] B[z][k][n], every operand with arbitrary element strides, so that NN / NT /
// TN products, head-split attention operands and channels-first image features are all plain calls.  z = z1 * inner + z2
// with separate strides for the two batch digits (scene, head).  v_mfma_f32_32x32x2_f32, 64x64 tile per work-group.
struct BGemmArgs {
    const void *A; const void *B; float *C;
    int M, N, K;
    long a_rs, a_cs, b_rs, b_cs, c_rs, c_cs;
    int batch, inner;
    long a_s1, a_s2, b_s1, b_s2, c_s1, c_s2;
    int a_dtype, b_dtype;   // 0 fp32, 1 bf16, 2 fp16 (image features are consumed in their storage type)
    float alpha; int accumulate;
    int ksplit; long c_sk;  // K cut into ksplit slices, slice s writes its partial product at C + s * c_sk (the caller
                            // sums the slices with ptx_op_colsum: fixed order, and the chip is busy when M x N is small)
};

template <int DT>
__device__ __forceinline__ float load_t(const void *base, long off)
{
    if (DT == 0) return static_cast<const float *>(base)[off];
    const unsigned short u = static_cast<const unsigned short *>(base)[off];
    if (DT == 1) return __uint_as_float((unsigned int)u << 16);
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
__device__ __forceinline__ float load_a(const void *base, long off, int dt)
{
    if (dt == 0) return static_cast<const float *>(base)[off];
    const unsigned short u = static_cast<const unsigned short *>(base)[off];
    if (dt == 1) return __uint_as_float((unsigned int)u << 16);
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}

constexpr int TBK = 32;
// ADT / BDT: storage types of the operands, compile-time: a run-time type test inside the fetch makes every load its own
// basic block with its own s_waitcnt vmcnt(0) (r03: that, not the arithmetic, was the 3.6 us per K step of this kernel)
template <int ADT, int BDT>
__global__ __launch_bounds__(256) void k_bgemm(BGemmArgs g)
{
    __shared__ __attribute__((aligned(16))) float As[64][TBK + 4];     // 144-B rows: b128 fragment reads, conflict-free
    __shared__ __attribute__((aligned(16))) float Bs[64][TBK + 4];     // B tile stored n-major: Bs[n][k]
    const int zz = blockIdx.z, z = zz / g.ksplit, ks = zz - z * g.ksplit, z1 = z / g.inner, z2 = z - z1 * g.inner;
    const long ao = z1 * g.a_s1 + z2 * g.a_s2, bo = z1 * g.b_s1 + z2 * g.b_s2, co = z1 * g.c_s1 + z2 * g.c_s2 + ks * g.c_sk;
    const int kper = ((g.K + g.ksplit - 1) / g.ksplit + TBK - 1) / TBK * TBK;
    const int kbeg = ks * kper, kend = min(g.K, kbeg + kper);
    const int row0 = blockIdx.x * 64, col0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    // fp32 MFMA accumulation in chunks of 256 k, the chunks summed in double: weight gradients contract over every slot /
    // token of the batch (K up to a few 10^5), where a single fp32 accumulator would cost three digits
    f32x16 acc;
    double dacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.0f; dacc[i] = 0.0; }
    // staging maps: pick the thread -> element map whose fastest index follows the unit stride of the operand
    const bool a_k_fast = g.a_cs == 1 || g.a_rs != 1;       // A: contiguous along k (or neither)
    const bool b_n_fast = g.b_cs == 1 || g.b_rs != 1;       // B: contiguous along n
    constexpr int E = 64 * TBK / 256;                       // elements of each tile per thread
    int am[E], ak[E], bk[E], bn[E];
    long arow[E], bcol[E];                                  // element offsets of (row, k = 0) / (k = 0, column), rows / columns clamped
    bool aok[E], bok[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = tid + 256 * e;
        if (a_k_fast) { am[e] = idx / TBK; ak[e] = idx % TBK; } else { ak[e] = idx / 64; am[e] = idx % 64; }
        if (b_n_fast) { bk[e] = idx / 64; bn[e] = idx % 64; } else { bn[e] = idx / TBK; bk[e] = idx % TBK; }
        aok[e] = row0 + am[e] < g.M; bok[e] = col0 + bn[e] < g.N;
        arow[e] = ao + (long)min(row0 + am[e], g.M - 1) * g.a_rs;
        bcol[e] = bo + (long)min(col0 + bn[e], g.N - 1) * g.b_cs;
    }
    // Two tiles are in flight in registers while a third is multiplied out of LDS: every product of the training path is
    // latency-bound here (a few dozen work-groups walking K = 700 .. 300 000 with one dependent round trip per step --
    // r02: 61 launches, 4.7 of the step's 11 ms).  The fetches are BRANCH-FREE (clamped addresses, out-of-range elements
    // zeroed after the load): with predicated loads hipcc cannot count the outstanding ones and drains them all
    // (s_waitcnt vmcnt(0)) before every stash -- the prefetch then hides nothing (3.6 us per K step)
    float ra0[E], rb0[E], ra1[E], rb1[E];
    auto fetch = [&](int k0, float (&ra)[E], float (&rb)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const long ai = arow[e] + (long)min(k0 + ak[e], kend - 1) * g.a_cs;
            const long bi = bcol[e] + (long)min(k0 + bk[e], kend - 1) * g.b_rs;
            // raw values only: the out-of-range elements are zeroed when the tile is stashed, two steps later -- a select
            // right here would make the wave wait for each load as it is issued
            ra[e] = load_t<ADT>(g.A, ai); rb[e] = load_t<BDT>(g.B, bi);
        }
    };
    auto step = [&](int k0, float (&ra)[E], float (&rb)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            As[am[e]][ak[e]] = (aok[e] && k0 + ak[e] < kend) ? ra[e] : 0.0f;
            Bs[bn[e]][bk[e]] = (bok[e] && k0 + bk[e] < kend) ? rb[e] : 0.0f;
        }
        __syncthreads();
        fetch(k0 + 2 * TBK, ra, rb);    // this set is free again: two steps ahead.  Unconditional (beyond the end the clamped
                                        // addresses re-read the last column): a branch here hides the number of loads in flight
#pragma unroll
        for (int kk = 0; kk < TBK / 8; ++kk) {          // lanes hh = 0 / 1 contract k = 8 kk + j and 8 kk + 4 + j
            const float4 a4 = *reinterpret_cast<const float4 *>(&As[wr * 32 + li][kk * 8 + hh * 4]);
            const float4 b4 = *reinterpret_cast<const float4 *>(&Bs[wc * 32 + li][kk * 8 + hh * 4]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        __syncthreads();
        if ((((k0 - kbeg) / TBK) & 7) == 7) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { dacc[i] += (double)acc[i]; acc[i] = 0.0f; }
        }
    };
    if (kbeg < kend) {
        fetch(kbeg, ra0, rb0);
        fetch(kbeg + TBK, ra1, rb1);
        for (int k0 = kbeg; k0 < kend; k0 += 2 * TBK) {     // an odd number of steps runs one step on zeros
            step(k0, ra0, rb0);
            step(k0 + TBK, ra1, rb1);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)(dacc[i] + (double)acc[i]);
    const int n = col0 + wc * 32 + li;
    if (n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (m < g.M) {
            float *dst = g.C + co + m * g.c_rs + n * g.c_cs;
            const float v = g.alpha * acc[r];
            *dst = g.accumulate ? *dst + v : v;
        }
    }
}

// ------------------------------------------------------------------------------ thin batched products
// The training path's query-0 products of AttentionPool2d (one query row per (image, head): (1 x 32)(32 x 226), (1 x 226)
// (226 x 32) and the outer products of their backward, 960 batches each) cost 38-56 us apiece as 64 x 64 MFMA tiles with one
// valid row (r03: 280 us of a 5.4 ms step).  fp32 operands, no K slices.
// k_bthin_out: a thread per output element, K <= 64 walked eight requests at a time (branch-free, clamped).
template <bool KVEC>     // KVEC: both operands contiguous along k, K % 4 == 0, 16-byte aligned: four k per request
__global__ __launch_bounds__(256) void k_bthin_out(BGemmArgs g, unsigned total)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    // fastest index along C's unit stride (32-bit index arithmetic: a 64-bit division per thread is most of a K = 1 product)
    const bool n_fast = g.c_cs == 1 || g.c_rs != 1;
    const unsigned per = (unsigned)g.M * (unsigned)g.N, z = i / per, r = i - z * per;
    const unsigned m = n_fast ? r / (unsigned)g.N : r % (unsigned)g.M, n = n_fast ? r % (unsigned)g.N : r / (unsigned)g.M;
    const unsigned z1 = z / (unsigned)g.inner, z2 = z - z1 * (unsigned)g.inner;
    const float *A = static_cast<const float *>(g.A) + z1 * g.a_s1 + z2 * g.a_s2 + (long)m * g.a_rs;
    const float *B = static_cast<const float *>(g.B) + z1 * g.b_s1 + z2 * g.b_s2 + (long)n * g.b_cs;
    float acc = 0.0f;
    if (KVEC) {
        for (int k0 = 0; k0 < g.K; k0 += 16) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(k0 + 4 * u, g.K - 4);
                a[u] = *reinterpret_cast<const float4 *>(A + k); b[u] = *reinterpret_cast<const float4 *>(B + k);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float w = k0 + 4 * u < g.K ? 1.0f : 0.0f;
                acc = fmaf(a[u].x * w, b[u].x, acc); acc = fmaf(a[u].y * w, b[u].y, acc);
                acc = fmaf(a[u].z * w, b[u].z, acc); acc = fmaf(a[u].w * w, b[u].w, acc);
            }
        }
    } else {
        for (int k0 = 0; k0 < g.K; k0 += 8) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(k0 + u, g.K - 1);
                a[u] = A[(long)k * g.a_cs]; b[u] = B[(long)k * g.b_rs];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(k0 + u < g.K ? a[u] : 0.0f, b[u], acc);
        }
    }
    float *dst = g.C + z1 * g.c_s1 + z2 * g.c_s2 + (long)m * g.c_rs + (long)n * g.c_cs;
    const float v = g.alpha * acc;
    *dst = g.accumulate ? *dst + v : v;
}
// k_bthin_row: a wave per (batch, row m), N <= 32 with B contiguous along n, K <= 256 dealt to the lanes four at a time: every
// lane requests its rows of B up front (eight 16-B loads per row; N = 32 exactly), the 32 partial sums are reduced across the wave.
__global__ __launch_bounds__(256) void k_bthin_row(BGemmArgs g, long rows)
{
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows) return;
    const int lane = threadIdx.x & 63;
    const long z = w / g.M;
    const int m = (int)(w - z * g.M);
    const long z1 = z / g.inner, z2 = z - z1 * g.inner;
    const float *A = static_cast<const float *>(g.A) + z1 * g.a_s1 + z2 * g.a_s2 + (long)m * g.a_rs;
    const float *B = static_cast<const float *>(g.B) + z1 * g.b_s1 + z2 * g.b_s2;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = lane + 64 * u;                         // K <= 256
        const int kc = min(k, g.K - 1);
        const float a = k < g.K ? A[(long)kc * g.a_cs] : 0.0f;
        const float4 *row = reinterpret_cast<const float4 *>(B + (long)kc * g.b_rs);       // N = 32, 16-byte aligned rows
        float4 bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) bv[q] = row[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc[4 * q] = fmaf(a, bv[q].x, acc[4 * q]); acc[4 * q + 1] = fmaf(a, bv[q].y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(a, bv[q].z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(a, bv[q].w, acc[4 * q + 3]);
        }
    }
    float mine = 0.0f;
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        const float s = wave_sum(acc[n]);
        if (lane == n) mine = s;
    }
    if (lane < g.N) {
        float *dst = g.C + z1 * g.c_s1 + z2 * g.c_s2 + (long)m * g.c_rs + (long)lane * g.c_cs;
        const float v = g.alpha * mine;
        *dst = g.accumulate ? *dst + v : v;
    }
}

// ------------------------------------------------------------------------------ column reductions
// out[n] (+)= sum_r f(x[r][n]) over R rows of a dense (R,N) matrix; fixed order: a work-group owns 64 columns and walks
// the rows in 4 interleaved slices that are added in slice order, accumulating in double (the sums run over every
// slot / token of the batch).  mode 0: x, 1: x * y, 2: x * x, 3: (x - y[col])^2 with y a per-column vector (variance
// around the mean: robust where E[x^2] - E[x]^2 cancels)
__device__ __forceinline__ double colsum_term(const float *__restrict__ x, const float *__restrict__ y, size_t r, int N, int c,
                                               int mode, float yc)
{
    const float a = x[r * N + c];
    if (mode == 0) return (double)a;
    if (mode == 1) return (double)a * (double)y[r * N + c];
    if (mode == 2) return (double)a * (double)a;
    const double d = (double)a - (double)yc;
    return d * d;
}
// stage 1: block (column tile, row split) -> part[split][c]; rows are dealt to the 4 * nsplit slices round-robin.
// V = 4: a thread owns four adjacent columns (16-B loads: the big reductions run over 10^5 rows of 256 columns and are
// pure streaming); V = 1: any N / alignment.
template <int V>
__global__ __launch_bounds__(256) void k_colsum(const float *__restrict__ x, const float *__restrict__ y, int R, int N,
                                                int mode, int nsplit, double *__restrict__ part)
{
    __shared__ double red[4][64 * V];
    const int c0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * V, sl = threadIdx.x >> 6, sp = blockIdx.y;
    double s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0.0;
    if (c0 < N) {
        float yc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) yc[v] = (mode == 3 && c0 + v < N) ? y[c0 + v] : 0.0f;
        const int step = 4 * nsplit;
        auto term = [&](size_t r, double (&t)[V]) {
            float a[V], b[V];
            if (V == 4) {
                const float4 q = *reinterpret_cast<const float4 *>(x + r * N + c0);
                a[0] = q.x; a[1 % V] = q.y; a[2 % V] = q.z; a[3 % V] = q.w;
                if (mode == 1) { const float4 w = *reinterpret_cast<const float4 *>(y + r * N + c0); b[0] = w.x; b[1 % V] = w.y; b[2 % V] = w.z; b[3 % V] = w.w; }
            } else {
                a[0] = x[r * N + c0];
                if (mode == 1) b[0] = y[r * N + c0];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (mode == 0) t[v] = (double)a[v];
                else if (mode == 1) t[v] = (double)a[v] * (double)b[v];
                else if (mode == 2) t[v] = (double)a[v] * (double)a[v];
                else { const double d = (double)a[v] - (double)yc[v]; t[v] = d * d; }
            }
        };
        int r = sp * 4 + sl;
        for (; r + 3 * step < R; r += 4 * step) {           // four rows in flight; summed in row order
            double t0[V], t1[V], t2[V], t3[V];
            term((size_t)r, t0); term((size_t)r + step, t1); term((size_t)r + 2 * step, t2); term((size_t)r + 3 * step, t3);
#pragma unroll
            for (int v = 0; v < V; ++v) { s[v] += t0[v]; s[v] += t1[v]; s[v] += t2[v]; s[v] += t3[v]; }
        }
        for (; r < R; r += step) {
            double t0[V];
            term((size_t)r, t0);
#pragma unroll
            for (int v = 0; v < V; ++v) s[v] += t0[v];
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) red[sl][(threadIdx.x & 63) * V + v] = s[v];
    __syncthreads();
    if (sl == 0 && c0 < N) {
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (c0 + v < N) {
                const int i = (threadIdx.x & 63) * V + v;
                part[(size_t)sp * N + c0 + v] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
            }
    }
}
// stage 2: the splits, dealt to 16 lanes per column round-robin and combined in lane order (fixed order)
// (r03: stage 2 inside stage 1 by the last split of a column tile to arrive -- agent-scope partials around a ticket, the pattern
//  of fattn.hip / mlp.hip -- saves 54 launches per training step and made the step SLOWER, 5.4 vs 4.9 ms: up to 129 splits take
//  their ticket from the same word, one after the other; with fewer splits the big reductions over 600 k slot rows starve)
__global__ __launch_bounds__(256) void k_colsum_fin(const double *__restrict__ part, int N, int nsplit, float scale, int accumulate,
                                                    float *__restrict__ out)
{
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    double t = 0.0;
    if (c < N)
        for (int sp = q; sp < nsplit; sp += 16) t += part[(size_t)sp * N + c];
    red[q][cl] = t;
    __syncthreads();
    if (q == 0 && c < N) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i][cl];
        tot *= (double)scale;
        out[c] = accumulate ? (float)((double)out[c] + tot) : (float)tot;
    }
}

// ------------------------------------------------------------------------------ element-wise family
// op 0: y = a + b              1: y = a * s          2: y = gelu_erf(a)         3: y = b * gelu'(a)  (a = pre-activation, b = dy)
//    4: y = relu(a)            5: y = b * (a > 0)    6: y = a + bias[col]       7: y = a + s * b       8: y = a * b
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_g(float x)
{
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
__global__ void k_eltwise(int op, const float *__restrict__ a, const float *__restrict__ b, float s, long n, int ncol,
                          float *__restrict__ y)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float av = a[i];
        float v;
        switch (op) {
            case 0: v = av + b[i]; break;
            case 1: v = av * s; break;
            case 2: v = gelu_f(av); break;
            case 3: v = b[i] * gelu_g(av); break;
            case 4: v = fmaxf(av, 0.0f); break;
            case 5: v = av > 0.0f ? b[i] : 0.0f; break;
            case 6: v = av + b[i % ncol]; break;
            case 7: v = fmaf(s, b[i], av); break;
            default: v = av * b[i]; break;
        }
        y[i] = v;
    }
}

// Dropout / DropPath with a counter-based generator: element i of stream `seed` is kept iff hash(seed, i >> shift) maps
// above p; kept values are scaled by 1 / (1 - p).  The mask is a pure function of (seed, index): the backward pass
// recomputes it (same kernel, dy in place of x).  shift groups elements that share one decision (DropPath: one per
// sample, PRE:268 -> group = elements per sample; a non-power-of-two group is passed as `group`).
__device__ __forceinline__ uint32_t mix32(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((x ^ (x >> 31)) >> 32);
}
__global__ void k_dropout(const float *__restrict__ x, long n, long group, float p, uint64_t seed, float *__restrict__ y)
{
    const float keep_scale = 1.0f / (1.0f - p);
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint32_t r = mix32(seed * 0x100000001B3ull + (uint64_t)(i / group));
        y[i] = r >= thresh ? x[i] * keep_scale : 0.0f;
    }
}

// ------------------------------------------------------------------------------ LayerNorm (rows of C <= 512, C % 64 == 0)
constexpr int kLnMax = 8;
__global__ __launch_bounds__(256) void k_ln_fwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                const float *__restrict__ add, int add_rows, int R, int C, float eps,
                                                float *__restrict__ y, float *__restrict__ stats)
{
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= R) return;
    const int lane = lane_id();
    float v[kLnMax], s = 0.0f;
#pragma unroll
    for (int q = 0; q < kLnMax; ++q) { const int c = lane + 64 * q; v[q] = c < C ? x[(size_t)row * C + c] : 0.0f; s += v[q]; }
    const float mean = wave_sum(s) / (float)C;
    float var = 0.0f;
#pragma unroll
    for (int q = 0; q < kLnMax; ++q) { const int c = lane + 64 * q; const float d = c < C ? v[q] - mean : 0.0f; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)C + eps);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
    for (int q = 0; q < kLnMax; ++q) {
        const int c = lane + 64 * q;
        if (c < C) {
            float o = (v[q] - mean) * rstd * w[c] + b[c];
            if (add) o += add[(size_t)(row % add_rows) * C + c];        // per-slot bias table (PRE:215-217)
            y[(size_t)row * C + c] = o;
        }
    }
}
// dx = rstd (g - mean_c(g) - xhat mean_c(g xhat)), g = dy w;  xhat_out = xhat (for dgamma = colsum(dy * xhat))
__global__ __launch_bounds__(256) void k_ln_bwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ dy,
                                                const float *__restrict__ stats, int R, int C, float *__restrict__ dx,
                                                float *__restrict__ xhat_out)
{
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= R) return;
    const int lane = lane_id();
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[kLnMax], g[kLnMax], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int q = 0; q < kLnMax; ++q) {
        const int c = lane + 64 * q;
        if (c < C) {
            xh[q] = (x[(size_t)row * C + c] - mean) * rstd;
            g[q] = dy[(size_t)row * C + c] * w[c];
        } else { xh[q] = 0.0f; g[q] = 0.0f; }
        s1 += g[q]; s2 = fmaf(g[q], xh[q], s2);
    }
    s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int q = 0; q < kLnMax; ++q) {
        const int c = lane + 64 * q;
        if (c < C) {
            dx[(size_t)row * C + c] = rstd * (g[q] - s1 - xh[q] * s2);
            xhat_out[(size_t)row * C + c] = xh[q];
        }
    }
}

// ------------------------------------------------------------------------------ BatchNorm over the rows of (R,C), train mode
// apply: y = relu?( (x - mean) * rstd * w + b ); mean = colsum(x) / R, then the sum of squares AROUND that mean
// (k_colsum mode 3) through k_bn_stats
__global__ void k_bn_stats(const float *__restrict__ mean_in, const float *__restrict__ sumsq, int C, long R, float eps, float momentum,
                           float *__restrict__ mean_rstd, float *__restrict__ run_mean, float *__restrict__ run_var)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = mean_in[c];
    const float var = sumsq[c] / (float)R;                                     // biased: what normalises (PRE:74)
    mean_rstd[c] = mean;
    mean_rstd[C + c] = 1.0f / sqrtf(var + eps);
    if (run_mean) {     // running statistics as nn.BatchNorm updates them: unbiased variance, momentum 0.1
        const float unb = R > 1 ? var * (float)R / (float)(R - 1) : var;
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
    }
}
__global__ void k_bn_apply(const float *__restrict__ x, const float *__restrict__ mean_rstd, const float *__restrict__ w,
                           const float *__restrict__ b, long n, int C, int relu, float *__restrict__ y)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = (x[i] - mean_rstd[c]) * mean_rstd[C + c] * w[c] + b[c];
        y[i] = relu ? fmaxf(v, 0.0f) : v;
    }
}
// backward, two launches around two column reductions:
//   1. g = dy * (y > 0 if relu);  xhat = (x - mean) rstd;  write g and g * xhat          (k_bn_bwd_prep)
//   2. dbeta = colsum(g), dgamma = colsum(g xhat)                                         (k_colsum)
//   3. dx = w rstd (g - dbeta / R - xhat dgamma / R)                                      (k_bn_bwd_dx)
__global__ void k_bn_bwd_prep(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                              const float *__restrict__ mean_rstd, long n, int C, int relu, float *__restrict__ g,
                              float *__restrict__ gx)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float gv = (relu && !(y[i] > 0.0f)) ? 0.0f : dy[i];
        g[i] = gv;
        gx[i] = gv * ((x[i] - mean_rstd[c]) * mean_rstd[C + c]);
    }
}
__global__ void k_bn_bwd_dx(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ mean_rstd,
                            const float *__restrict__ w, const float *__restrict__ dbeta, const float *__restrict__ dgamma,
                            long n, int C, long R, float *__restrict__ dx)
{
    const float invR = 1.0f / (float)R;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float xh = (x[i] - mean_rstd[c]) * mean_rstd[C + c];
        dx[i] = w[c] * mean_rstd[C + c] * (g[i] - dbeta[c] * invR - xh * dgamma[c] * invR);
    }
}

// ------------------------------------------------------------------------------ softmax over the last dim of (rows, L)
// mask (optional): (B, L) uint8, 1 = valid; row r belongs to scene r / rows_per_scene; masked scores are FILLED with
// -1e9 (PRE:247), not removed.  One wave per row.
__global__ __launch_bounds__(256) void k_softmax_fwd(const float *__restrict__ s, const uint8_t *__restrict__ mask,
                                                     long rows, int L, long rows_per_scene, float *__restrict__ p)
{
    const long row = blockIdx.x * 4l + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = lane_id();
    const uint8_t *mk = mask ? mask + (row / rows_per_scene) * L : nullptr;
    const float *sr = s + row * L;
    float mx = -INFINITY;
    for (int i = lane; i < L; i += 64) { const float v = (mk && mk[i] == 0) ? -1e9f : sr[i]; mx = fmaxf(mx, v); }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int i = lane; i < L; i += 64) { const float v = (mk && mk[i] == 0) ? -1e9f : sr[i]; sum += expf(v - mx); }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < L; i += 64) { const float v = (mk && mk[i] == 0) ? -1e9f : sr[i]; p[row * L + i] = expf(v - mx) * inv; }
}
// ds = p (dp - sum_l p dp); masked positions receive no gradient (masked_fill)
__global__ __launch_bounds__(256) void k_softmax_bwd(const float *__restrict__ p, const float *__restrict__ dp,
                                                     const uint8_t *__restrict__ mask, long rows, int L, long rows_per_scene,
                                                     float *__restrict__ ds)
{
    const long row = blockIdx.x * 4l + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = lane_id();
    const uint8_t *mk = mask ? mask + (row / rows_per_scene) * L : nullptr;
    float dot = 0.0f;
    for (int i = lane; i < L; i += 64) dot = fmaf(p[row * L + i], dp[row * L + i], dot);
    dot = wave_sum(dot);
    for (int i = lane; i < L; i += 64) {
        const float v = p[row * L + i] * (dp[row * L + i] - dot);
        ds[row * L + i] = (mk && mk[i] == 0) ? 0.0f : v;
    }
}

// ------------------------------------------------------------------------------ slot-network pieces (PRE:87-107, 126-142)
// x6[r] = [rel (0 on padded slots), p], r = (cluster, k); centre row = cluster number, the slots are read through `src`
// (row of an un-gathered (.,K,3) array) when given.  padmask[r] = 1 on padded slots.
__global__ void k_slot_inputs(const float *__restrict__ center, const float *__restrict__ cluster, const int32_t *__restrict__ src,
                              long nclus, int K, float *__restrict__ x6, uint8_t *__restrict__ padmask)
{
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < nclus * K; r += (long)gridDim.x * blockDim.x) {
        const long cl = r / K; const int k = (int)(r - cl * K);
        const long s = src ? src[cl] : cl;
        const float *pk = cluster + (s * K + k) * 3, *c = center + cl * 3;
        const float px = pk[0], py = pk[1], pz = pk[2];
        const bool pad = px == 0.0f && py == 0.0f && pz == 0.0f;                 // PRE:94 / PRE:132
        float *o = x6 + r * 6;
        o[0] = pad ? 0.0f : px - c[0]; o[1] = pad ? 0.0f : py - c[1]; o[2] = pad ? 0.0f : pz - c[2];
        o[3] = px; o[4] = py; o[5] = pz;
        padmask[r] = pad ? 1 : 0;
    }
}
// dcenter[cl] = - sum over non-padded slots of dx6[r][0..2]  (rel = p - c; the points carry no gradient)
__global__ void k_slot_inputs_bwd(const float *__restrict__ dx6, const uint8_t *__restrict__ padmask, long nclus, int K,
                                  float *__restrict__ dcenter)
{
    for (long cl = blockIdx.x * (long)blockDim.x + threadIdx.x; cl < nclus; cl += (long)gridDim.x * blockDim.x) {
        float a = 0.0f, b = 0.0f, c = 0.0f;
        for (int k = 0; k < K; ++k) {
            if (padmask[cl * K + k]) continue;
            const float *d = dx6 + (cl * K + k) * 6;
            a -= d[0]; b -= d[1]; c -= d[2];
        }
        dcenter[cl * 3] = a; dcenter[cl * 3 + 1] = b; dcenter[cl * 3 + 2] = c;
    }
}
// pooling over the K slots of a cluster, h (nclus, K, C): mode 0 mean (PRE:102), 1 max with first-arg-max (PRE:140)
__global__ void k_slot_pool(const float *__restrict__ h, long nclus, int K, int C, int mode, float *__restrict__ out,
                            int32_t *__restrict__ arg)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nclus * C; i += (long)gridDim.x * blockDim.x) {
        const long cl = i / C; const int c = (int)(i - cl * C);
        const float *hp = h + cl * K * C + c;
        if (mode == 0) {
            float s = 0.0f;
            for (int k = 0; k < K; ++k) s += hp[(long)k * C];
            out[i] = s / (float)K;
        } else {
            float best = hp[0]; int bi = 0;
            for (int k = 1; k < K; ++k) { const float v = hp[(long)k * C]; if (v > best) { best = v; bi = k; } }
            out[i] = best; arg[i] = bi;
        }
    }
}
__global__ void k_slot_pool_bwd(const float *__restrict__ dout, const int32_t *__restrict__ arg, long nclus, int K, int C,
                                int mode, float *__restrict__ dh)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nclus * K * C; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long rk = i / C; const int k = (int)(rk % K); const long cl = rk / K;
        const float g = dout[cl * C + c];
        dh[i] = mode == 0 ? g / (float)K : (arg[cl * C + c] == k ? g : 0.0f);
    }
}
// centres after the offset network: off = tanh(raw) margin; new = clamp(c0 + off, min, max)   (PRE:59-62)
// grad mask gm = 1 where the clamp is inactive (strictly inside the box), 0.5 on exact ties like torch.min / torch.max
__global__ void k_offset_apply(const float *__restrict__ c0, const float *__restrict__ raw, const float *__restrict__ minmax,
                               long n, int M, float margin, float *__restrict__ cout, float *__restrict__ dcoef)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long cl = i / 3; const int d = (int)(i - cl * 3); const long b = cl / M;
        const float t = tanhf(raw[i]);
        const float v = c0[i] + t * margin;
        const float mn = minmax[b * 6 + d], mx = minmax[b * 6 + 3 + d];
        const float lo = fminf(v, mx);
        cout[i] = fmaxf(lo, mn);
        float gm = v < mx ? 1.0f : (v == mx ? 0.5f : 0.0f);
        gm *= lo > mn ? 1.0f : (lo == mn ? 0.5f : 0.0f);
        dcoef[i] = gm * margin * (1.0f - t * t);           // d cout / d raw
    }
}

// ------------------------------------------------------------------------------ per-slot bias table of ProxyAttention (PRE:212-215)
// table[j][y s + x] = bilinear(pb[j] 4x4 -> s x s)[y][x] + pc[j][y] + pr[j][x], first C entries of the grid
__device__ __forceinline__ void bilin_taps(int y, int x, int s, int &y0, int &y1, int &x0, int &x1, float &ly0, float &ly1,
                                           float &lx0, float &lx1)
{
    const float sc = 4.0f / (float)s;
    float sy = sc * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = sc * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    y0 = (int)sy; x0 = (int)sx;
    y1 = y0 + (y0 < 3 ? 1 : 0); x1 = x0 + (x0 < 3 ? 1 : 0);
    ly1 = sy - (float)y0; ly0 = 1.f - ly1; lx1 = sx - (float)x0; lx0 = 1.f - lx1;
}
__global__ void k_slotbias_fwd(const float *pb, const float *pc, const float *pr, int Mk, int s, int C, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mk * C) return;
    const int j = i / C, yx = i - j * C, y = yx / s, x = yx - y * s;
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bilin_taps(y, x, s, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
    const float *p = pb + (size_t)j * 16;
    const float v = ly0 * (lx0 * p[y0 * 4 + x0] + lx1 * p[y0 * 4 + x1]) + ly1 * (lx0 * p[y1 * 4 + x0] + lx1 * p[y1 * 4 + x1]);
    out[i] = v + (pc[(size_t)j * s + y] + pr[(size_t)j * s + x]);
}
// one wave per kept slot j: the C table gradients of the slot are staged in LDS; the 16 taps of pb are dealt to 16 x 4 lanes
// (four interleaved quarters of the grid per tap, combined across the quad in lane order), then lanes 0 .. 2 s - 1 sum the rows
// (pc) and the columns (pr).  (r04: one lane per tap walked all C entries with four compares each -- 32 us per call)
__global__ __launch_bounds__(256) void k_slotbias_bwd(const float *dtab, int Mk, int s, int C, float *dpb, float *dpc, float *dpr)
{
    __shared__ float g[4][512];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wv;
    if (j >= Mk) return;
    for (int i = lane; i < C; i += 64) g[wv][i] = dtab[(size_t)j * C + i];
    // same wave: LDS accesses complete in order
    const float *gj = g[wv];
    {
        const int tap = lane >> 2, part = lane & 3, ty = tap >> 2, tx = tap & 3;
        float acc = 0.0f;
        for (int yx = part; yx < C; yx += 4) {
            const int y = yx / s, x = yx - y * s;
            int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
            bilin_taps(y, x, s, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
            const float wy = (y0 == ty ? ly0 : 0.0f) + (y1 == ty ? ly1 : 0.0f);
            const float wx = (x0 == tx ? lx0 : 0.0f) + (x1 == tx ? lx1 : 0.0f);
            acc = fmaf(gj[yx], wy * wx, acc);
        }
        const float a1 = __shfl_xor(acc, 1);
        acc = (lane & 1) ? a1 + acc : acc + a1;
        const float a2 = __shfl_xor(acc, 2);
        acc = (lane & 2) ? a2 + acc : acc + a2;
        if (part == 0) dpb[(size_t)j * 16 + tap] = acc;
    }
    if (lane < s) {                        // pc[y]: sum over the row
        const int y = lane;
        float acc = 0.0f;
        for (int x = 0; x < s; ++x) { const int yx = y * s + x; if (yx < C) acc += gj[yx]; }
        dpc[(size_t)j * s + y] = acc;
    } else if (lane >= 32 && lane < 32 + s) {     // pr[x]: sum over the column
        const int x = lane - 32;
        float acc = 0.0f;
        for (int y = 0; y * s + x < C; ++y) acc += gj[y * s + x];
        dpr[(size_t)j * s + x] = acc;
    }
}

// ------------------------------------------------------------------------------ row gather / scatter (kept centres)
__global__ void k_rows_gather(const float *__restrict__ x, const int32_t *__restrict__ src, long rows, int C, float *__restrict__ y)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows * C; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        y[i] = x[(long)src[r] * C + (i - r * C)];
    }
}
// dx zero-filled by the caller; src rows are distinct (a cluster is kept at most once)
__global__ void k_rows_scatter(const float *__restrict__ dy, const int32_t *__restrict__ src, long rows, int C, float *__restrict__ dx)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows * C; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        dx[(long)src[r] * C + (i - r * C)] = dy[i];
    }
}
// src[b * Mk + j] = b * M + order[b][keep[b][j]]  (row of the un-gathered (B,M,..) arrays of kept cluster j)
__global__ void k_keep_rows(const int32_t *__restrict__ order, const int32_t *__restrict__ keep, int B, int M, int Mt, int Mk,
                            int32_t *__restrict__ src)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Mk) return;
    const int b = i / Mk;
    src[i] = b * M + order[(size_t)b * Mt + keep[i]];
}

// ------------------------------------------------------------------------------ affine apply, backward (PRE:459-465)
// The reference scatters new = T (p - c) + c + t of EVERY valid slot with index_put_ (PRE:495); autograd hands every such
// slot the gradient of the point it targeted (duplicates included: IndexPutBackward gathers, it does not arbitrate), and
// points removed afterwards (PRE:467) have no gradient.  One wave per kept cluster:
//   g_k = dout[opos[idx_k]] (0 if padded / dropped);  dt = sum g_k;  dT = sum g_k (p_k - c)^T;  dc = sum (g_k - T^T g_k)
struct DoutList { const float *p[32]; };       // per-scene output gradients ((n_b,3) each; null = no gradient); all null: `dout` (B,N,3)
__global__ __launch_bounds__(256) void k_affine_bwd(const float *__restrict__ dout, DoutList dl, const int32_t *__restrict__ opos,
                                                    const int32_t *__restrict__ idx, const float *__restrict__ cluster,
                                                    const float *__restrict__ kcenter,
                                                    const float *__restrict__ transform, int B, int N, int Mk, int K,
                                                    float *__restrict__ dtranslate, float *__restrict__ dtransform,
                                                    float *__restrict__ dkcenter)
{
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= B * Mk) return;
    const int lane = lane_id(), b = w / Mk;
    const long s = w;                           // kidx / kcluster: the gathered (B,Mk,K,..) arrays of the selection
    float g[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    if (lane < K) {
        const int id = idx[s * K + lane];
        if (id >= 0) {
            const int pos = opos[(size_t)b * N + id];
            const float *gb = dout ? dout + (size_t)b * N * 3 : dl.p[b];
            if (pos >= 0 && gb) {
                const float *gp = gb + (size_t)pos * 3;
                g[0] = gp[0]; g[1] = gp[1]; g[2] = gp[2];
            }
        }
        const float *pk = cluster + (s * K + lane) * 3;
        d[0] = pk[0] - kcenter[(size_t)w * 3]; d[1] = pk[1] - kcenter[(size_t)w * 3 + 1]; d[2] = pk[2] - kcenter[(size_t)w * 3 + 2];
    }
    const float *T = transform + (size_t)w * 9;
    float acc[15];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc[i] = g[i];                                                           // dt
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[3 + 3 * i + j] = g[i] * d[j];            // dT[i][j]
        acc[12 + i] = g[i] - (T[i] * g[0] + T[3 + i] * g[1] + T[6 + i] * g[2]);  // dc = g - T^T g
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { dtranslate[(size_t)w * 3 + i] = acc[i]; dkcenter[(size_t)w * 3 + i] = acc[12 + i]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) dtransform[(size_t)w * 9 + i] = acc[3 + i];
    }
}
// forward companion: output position of every input point (-1 = dropped), from the tags and the tile prefix, as k_affine
// compacts them (PRE:516-523 keeps the original order)
__global__ __launch_bounds__(256) void k_out_positions(const uint32_t *__restrict__ tag, const int32_t *__restrict__ tile_counts,
                                                       int N, int32_t *__restrict__ opos)
{
    const int b = blockIdx.y, tile = blockIdx.x, ntiles = gridDim.x;
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    __shared__ int s_cnt[kTilePts / 256][4];
    __shared__ int s_base;
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < tile; ++t) acc += tile_counts[b * ntiles + t];
        s_base = acc;
    }
    bool keep[kTilePts / 256]; unsigned long long bal[kTilePts / 256];
#pragma unroll
    for (int r = 0; r < kTilePts / 256; ++r) {
        const int n = tile * kTilePts + r * 256 + tid;
        keep[r] = n < N && (tag[(size_t)b * N + n] >> 31) == 0;
        bal[r] = __ballot(keep[r]);
        if (lane == 0) s_cnt[r][wid] = __popcll(bal[r]);
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    int run = s_base;
#pragma unroll
    for (int r = 0; r < kTilePts / 256; ++r) {
        const int n = tile * kTilePts + r * 256 + tid;
        int before = 0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) before += ww < wid ? s_cnt[r][ww] : 0;
        if (n < N) opos[(size_t)b * N + n] = keep[r] ? run + before + __popcll(bal[r] & lt) : -1;
        run += s_cnt[r][0] + s_cnt[r][1] + s_cnt[r][2] + s_cnt[r][3];
    }
}

// ------------------------------------------------------------------------------ image tokens (PRE:155-157)
// tok[img][0] = mean_p tok[img][1 + p]  (the conv already added its bias);  then tok[img][t] += pos[t]
__global__ void k_tokens_finish(float *__restrict__ tok, const float *__restrict__ pos, int nimg, int hw, int C)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)nimg * C; i += (long)gridDim.x * blockDim.x) {
        const long im = i / C; const int c = (int)(i - im * C);
        float *t = tok + im * (hw + 1) * C + c;
        float s = 0.0f;
        for (int p = 1; p <= hw; ++p) s += t[(long)p * C];
        t[0] = s / (float)hw + pos[c];
        for (int p = 1; p <= hw; ++p) t[(long)p * C] += pos[(long)p * C + c];
    }
}
// backward: dtok[img][1 + p] += dtok[img][0] / hw, then dtok[img][0] = 0 (in place: what is left is the gradient of the
// pixel tokens BEFORE the positional embedding; the positional / bias sums are column reductions)
__global__ void k_tokens_finish_bwd(float *__restrict__ dtok, int nimg, int hw, int C)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)nimg * C; i += (long)gridDim.x * blockDim.x) {
        const long im = i / C; const int c = (int)(i - im * C);
        float *t = dtok + im * (hw + 1) * C + c;
        const float g0 = t[0] / (float)hw;
        for (int p = 1; p <= hw; ++p) t[(long)p * C] += g0;
        t[0] = 0.0f;
    }
}

static inline int blocks_for(long n, int per = 256) { long b = (n + per - 1) / per; return (int)(b < 1 ? 1 : (b > 65535 * 4 ? 65535 * 4 : b)); }

}  // namespace ptx

using namespace ptx;

extern "C" {

int ptx_op_gemm(const void *A, const void *B, float *C, int M, int N, int K, long a_rs, long a_cs, long b_rs, long b_cs,
                long c_rs, long c_cs, int batch, int inner, long a_s1, long a_s2, long b_s1, long b_s2, long c_s1, long c_s2,
                int a_dtype, int b_dtype, float alpha, int accumulate, int ksplit, long c_sk, void *stream)
{
    PTX_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && batch >= 1 && inner >= 1 && batch % inner == 0,
                "ptx_op_gemm: bad arguments (M=%d N=%d K=%d batch=%d inner=%d)", M, N, K, batch, inner);
    PTX_REQUIRE(a_dtype >= 0 && a_dtype <= 2 && b_dtype >= 0 && b_dtype <= 2 && batch <= 65535,
                "ptx_op_gemm: a_dtype=%d b_dtype=%d batch=%d", a_dtype, b_dtype, batch);
    PTX_REQUIRE(ksplit >= 1 && (long)batch * ksplit <= 65535 && (ksplit == 1 || !accumulate), "ptx_op_gemm: ksplit=%d", ksplit);
    BGemmArgs g{A, B, C, M, N, K, a_rs, a_cs, b_rs, b_cs, c_rs, c_cs, batch, inner, a_s1, a_s2, b_s1, b_s2, c_s1, c_s2,
                a_dtype, b_dtype, alpha, accumulate, ksplit, c_sk};
    const dim3 grid(cdiv(M, 64), cdiv(N, 64), batch * ksplit);
    hipStream_t st = static_cast<hipStream_t>(stream);
    PTX_REQUIRE(a_dtype == 0 || b_dtype == 0, "ptx_op_gemm: at most one 16-bit operand (a_dtype=%d, b_dtype=%d)", a_dtype, b_dtype);
    if (a_dtype == 0 && b_dtype == 0 && ksplit == 1 && batch >= 64 && (M <= 2 || N <= 2 || K <= 2)) {
        // thin products of many batches: most of a 64 x 64 MFMA tile would be padding
        const bool al16 = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
        if (M <= 2 && N == 32 && b_cs == 1 && K <= 256 && K > 64 && al16 && b_rs % 4 == 0 && b_s1 % 4 == 0 && b_s2 % 4 == 0) {
            const long rows = (long)batch * M;
            hipLaunchKernelGGL(k_bthin_row, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, g, rows);
            PTX_LAUNCHED("k_bthin_row");
            return PTX_OK;
        }
        const long total = (long)batch * M * N;
        if (K <= 64 && total < (1l << 31)) {
            const bool kvec = al16 && a_cs == 1 && b_rs == 1 && K % 4 == 0 && a_rs % 4 == 0 && b_cs % 4 == 0 && a_s1 % 4 == 0 &&
                              a_s2 % 4 == 0 && b_s1 % 4 == 0 && b_s2 % 4 == 0;
            const dim3 tg((unsigned)((total + 255) / 256));
            if (kvec) hipLaunchKernelGGL(k_bthin_out<true>, tg, dim3(256), 0, st, g, (unsigned)total);
            else      hipLaunchKernelGGL(k_bthin_out<false>, tg, dim3(256), 0, st, g, (unsigned)total);
            PTX_LAUNCHED("k_bthin_out");
            return PTX_OK;
        }
    }
    if (a_dtype == 0 && b_dtype == 0) hipLaunchKernelGGL((k_bgemm<0, 0>), grid, dim3(256), 0, st, g);
    else if (a_dtype == 1) hipLaunchKernelGGL((k_bgemm<1, 0>), grid, dim3(256), 0, st, g);
    else if (a_dtype == 2) hipLaunchKernelGGL((k_bgemm<2, 0>), grid, dim3(256), 0, st, g);
    else if (b_dtype == 1) hipLaunchKernelGGL((k_bgemm<0, 1>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_bgemm<0, 2>), grid, dim3(256), 0, st, g);
    PTX_LAUNCHED("k_bgemm");
    return PTX_OK;
}

int ptx_op_colsum(const float *x, const float *y, int R, int N, int mode, float scale, int accumulate, float *out,
                  double *scratch, int nsplit, void *stream)
{
    PTX_REQUIRE(x && out && scratch && R >= 1 && N >= 1 && mode >= 0 && mode <= 3 && ((mode != 1 && mode != 3) || y) &&
                nsplit >= 1 && nsplit <= 1024, "ptx_op_colsum: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = N % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && mode != 3 ? true
                     : (N % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && mode == 3);
    if (vec) hipLaunchKernelGGL(k_colsum<4>, dim3(cdiv(N, 256), nsplit), dim3(256), 0, st, x, y, R, N, mode, nsplit, scratch);
    else     hipLaunchKernelGGL(k_colsum<1>, dim3(cdiv(N, 64), nsplit), dim3(256), 0, st, x, y, R, N, mode, nsplit, scratch);
    PTX_LAUNCHED("k_colsum");
    hipLaunchKernelGGL(k_colsum_fin, dim3(cdiv(N, 16)), dim3(256), 0, st, scratch, N, nsplit, scale, accumulate, out);
    PTX_LAUNCHED("k_colsum_fin");
    return PTX_OK;
}

int ptx_op_eltwise(int op, const float *a, const float *b, float s, long n, int ncol, float *y, void *stream)
{
    PTX_REQUIRE(a && y && n >= 1 && op >= 0 && op <= 8, "ptx_op_eltwise: bad arguments");
    PTX_REQUIRE(b || (op == 1 || op == 2 || op == 4), "ptx_op_eltwise: op %d needs a second operand", op);
    hipLaunchKernelGGL(k_eltwise, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), op, a, b, s, n,
                       ncol > 0 ? ncol : 1, y);
    PTX_LAUNCHED("k_eltwise");
    return PTX_OK;
}

int ptx_op_dropout(const float *x, long n, long group, float p, uint64_t seed, float *y, void *stream)
{
    PTX_REQUIRE(x && y && n >= 1 && group >= 1 && p >= 0.0f && p < 1.0f, "ptx_op_dropout: bad arguments (p=%f)", p);
    hipLaunchKernelGGL(k_dropout, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, group, p, seed, y);
    PTX_LAUNCHED("k_dropout");
    return PTX_OK;
}

int ptx_op_layernorm_fwd(const float *x, const float *w, const float *b, const float *add, int add_rows, int R, int C,
                         float eps, float *y, float *stats, void *stream)
{
    PTX_REQUIRE(x && w && b && y && stats && R >= 1 && C % 64 == 0 && C <= 64 * kLnMax, "ptx_op_layernorm_fwd: bad arguments (C=%d)", C);
    hipLaunchKernelGGL(k_ln_fwd, dim3(cdiv(R, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, b, add,
                       add_rows > 0 ? add_rows : 1, R, C, eps, y, stats);
    PTX_LAUNCHED("k_ln_fwd");
    return PTX_OK;
}

int ptx_op_layernorm_bwd(const float *x, const float *w, const float *dy, const float *stats, int R, int C, float *dx,
                         float *xhat, void *stream)
{
    PTX_REQUIRE(x && w && dy && stats && dx && xhat && R >= 1 && C % 64 == 0 && C <= 64 * kLnMax, "ptx_op_layernorm_bwd: bad arguments");
    hipLaunchKernelGGL(k_ln_bwd, dim3(cdiv(R, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, dy, stats, R, C, dx, xhat);
    PTX_LAUNCHED("k_ln_bwd");
    return PTX_OK;
}

int ptx_op_bn_stats(const float *mean, const float *sumsq_centred, int C, long R, float eps, float momentum, float *mean_rstd,
                    float *run_mean, float *run_var, void *stream)
{
    PTX_REQUIRE(mean && sumsq_centred && mean_rstd && C >= 1 && R >= 1, "ptx_op_bn_stats: bad arguments");
    hipLaunchKernelGGL(k_bn_stats, dim3(cdiv(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mean, sumsq_centred, C, R, eps,
                       momentum, mean_rstd, run_mean, run_var);
    PTX_LAUNCHED("k_bn_stats");
    return PTX_OK;
}

int ptx_op_bn_apply(const float *x, const float *mean_rstd, const float *w, const float *b, long R, int C, int relu, float *y,
                    void *stream)
{
    PTX_REQUIRE(x && mean_rstd && w && b && y && R >= 1 && C >= 1, "ptx_op_bn_apply: bad arguments");
    hipLaunchKernelGGL(k_bn_apply, dim3(blocks_for(R * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, mean_rstd, w, b,
                       R * C, C, relu, y);
    PTX_LAUNCHED("k_bn_apply");
    return PTX_OK;
}

int ptx_op_bn_bwd_prep(const float *x, const float *y, const float *dy, const float *mean_rstd, long R, int C, int relu,
                       float *g, float *gx, void *stream)
{
    PTX_REQUIRE(x && dy && mean_rstd && g && gx && (!relu || y), "ptx_op_bn_bwd_prep: bad arguments");
    hipLaunchKernelGGL(k_bn_bwd_prep, dim3(blocks_for(R * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, dy,
                       mean_rstd, R * C, C, relu, g, gx);
    PTX_LAUNCHED("k_bn_bwd_prep");
    return PTX_OK;
}

int ptx_op_bn_bwd_dx(const float *x, const float *g, const float *mean_rstd, const float *w, const float *dbeta,
                     const float *dgamma, long R, int C, float *dx, void *stream)
{
    PTX_REQUIRE(x && g && mean_rstd && w && dbeta && dgamma && dx, "ptx_op_bn_bwd_dx: bad arguments");
    hipLaunchKernelGGL(k_bn_bwd_dx, dim3(blocks_for(R * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, g, mean_rstd, w,
                       dbeta, dgamma, R * C, C, R, dx);
    PTX_LAUNCHED("k_bn_bwd_dx");
    return PTX_OK;
}

int ptx_op_softmax_fwd(const float *s, const uint8_t *mask, long rows, int L, long rows_per_scene, float *p, void *stream)
{
    PTX_REQUIRE(s && p && rows >= 1 && L >= 1 && rows_per_scene >= 1, "ptx_op_softmax_fwd: bad arguments");
    hipLaunchKernelGGL(k_softmax_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), s, mask,
                       rows, L, rows_per_scene, p);
    PTX_LAUNCHED("k_softmax_fwd");
    return PTX_OK;
}

int ptx_op_softmax_bwd(const float *p, const float *dp, const uint8_t *mask, long rows, int L, long rows_per_scene, float *ds,
                       void *stream)
{
    PTX_REQUIRE(p && dp && ds && rows >= 1 && L >= 1 && rows_per_scene >= 1, "ptx_op_softmax_bwd: bad arguments");
    hipLaunchKernelGGL(k_softmax_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), p, dp,
                       mask, rows, L, rows_per_scene, ds);
    PTX_LAUNCHED("k_softmax_bwd");
    return PTX_OK;
}

int ptx_op_slot_inputs(const float *center, const float *cluster, const int32_t *src, long nclus, int K, float *x6,
                       uint8_t *padmask, void *stream)
{
    PTX_REQUIRE(center && cluster && x6 && padmask && nclus >= 1 && K >= 1, "ptx_op_slot_inputs: bad arguments");
    hipLaunchKernelGGL(k_slot_inputs, dim3(blocks_for(nclus * K)), dim3(256), 0, static_cast<hipStream_t>(stream), center,
                       cluster, src, nclus, K, x6, padmask);
    PTX_LAUNCHED("k_slot_inputs");
    return PTX_OK;
}

int ptx_op_slot_inputs_bwd(const float *dx6, const uint8_t *padmask, long nclus, int K, float *dcenter, void *stream)
{
    PTX_REQUIRE(dx6 && padmask && dcenter && nclus >= 1, "ptx_op_slot_inputs_bwd: bad arguments");
    hipLaunchKernelGGL(k_slot_inputs_bwd, dim3(blocks_for(nclus)), dim3(256), 0, static_cast<hipStream_t>(stream), dx6, padmask,
                       nclus, K, dcenter);
    PTX_LAUNCHED("k_slot_inputs_bwd");
    return PTX_OK;
}

int ptx_op_slot_pool(const float *h, long nclus, int K, int C, int mode, float *out, int32_t *arg, void *stream)
{
    PTX_REQUIRE(h && out && (mode == 0 || arg) && nclus >= 1, "ptx_op_slot_pool: bad arguments");
    hipLaunchKernelGGL(k_slot_pool, dim3(blocks_for(nclus * C)), dim3(256), 0, static_cast<hipStream_t>(stream), h, nclus, K, C,
                       mode, out, arg);
    PTX_LAUNCHED("k_slot_pool");
    return PTX_OK;
}

int ptx_op_slot_pool_bwd(const float *dout, const int32_t *arg, long nclus, int K, int C, int mode, float *dh, void *stream)
{
    PTX_REQUIRE(dout && dh && (mode == 0 || arg) && nclus >= 1, "ptx_op_slot_pool_bwd: bad arguments");
    hipLaunchKernelGGL(k_slot_pool_bwd, dim3(blocks_for(nclus * K * C)), dim3(256), 0, static_cast<hipStream_t>(stream), dout,
                       arg, nclus, K, C, mode, dh);
    PTX_LAUNCHED("k_slot_pool_bwd");
    return PTX_OK;
}

int ptx_op_offset_apply(const float *c0, const float *raw, const float *minmax, long nclus, int M, float margin, float *cout,
                        float *dcoef, void *stream)
{
    PTX_REQUIRE(c0 && raw && minmax && cout && dcoef && nclus >= 1 && M >= 1, "ptx_op_offset_apply: bad arguments");
    hipLaunchKernelGGL(k_offset_apply, dim3(blocks_for(nclus * 3)), dim3(256), 0, static_cast<hipStream_t>(stream), c0, raw,
                       minmax, nclus * 3, M, margin, cout, dcoef);
    PTX_LAUNCHED("k_offset_apply");
    return PTX_OK;
}

int ptx_op_slotbias_fwd(const float *pb, const float *pc, const float *pr, int Mk, int s, int C, float *table, void *stream)
{
    PTX_REQUIRE(pb && pc && pr && table && Mk >= 1 && s >= 1 && C <= s * s, "ptx_op_slotbias_fwd: bad arguments");
    hipLaunchKernelGGL(k_slotbias_fwd, dim3(cdiv(Mk * C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pb, pc, pr, Mk,
                       s, C, table);
    PTX_LAUNCHED("k_slotbias_fwd");
    return PTX_OK;
}

int ptx_op_slotbias_bwd(const float *dtable, int Mk, int s, int C, float *dpb, float *dpc, float *dpr, void *stream)
{
    PTX_REQUIRE(dtable && dpb && dpc && dpr && Mk >= 1, "ptx_op_slotbias_bwd: bad arguments");
    PTX_REQUIRE(s <= 23 && C <= 512, "ptx_op_slotbias_bwd: s=%d C=%d", s, C);
    hipLaunchKernelGGL(k_slotbias_bwd, dim3(cdiv(Mk, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), dtable, Mk, s, C, dpb,
                       dpc, dpr);
    PTX_LAUNCHED("k_slotbias_bwd");
    return PTX_OK;
}

int ptx_op_rows_gather(const float *x, const int32_t *src, long rows, int C, float *y, void *stream)
{
    PTX_REQUIRE(x && src && y && rows >= 1 && C >= 1, "ptx_op_rows_gather: bad arguments");
    hipLaunchKernelGGL(k_rows_gather, dim3(blocks_for(rows * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, src, rows, C, y);
    PTX_LAUNCHED("k_rows_gather");
    return PTX_OK;
}

int ptx_op_rows_scatter(const float *dy, const int32_t *src, long rows, int C, float *dx, void *stream)
{
    PTX_REQUIRE(dy && src && dx && rows >= 1 && C >= 1, "ptx_op_rows_scatter: bad arguments");
    hipLaunchKernelGGL(k_rows_scatter, dim3(blocks_for(rows * C)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, src, rows, C, dx);
    PTX_LAUNCHED("k_rows_scatter");
    return PTX_OK;
}

int ptx_op_keep_rows(const int32_t *order, const int32_t *keep, int B, int M, int Mt, int Mk, int32_t *src, void *stream)
{
    PTX_REQUIRE(order && keep && src && B >= 1, "ptx_op_keep_rows: bad arguments");
    hipLaunchKernelGGL(k_keep_rows, dim3(cdiv(B * Mk, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), order, keep, B, M, Mt, Mk, src);
    PTX_LAUNCHED("k_keep_rows");
    return PTX_OK;
}

int ptx_op_affine_bwd(const float *dout, const int32_t *opos, const int32_t *kidx, const float *kcluster,
                      const float *kcenter, const float *transform, int B, int N, int Mk, int K, float *dtranslate,
                      float *dtransform, float *dkcenter, void *stream)
{
    PTX_REQUIRE(dout && opos && kidx && kcluster && kcenter && transform && dtranslate && dtransform && dkcenter && K <= 64,
                "ptx_op_affine_bwd: bad arguments");
    hipLaunchKernelGGL(k_affine_bwd, dim3(cdiv(B * Mk, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), dout, DoutList{}, opos, kidx,
                       kcluster, kcenter, transform, B, N, Mk, K, dtranslate, dtransform, dkcenter);
    PTX_LAUNCHED("k_affine_bwd");
    return PTX_OK;
}

/* the same with one gradient per scene (the module returns a LIST of (n_b,3) tensors, PRE:467): douts [host] = B device
 * pointers, NULL where a scene's output received no gradient */
int ptx_op_affine_bwd_list(const float *const *douts, const int32_t *opos, const int32_t *kidx, const float *kcluster,
                           const float *kcenter, const float *transform, int B, int N, int Mk, int K, float *dtranslate,
                           float *dtransform, float *dkcenter, void *stream)
{
    PTX_REQUIRE(douts && opos && kidx && kcluster && kcenter && transform && dtranslate && dtransform && dkcenter && K <= 64 &&
                B >= 1 && B <= 32, "ptx_op_affine_bwd_list: bad arguments");
    DoutList dl{};
    for (int b = 0; b < B; ++b) dl.p[b] = douts[b];
    hipLaunchKernelGGL(k_affine_bwd, dim3(cdiv(B * Mk, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), (const float *)nullptr, dl,
                       opos, kidx, kcluster, kcenter, transform, B, N, Mk, K, dtranslate, dtransform, dkcenter);
    PTX_LAUNCHED("k_affine_bwd");
    return PTX_OK;
}

/* output position of every input point after remove_points_by_index (PRE:516-523), -1 = dropped; also the per-scene
 * survivor counts.  tile_counts: scratch of B * ceil(N / 2048) int32. */
int ptx_op_out_positions(const uint32_t *tag, int B, int N, int32_t *tile_counts, int32_t *opos, int32_t *counts, void *stream)
{
    PTX_REQUIRE(tag && tile_counts && opos && counts && B >= 1 && N >= 1, "ptx_op_out_positions: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PTX_TRY(launch_tile_count(tag, B, N, tile_counts, counts, nullptr, st));
    hipLaunchKernelGGL(k_out_positions, dim3(cdiv(N, kTilePts), B), dim3(256), 0, st, tag, tile_counts, N, opos);
    PTX_LAUNCHED("k_out_positions");
    return PTX_OK;
}

int ptx_op_tokens_finish(float *tok, const float *pos, int nimg, int hw, int C, void *stream)
{
    PTX_REQUIRE(tok && pos && nimg >= 1 && hw >= 1, "ptx_op_tokens_finish: bad arguments");
    hipLaunchKernelGGL(k_tokens_finish, dim3(blocks_for((long)nimg * C)), dim3(256), 0, static_cast<hipStream_t>(stream), tok, pos, nimg, hw, C);
    PTX_LAUNCHED("k_tokens_finish");
    return PTX_OK;
}

int ptx_op_tokens_finish_bwd(float *dtok, int nimg, int hw, int C, void *stream)
{
    PTX_REQUIRE(dtok && nimg >= 1 && hw >= 1, "ptx_op_tokens_finish_bwd: bad arguments");
    hipLaunchKernelGGL(k_tokens_finish_bwd, dim3(blocks_for((long)nimg * C)), dim3(256), 0, static_cast<hipStream_t>(stream), dtok, nimg, hw, C);
    PTX_LAUNCHED("k_tokens_finish_bwd");
    return PTX_OK;
}

}  // extern "C"
