// One ProxyBlock of the training step as two enqueue calls (include/proxyt.h "one ProxyBlock ... as two calls").
//
// r04: the training step issued ~400 launches one ctypes call at a time -- 4.9 ms of host enqueue for 3.7 ms of kernels, a
// third of those kernels the tiny batched products, soft-maxes and dropouts of proxy attention and the two-launch column
// sums behind every bias / LayerNorm gradient.  This file keeps the arithmetic of proxytransformation_amd/train.py's nodes
// (same formulas, same dropout masks for the same seeds) and changes how it is issued:
//   * ptx_train_block_fwd / _bwd enqueue the whole chain  norm1 (+ slot bias) -> qkv / proxy_proj -> proxy attention -> proj
//     -> Dropout + DropPath + residual -> norm2 -> fc1 -> GELU + Dropout -> fc2 -> Dropout + DropPath + residual -> trailing
//     LayerNorm -> Linear head -> BatchNorm1d (batch statistics)  (PRE:273-276, 441-446) from C++;
//   * proxy attention (PRE:230-252) is five kernels instead of 22 launches: forward A (per head, four proxies per work-group:
//     scores against every token, soft-max over the tokens, dropout, P V), forward B (64 tokens per work-group: scores against
//     the proxies, masked soft-max, dropout, output), backward A / B / C (token tile -> proxy rows -> token tile, the
//     reductions over tokens as per-tile partials summed in tile order);
//   * residual + Dropout + DropPath + LayerNorm are one pass over the rows, and the kernels that PRODUCE the rows of a
//     gradient also emit its column sums per 16-row chunk; one fixed-order pass at the end of the backward (k_t_finalize, a
//     job table) turns every partial -- bias, LayerNorm, slot-bias table and K-sliced weight gradients -- into its parameter
//     gradient.  Fixed summation orders everywhere: two runs give the same bits.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "split3.h"

namespace ptx {

// ------------------------------------------------------------------------------ dropout sites
// keep(seed, i) is train_ops.hip's k_dropout: the masks of a (seed, element) pair are the same in both files
__device__ __forceinline__ uint32_t tf_mix32(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((x ^ (x >> 31)) >> 32);
}
struct Drop1 { uint64_t seed; uint32_t thresh; float ks; int on; };
static Drop1 make_drop(float p, uint64_t seed)
{
    Drop1 d;
    d.on = p > 0.0f ? 1 : 0; d.seed = seed * 0x100000001B3ull;
    d.thresh = (uint32_t)((double)p * 4294967296.0); d.ks = 1.0f / (1.0f - p);
    return d;
}
__device__ __forceinline__ float drop_apply(const Drop1 &d, float v, uint64_t i)
{
    if (!d.on) return v;
    return tf_mix32(d.seed + i) >= d.thresh ? v * d.ks : 0.0f;
}
__device__ __forceinline__ float tf_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float tf_gelu_g(float x)
{
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

constexpr int kRowsPerChunk = 8;       // rows of one work-group in the backward row kernels (4 waves x 2 rows: ~R / 8 work-groups
                                       // fill the chip; 16 rows left these passes latency-bound at 260 work-groups, r04)
constexpr int kMaxQ = 8;               // C <= 512: eight values per lane

// ------------------------------------------------------------------------------ residual + dropouts + LayerNorm, forward
// x = a + DropPath(Dropout(b))  (b == null: x = a);  y = LN(x) w + bias (+ add[row % add_rows]);  one wave per row.
struct LnFwdArgs {
    const float *a, *b; Drop1 d1, d2; int rows_per_scene;
    const float *w, *bias, *add; int add_rows; int R, C; float eps;
    float *xout, *y, *stats;
};
__global__ __launch_bounds__(256) void k_t_ln_fwd(LnFwdArgs g)
{
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= g.R) return;
    const int lane = lane_id(), C = g.C;
    const size_t base = (size_t)row * C;
    const uint64_t scene = (uint64_t)(row / g.rows_per_scene);
    float v[kMaxQ], s = 0.0f;
#pragma unroll
    for (int q = 0; q < kMaxQ; ++q) {
        const int c = lane + 64 * q;
        float t = 0.0f;
        if (c < C) {
            t = g.a[base + c];
            if (g.b) {
                float o = drop_apply(g.d1, g.b[base + c], base + c);
                o = drop_apply(g.d2, o, scene);
                t = t + o;
            }
            if (g.xout) g.xout[base + c] = t;
        }
        v[q] = t; s += t;
    }
    const float mean = wave_sum(s) / (float)C;
    float var = 0.0f;
#pragma unroll
    for (int q = 0; q < kMaxQ; ++q) { const int c = lane + 64 * q; const float d = c < C ? v[q] - mean : 0.0f; var = fmaf(d, d, var); }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)C + g.eps);
    if (lane == 0) { g.stats[2 * row] = mean; g.stats[2 * row + 1] = rstd; }
#pragma unroll
    for (int q = 0; q < kMaxQ; ++q) {
        const int c = lane + 64 * q;
        if (c < C) {
            float o = (v[q] - mean) * rstd * g.w[c] + g.bias[c];
            if (g.add) o += g.add[(size_t)(row % g.add_rows) * C + c];
            g.y[base + c] = o;
        }
    }
}

// y = Dropout(gelu(x)), element-wise over (R, H), H % 4 == 0
__global__ __launch_bounds__(256) void k_t_gelu_drop(const float *__restrict__ x, long n4, Drop1 d, float *__restrict__ y)
{
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        float4 o;
        o.x = drop_apply(d, tf_gelu(v.x), 4 * i); o.y = drop_apply(d, tf_gelu(v.y), 4 * i + 1);
        o.z = drop_apply(d, tf_gelu(v.z), 4 * i + 2); o.w = drop_apply(d, tf_gelu(v.w), 4 * i + 3);
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}

// ------------------------------------------------------------------------------ BatchNorm1d over (R, nout <= 9 columns)
// (the head's output is 3 or 9 columns wide: six launches of the general column-sum machinery for 37 k values in r03; one
//  1024-thread work-group walking them three times took 25-40 us, r04).  Two launches of kBnParts work-groups: per-part sums
//  in double, then every work-group adds the parts in part order and normalises its share of the rows.
constexpr int kBnMaxC = 9;      // = kHeadMax
constexpr int kBnParts = 8;
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// part[(blockIdx.x * 2 + k) * kBnMaxC + j] = sum over this work-group's elements of column j of f_k:
//   MODE 0: f_0 = x, f_1 = x^2        MODE 1: f_0 = dy, f_1 = dy * xhat (mr = mean, rstd)
template <int MODE>
__global__ __launch_bounds__(256) void k_t_bn_part(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ mr,
                                                   int R, int nc, double *__restrict__ part)
{
    __shared__ double red[4][2][kBnMaxC];
    const long n = (long)R * nc, per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    double a0[kBnMaxC], a1[kBnMaxC];
#pragma unroll
    for (int j = 0; j < kBnMaxC; ++j) { a0[j] = 0.0; a1[j] = 0.0; }
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const int c = (int)(i % nc);
        double f0, f1;
        if (MODE == 0) { f0 = (double)x[i]; f1 = f0 * f0; }
        else { const float g = dy[i]; f0 = (double)g; f1 = (double)(g * ((x[i] - mr[c]) * mr[nc + c])); }
#pragma unroll
        for (int j = 0; j < kBnMaxC; ++j) if (j == c) { a0[j] += f0; a1[j] += f1; }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < kBnMaxC; ++j) {
        if (j < nc) {
            const double s0 = wave_sum_d(a0[j]), s1 = wave_sum_d(a1[j]);
            if (lane == 0) { red[wv][0][j] = s0; red[wv][1][j] = s1; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kBnMaxC) {
        const int k = threadIdx.x / kBnMaxC, j = threadIdx.x % kBnMaxC;
        part[((size_t)blockIdx.x * 2 + k) * kBnMaxC + j] = j < nc ? ((red[0][k][j] + red[1][k][j]) + red[2][k][j]) + red[3][k][j] : 0.0;
    }
}
// forward: mean / biased variance from the parts (sum x^2 - R mean^2 in double: fp32 data leave 29 digits of headroom),
// running statistics (work-group 0), y = (x - mean) rstd w + b
__global__ __launch_bounds__(256) void k_t_bn_small_fwd(const float *__restrict__ x, const double *__restrict__ part, int nparts, int R,
                                                        int nc, const float *__restrict__ w, const float *__restrict__ b, float eps,
                                                        float momentum, float *__restrict__ run_mean, float *__restrict__ run_var,
                                                        float *__restrict__ y, float *__restrict__ mr)
{
    __shared__ float s_mean[kBnMaxC], s_rstd[kBnMaxC];
    if (threadIdx.x < nc) {
        const int c = threadIdx.x;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int p = 0; p < nparts; ++p) { s0 += part[((size_t)p * 2) * kBnMaxC + c]; s1 += part[((size_t)p * 2 + 1) * kBnMaxC + c]; }
        const double mean = s0 / (double)R;
        double var = s1 / (double)R - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = 1.0f / sqrtf((float)var + eps);
        s_mean[c] = (float)mean; s_rstd[c] = rstd;
        if (blockIdx.x == 0) {
            mr[c] = (float)mean; mr[nc + c] = rstd;
            if (run_mean) {
                const float unb = R > 1 ? (float)(var * (double)R / (double)(R - 1)) : (float)var;
                run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
                run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
            }
        }
    }
    __syncthreads();
    const long n = (long)R * nc;
    for (long i = blockIdx.x * 256l + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % nc);
        y[i] = (x[i] - s_mean[c]) * s_rstd[c] * w[c] + b[c];
    }
}
// backward: dbeta = sum dy, dgamma = sum dy xhat (from the parts), dx = w rstd (dy - dbeta / R - xhat dgamma / R); the column
// sums of dx (the bias gradient of the Linear in front: zero up to rounding, evaluated like the reference evaluates it) leave
// as one partial per work-group for the block's final fixed-order pass
__global__ __launch_bounds__(256) void k_t_bn_small_bwd(const float *__restrict__ x, const float *__restrict__ mr,
                                                        const float *__restrict__ w, const float *__restrict__ dy,
                                                        const double *__restrict__ part, int nparts, int R, int nc,
                                                        float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                        float *__restrict__ dxpart)
{
    __shared__ float s_db[kBnMaxC], s_dg[kBnMaxC];
    __shared__ float red[4][kBnMaxC];
    if (threadIdx.x < nc) {
        const int c = threadIdx.x;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int p = 0; p < nparts; ++p) { s0 += part[((size_t)p * 2) * kBnMaxC + c]; s1 += part[((size_t)p * 2 + 1) * kBnMaxC + c]; }
        s_db[c] = (float)s0; s_dg[c] = (float)s1;
        if (blockIdx.x == 0) { dbeta[c] = (float)s0; dgamma[c] = (float)s1; }
    }
    __syncthreads();
    const float invR = 1.0f / (float)R;
    const long n = (long)R * nc, per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    float acc[kBnMaxC];
#pragma unroll
    for (int j = 0; j < kBnMaxC; ++j) acc[j] = 0.0f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const int c = (int)(i % nc);
        const float xh = (x[i] - mr[c]) * mr[nc + c];
        const float v = w[c] * mr[nc + c] * (dy[i] - s_db[c] * invR - xh * s_dg[c] * invR);
        dx[i] = v;
#pragma unroll
        for (int j = 0; j < kBnMaxC; ++j) if (j == c) acc[j] += v;
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < kBnMaxC; ++j) {
        if (j < nc) { const float sv = wave_sum(acc[j]); if (lane == 0) red[wv][j] = sv; }
    }
    __syncthreads();
    if (threadIdx.x < nc) dxpart[(size_t)blockIdx.x * nc + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// ------------------------------------------------------------------------------ backward row kernels with column partials
// A work-group owns kRowsPerChunk rows (wave w: rows w, w + 4, ...); lane l owns columns l + 64 q.  Column sums of the chunk
// are combined across the four waves in wave order and written to part[(chunk * nsum + k) * ncols + c].
template <int NS, int MQ>
__device__ __forceinline__ void chunk_partials(float (&acc)[NS][MQ], int nq, int ncols, float *__restrict__ part, float *red)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int q = 0; q < MQ; ++q)
            if (q < nq) red[(wv * NS + k) * ncols + lane + 64 * q] = acc[k][q];
    __syncthreads();
    for (int i = threadIdx.x; i < NS * ncols; i += 256) {
        const float t = ((red[i] + red[NS * ncols + i]) + red[2 * NS * ncols + i]) + red[3 * NS * ncols + i];
        part[(size_t)blockIdx.x * NS * ncols + i] = t;
    }
}

// LayerNorm backward: dx = rstd (g - mean(g) - xhat mean(g xhat)) + dres, g = dy w;  partials: dgamma = sum dy xhat,
// dbeta = sum dy;  optionally dd = DropPath'(Dropout'(dx)) with its own column partial (the bias gradient of the Linear
// whose output went through those dropouts into the residual)
struct LnBwdArgs {
    const float *x, *stats, *w, *dy, *dres; Drop1 d1, d2; int rows_per_scene; int R, C;
    float *dx, *dd; float *part;      // part: [chunks][2 or 3][C]
    const float *dres2;               // a second gradient of the same rows to add (the other block's use of the point proxies)
};
template <bool DD>
__global__ __launch_bounds__(256) void k_t_ln_bwd(LnBwdArgs g)
{
    extern __shared__ float red[];
    constexpr int NS = DD ? 3 : 2;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, C = g.C, nq = C / 64;
    if ((C & 255) == 0) {
        // r05: 16-byte accesses -- lane l owns columns 256 u + 4 l .. + 3 (C = 256: one float4 per array and row); column partials in
        // the same order as in the dword form below (rows in wave order, then the four waves)
        constexpr int NG = kMaxQ / 4;
        const int ng = C >> 8;
        float acc4[NS][NG][4];
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int u = 0; u < NG; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc4[k][u][j] = 0.0f;
        float w4[NG][4];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const float4 t = u < ng ? *reinterpret_cast<const float4 *>(g.w + 256 * u + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
            w4[u][0] = t.x; w4[u][1] = t.y; w4[u][2] = t.z; w4[u][3] = t.w;
        }
        for (int rr = wv; rr < kRowsPerChunk; rr += 4) {
            const int row = blockIdx.x * kRowsPerChunk + rr;
            if (row >= g.R) break;
            const size_t base = (size_t)row * C;
            const float mean = g.stats[2 * row], rstd = g.stats[2 * row + 1];
            float xh[NG][4], gg[NG][4], dyv[NG][4], r1[NG][4], r2[NG][4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const bool on = u < ng;
                const size_t i = base + 256 * u + 4 * lane;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 xv = on ? *reinterpret_cast<const float4 *>(g.x + i) : z, dv = on ? *reinterpret_cast<const float4 *>(g.dy + i) : z;
                const float4 a1 = (on && g.dres) ? *reinterpret_cast<const float4 *>(g.dres + i) : z;
                const float4 a2 = (on && g.dres2) ? *reinterpret_cast<const float4 *>(g.dres2 + i) : z;
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
                r1[u][0] = a1.x; r1[u][1] = a1.y; r1[u][2] = a1.z; r1[u][3] = a1.w;
                r2[u][0] = a2.x; r2[u][1] = a2.y; r2[u][2] = a2.z; r2[u][3] = a2.w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[u][j] = on ? (xs[j] - mean) * rstd : 0.0f;
                    dyv[u][j] = ds[j];
                    gg[u][j] = ds[j] * w4[u][j];
                    s1 += gg[u][j]; s2 = fmaf(gg[u][j], xh[u][j], s2);
                }
            }
            s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
            const uint64_t scene = (uint64_t)(row / g.rows_per_scene);
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                if (u < ng) {
                    const size_t i = base + 256 * u + 4 * lane;
                    float v[4], o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = rstd * (gg[u][j] - s1 - xh[u][j] * s2);
                        if (g.dres) v[j] = r1[u][j] + v[j];
                        if (g.dres2) v[j] = v[j] + r2[u][j];
                        acc4[0][u][j] += dyv[u][j] * xh[u][j];
                        acc4[1][u][j] += dyv[u][j];
                        if (DD) {
                            o[j] = drop_apply(g.d2, v[j], scene);
                            o[j] = drop_apply(g.d1, o[j], i + j);
                            acc4[NS - 1][u][j] += o[j];
                        }
                    }
                    *reinterpret_cast<float4 *>(g.dx + i) = make_float4(v[0], v[1], v[2], v[3]);
                    if (DD) *reinterpret_cast<float4 *>(g.dd + i) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int u = 0; u < NG; ++u)
                if (u < ng)
                    *reinterpret_cast<float4 *>(&red[(wv * NS + k) * C + 256 * u + 4 * lane]) =
                        make_float4(acc4[k][u][0], acc4[k][u][1], acc4[k][u][2], acc4[k][u][3]);
        __syncthreads();
        for (int i = threadIdx.x; i < NS * C; i += 256)
            g.part[(size_t)blockIdx.x * NS * C + i] = ((red[i] + red[NS * C + i]) + red[2 * NS * C + i]) + red[3 * NS * C + i];
        return;
    }
    float acc[NS][kMaxQ];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) acc[k][q] = 0.0f;
    for (int rr = wv; rr < kRowsPerChunk; rr += 4) {
        const int row = blockIdx.x * kRowsPerChunk + rr;
        if (row >= g.R) break;
        const size_t base = (size_t)row * C;
        const float mean = g.stats[2 * row], rstd = g.stats[2 * row + 1];
        float xh[kMaxQ], gg[kMaxQ], dyv[kMaxQ], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {
            const int c = lane + 64 * q;
            if (q < nq) {
                xh[q] = (g.x[base + c] - mean) * rstd;
                dyv[q] = g.dy[base + c];
                gg[q] = dyv[q] * g.w[c];
            } else { xh[q] = 0.0f; gg[q] = 0.0f; dyv[q] = 0.0f; }
            s1 += gg[q]; s2 = fmaf(gg[q], xh[q], s2);
        }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
        const uint64_t scene = (uint64_t)(row / g.rows_per_scene);
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {
            const int c = lane + 64 * q;
            if (q < nq) {
                float v = rstd * (gg[q] - s1 - xh[q] * s2);
                if (g.dres) v = g.dres[base + c] + v;
                if (g.dres2) v = v + g.dres2[base + c];
                g.dx[base + c] = v;
                acc[0][q] += dyv[q] * xh[q];
                acc[1][q] += dyv[q];
                if (DD) {
                    float o = drop_apply(g.d2, v, scene);          // the backward of y = DropPath(Dropout(u)) applies both masks
                    o = drop_apply(g.d1, o, base + c);
                    g.dd[base + c] = o;
                    acc[NS - 1][q] += o;
                }
            }
        }
    }
    chunk_partials<NS, kMaxQ>(acc, nq, C, g.part, red);
}

// dhpre = gelu'(hpre) * Dropout'(dhact), with its column partial (fc1's bias gradient); (R, H), H <= 2048
constexpr int kMaxQH = 32;
__global__ __launch_bounds__(256) void k_t_gelu_bwd(const float *__restrict__ hpre, const float *__restrict__ dhact, Drop1 d, int R,
                                                    int H, float *__restrict__ dhpre, float *__restrict__ part)
{
    extern __shared__ float red[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((H & 255) == 0) {
        // r05: 16-byte accesses -- lane l owns columns 256 g + 4 l .. + 3 (the dword form below moved 51 MB in 35 us at the training
        // shape); the same sums in the same order per column: rows in wave order, waves combined as in chunk_partials
        const int ng = H >> 8;
        float acc[kMaxQH / 4][4];
#pragma unroll
        for (int g = 0; g < kMaxQH / 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][j] = 0.0f;
        for (int rr = wv; rr < kRowsPerChunk; rr += 4) {
            const int row = blockIdx.x * kRowsPerChunk + rr;
            if (row >= R) break;
            const size_t base = (size_t)row * H;
            float4 hv[kMaxQH / 4], dv[kMaxQH / 4];
#pragma unroll
            for (int g = 0; g < kMaxQH / 4; ++g)
                if (g < ng) {
                    const size_t i = base + 256 * g + 4 * lane;
                    hv[g] = *reinterpret_cast<const float4 *>(hpre + i);
                    dv[g] = *reinterpret_cast<const float4 *>(dhact + i);
                }
#pragma unroll
            for (int g = 0; g < kMaxQH / 4; ++g)
                if (g < ng) {
                    const size_t i = base + 256 * g + 4 * lane;
                    float4 o;
                    o.x = drop_apply(d, dv[g].x, i) * tf_gelu_g(hv[g].x);
                    o.y = drop_apply(d, dv[g].y, i + 1) * tf_gelu_g(hv[g].y);
                    o.z = drop_apply(d, dv[g].z, i + 2) * tf_gelu_g(hv[g].z);
                    o.w = drop_apply(d, dv[g].w, i + 3) * tf_gelu_g(hv[g].w);
                    *reinterpret_cast<float4 *>(dhpre + i) = o;
                    acc[g][0] += o.x; acc[g][1] += o.y; acc[g][2] += o.z; acc[g][3] += o.w;
                }
        }
#pragma unroll
        for (int g = 0; g < kMaxQH / 4; ++g)
            if (g < ng) *reinterpret_cast<float4 *>(&red[wv * H + 256 * g + 4 * lane]) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += 256)
            part[(size_t)blockIdx.x * H + i] = ((red[i] + red[H + i]) + red[2 * H + i]) + red[3 * H + i];
        return;
    }
    const int nq = H / 64;
    float acc[1][kMaxQH];
#pragma unroll
    for (int q = 0; q < kMaxQH; ++q) acc[0][q] = 0.0f;
    for (int rr = wv; rr < kRowsPerChunk; rr += 4) {
        const int row = blockIdx.x * kRowsPerChunk + rr;
        if (row >= R) break;
        const size_t base = (size_t)row * H;
#pragma unroll
        for (int q = 0; q < kMaxQH; ++q) {
            if (q < nq) {
                const size_t i = base + lane + 64 * q;
                const float v = drop_apply(d, dhact[i], i) * tf_gelu_g(hpre[i]);
                dhpre[i] = v; acc[0][q] += v;
            }
        }
    }
    chunk_partials<1, kMaxQH>(acc, nq, H, part, red);
}

// plain column partials of a dense (R, N) matrix, N % 64 == 0, N <= 2048 (the qkv bias gradient)
__global__ __launch_bounds__(256) void k_t_colpart(const float *__restrict__ x, int R, int N, float *__restrict__ part)
{
    extern __shared__ float red[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nq = N / 64;
    float acc[1][kMaxQH];
#pragma unroll
    for (int q = 0; q < kMaxQH; ++q) acc[0][q] = 0.0f;
    for (int rr = wv; rr < kRowsPerChunk; rr += 4) {
        const int row = blockIdx.x * kRowsPerChunk + rr;
        if (row >= R) break;
#pragma unroll
        for (int q = 0; q < kMaxQH; ++q)
            if (q < nq) acc[0][q] += x[(size_t)row * N + lane + 64 * q];
    }
    chunk_partials<1, kMaxQH>(acc, nq, N, part, red);
}

// Linear head backward (nout <= 9 columns): dg[r][c] = sum_j dt[r][j] W[j][c];  partial dW[j][c] = sum_r dt[r][j] g[r][c]
constexpr int kHeadMax = 9;
__global__ __launch_bounds__(256) void k_t_head_bwd(const float *__restrict__ dt, const float *__restrict__ coef,
                                                    const float *__restrict__ gin, const float *__restrict__ W, int R, int C, int nout,
                                                    float *__restrict__ dg, float *__restrict__ part)
{
    extern __shared__ float red[];               // [4][nout][C] for the partials; W staged behind it
    float *Ws = red + 4 * nout * C;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nq = C / 64;
    for (int i = threadIdx.x; i < nout * C; i += 256) Ws[i] = W[i];
    __syncthreads();
    float acc[kHeadMax][kMaxQ];
#pragma unroll
    for (int j = 0; j < kHeadMax; ++j)
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) acc[j][q] = 0.0f;
    for (int rr = wv; rr < kRowsPerChunk; rr += 4) {
        const int row = blockIdx.x * kRowsPerChunk + rr;
        if (row >= R) break;
        float t[kHeadMax];
#pragma unroll
        for (int j = 0; j < kHeadMax; ++j) t[j] = j < nout ? dt[(size_t)row * nout + j] * (coef ? coef[(size_t)row * nout + j] : 1.0f) : 0.0f;
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {
            if (q < nq) {
                const int c = lane + 64 * q;
                const float gv = gin[(size_t)row * C + c];
                float o = 0.0f;
#pragma unroll
                for (int j = 0; j < kHeadMax; ++j)
                    if (j < nout) { o = fmaf(t[j], Ws[j * C + c], o); acc[j][q] = fmaf(t[j], gv, acc[j][q]); }
                dg[(size_t)row * C + c] = o;
            }
        }
    }
    __syncthreads();
    // partials: the general helper with a run-time number of sums
#pragma unroll
    for (int j = 0; j < kHeadMax; ++j)
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q)
            if (j < nout && q < nq) red[(wv * nout + j) * C + lane + 64 * q] = acc[j][q];
    __syncthreads();
    const int tot = nout * C;
    for (int i = threadIdx.x; i < tot; i += 256)
        part[(size_t)blockIdx.x * tot + i] = ((red[i] + red[tot + i]) + red[2 * tot + i]) + red[3 * tot + i];
}

// ------------------------------------------------------------------------------ fixed-order sums of partials (one launch, many jobs)
// out[c] = sum_p part[p * pstride + c]: 16 interleaved slices of the parts per column, each in part order, combined in slice
// order; accumulated in double
constexpr int kMaxJobs = 28;
struct FinJob { const float *part; float *out; int nparts; int ncols; long pstride; };
struct FinJobs { FinJob j[kMaxJobs]; int blk0[kMaxJobs + 1]; int n; };
__global__ __launch_bounds__(256) void k_t_finalize(FinJobs J)
{
    __shared__ double red[256];
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.blk0[k + 1]) ++k;
    const FinJob jb = J.j[k];
    // few parts (K slices of a weight gradient: wide rows, 256-B requests) -> 64 columns x 4 slices per work-group;
    // many parts (row chunks) -> 16 columns x 16 slices, so that a thread's walk over the parts stays short
    const int cpb = jb.nparts <= 64 ? 64 : 16, ns = 256 / cpb;
    const int cl = threadIdx.x % cpb, sl = threadIdx.x / cpb, c = ((int)blockIdx.x - J.blk0[k]) * cpb + cl;
    double t = 0.0;
    if (c < jb.ncols) {
        int p = sl;
        for (; p + 3 * ns < jb.nparts; p += 4 * ns) {            // four loads in flight, added in part order
            const float a0 = jb.part[(size_t)p * jb.pstride + c], a1 = jb.part[(size_t)(p + ns) * jb.pstride + c];
            const float a2 = jb.part[(size_t)(p + 2 * ns) * jb.pstride + c], a3 = jb.part[(size_t)(p + 3 * ns) * jb.pstride + c];
            t += (double)a0; t += (double)a1; t += (double)a2; t += (double)a3;
        }
        for (; p < jb.nparts; p += ns) t += (double)jb.part[(size_t)p * jb.pstride + c];
    }
    red[sl * cpb + cl] = t;
    __syncthreads();
    if (sl == 0 && c < jb.ncols) {
        double tot = 0.0;
        for (int i = 0; i < ns; ++i) tot += red[i * cpb + cl];
        jb.out[c] = (float)tot;
    }
}
struct FinList {
    FinJobs J; int blocks;
    FinList() { J.n = 0; blocks = 0; J.blk0[0] = 0; }
    bool add(const float *part, float *out, int nparts, long ncols, long pstride)
    {
        if (J.n >= kMaxJobs || out == nullptr) return out == nullptr;
        J.j[J.n] = FinJob{part, out, nparts, (int)ncols, pstride};
        blocks += (int)((ncols + (nparts <= 64 ? 63 : 15)) / (nparts <= 64 ? 64 : 16));
        J.n += 1; J.blk0[J.n] = blocks;
        return true;
    }
    int launch(hipStream_t st)
    {
        if (J.n == 0) return PTX_OK;
        hipLaunchKernelGGL(k_t_finalize, dim3(blocks), dim3(256), 0, st, J);
        PTX_LAUNCHED("k_t_finalize");
        return PTX_OK;
    }
};

// ------------------------------------------------------------------------------ weight gradients: grouped TN products, K in slices
// C[m][n] = sum_k A[k][m] B[k][n] for A (K, M) = the rows' output gradients and B (K, N) = the rows' inputs, both row-major: the
// contraction runs over the ROWS (K = B * n tokens), the result is a small (M, N) weight.  All weight gradients of a block
// in ONE launch; every 64 x 64 tile is cut into `ks` slices of K whose partial products land c_sk floats apart and are summed in
// slice order by k_t_finalize.  Operands are split into three bf16 parts on the way into LDS (split3.h: six matrix-core
// products per 16 k, fp32-equivalent) -- the generic strided kernel did this on the fp32 matrix instruction at 1/16 of the bf16
// rate with scalar requests (r04: 2 x 156 us per step).  A lane requests eight consecutive k of ONE column m (coalesced along m
// across the wave), which is exactly the eight-k fragment the 32 x 32 x 16 instruction wants: split in registers, one 16-byte
// LDS write per part, same swizzled row layout as k_gemm64x.
struct TnProb { const float *A, *B; float *C; int M, N, K, ks; long c_sk; int tiles_n, tiles; };
struct TnBatch { TnProb p[6]; int blk0[7]; int n; };
template <int NP>     // 3: split operands (fp32-equivalent); 1: plain bf16 operands (compute_dtype 1)
__global__ __launch_bounds__(256) void k_t_tn_gemm(TnBatch tb)
{
    constexpr int kPlane = 64 * XROW, kBuf = 6 * kPlane;        // [A1 A2 A3 B1 B2 B3], 64 rows x 32 k each
    __shared__ __attribute__((aligned(16))) char smem[2 * kBuf];
    int pi = 0;
    while (pi + 1 < tb.n && (int)blockIdx.x >= tb.blk0[pi + 1]) ++pi;
    const TnProb pr = tb.p[pi];
    const int b = (int)blockIdx.x - tb.blk0[pi], slice = b / pr.tiles, tile = b - slice * pr.tiles;
    const int m0 = (tile / pr.tiles_n) * 64, n0 = (tile % pr.tiles_n) * 64;
    const int kper = ((pr.K + pr.ks - 1) / pr.ks + 31) / 32 * 32, kbeg = slice * kper, kend = min(pr.K, kbeg + kper);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int sm_ = tid & 63, koct = tid >> 6;                  // staging: column sm_ of both tiles, k octet koct of the step
    const float *Ap = pr.A + min(m0 + sm_, pr.M - 1), *Bp = pr.B + min(n0 + sm_, pr.N - 1);
    const bool aok = m0 + sm_ < pr.M, bok = n0 + sm_ < pr.N;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float xa[8], xb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(k0 + koct * 8 + u, pr.K - 1);
            xa[u] = Ap[(size_t)k * pr.M]; xb[u] = Bp[(size_t)k * pr.N];
        }
    };
    auto stash = [&](int k0, int buf) {
        float ya[8], yb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool kin = k0 + koct * 8 + u < kend;
            ya[u] = (aok && kin) ? xa[u] : 0.0f; yb[u] = (bok && kin) ? xb[u] : 0.0f;
        }
        u32x4 fa[3], fb[3];
        frag_parts<NP>(ya, fa); frag_parts<NP>(yb, fb);
        char *d = smem + buf * kBuf + sm_ * XROW + xswz(sm_, koct);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            *reinterpret_cast<u32x4 *>(d + q * kPlane) = fa[q];
            *reinterpret_cast<u32x4 *>(d + (3 + q) * kPlane) = fb[q];
        }
    };
    if (kbeg < kend) {
        fetch(kbeg);
        stash(kbeg, 0);
        __syncthreads();
        int buf = 0;
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            const bool more = k0 + 32 < kend;
            if (more) fetch(k0 + 32);
            const char *A_ = smem + buf * kBuf + (wr * 32 + li) * XROW;
            const char *B_ = smem + buf * kBuf + 3 * kPlane + (wc * 32 + li) * XROW;
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                const int o_ = xswz(li, kg * 2 + hh);
                u32x4 fa[3], fb[3];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    fa[q] = *reinterpret_cast<const u32x4 *>(A_ + o_ + q * kPlane);
                    fb[q] = *reinterpret_cast<const u32x4 *>(B_ + o_ + q * kPlane);
                }
                acc = mfma_parts<NP>(fa, fb, acc);
            }
            if (more) stash(k0 + 32, 1 - buf);
            __syncthreads();
            buf = 1 - buf;
        }
    }
    float *Cs = pr.C + (size_t)slice * pr.c_sk;
    const int n = n0 + wc * 32 + li;
    if (n < pr.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (m < pr.M) Cs[(size_t)m * pr.N + n] = acc[r];
        }
    }
}
struct TnList {
    TnBatch tb; int blocks;
    TnList() { tb.n = 0; blocks = 0; tb.blk0[0] = 0; }
    void add(const float *A, const float *B, float *C, int M, int N, int K, int ks, long c_sk)
    {
        TnProb &p = tb.p[tb.n];
        p = TnProb{A, B, C, M, N, K, ks, c_sk, cdiv(N, 64), cdiv(M, 64) * cdiv(N, 64)};
        blocks += p.tiles * ks; tb.n += 1; tb.blk0[tb.n] = blocks;
    }
    int launch(hipStream_t st, int compute_dtype)
    {
        if (tb.n == 0) return PTX_OK;
        if (compute_dtype == 1) hipLaunchKernelGGL(k_t_tn_gemm<1>, dim3(blocks), dim3(256), 0, st, tb);
        else hipLaunchKernelGGL(k_t_tn_gemm<3>, dim3(blocks), dim3(256), 0, st, tb);
        PTX_LAUNCHED("k_t_tn_gemm");
        return PTX_OK;
    }
};
static int tn_ksplit(int M, int N, int K)
{
    const int tiles = cdiv(M, 64) * cdiv(N, 64);
    // work-groups per product: more slices = more partial traffic (r04 sweep, k_t_tn_gemm + k_t_finalize per block: 256: 64.4 + 19.1 us,
    // 512: 65.8 + 19.5, 1024: 67.8 + 19.4, 2048: 71.3 + 21.6)
    constexpr int target = 512;
    int ks = target / tiles, kb = K / 128;
    if (ks > kb) ks = kb;
    return ks < 1 ? 1 : (ks > 64 ? 64 : ks);
}

// ------------------------------------------------------------------------------ weight transposes of one block, one launch
// out (cols, rows) = in (rows, cols)^T for up to six matrices: the input gradients run as NT products against w^T
struct TrJob { const float *in; float *out; int rows, cols, tiles_x; };
struct TrJobs { TrJob j[6]; int blk0[7]; int n; };
__global__ __launch_bounds__(256) void k_t_transposes(TrJobs J)
{
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.blk0[k + 1]) ++k;
    const TrJob jb = J.j[k];
    const int b = (int)blockIdx.x - J.blk0[k], r0 = (b / jb.tiles_x) * 32, c0 = (b % jb.tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + ty + 8 * r, col = c0 + tx;
        tile[ty + 8 * r][tx] = (row < jb.rows && col < jb.cols) ? jb.in[(size_t)row * jb.cols + col] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = c0 + ty + 8 * r, row = r0 + tx;
        if (col < jb.cols && row < jb.rows) jb.out[(size_t)col * jb.rows + row] = tile[tx][ty + 8 * r];
    }
}
struct TrList {
    TrJobs J; int blocks;
    TrList() { J.n = 0; blocks = 0; J.blk0[0] = 0; }
    void add(const float *in, float *out, int rows, int cols)
    {
        J.j[J.n] = TrJob{in, out, rows, cols, cdiv(cols, 32)};
        blocks += cdiv(rows, 32) * cdiv(cols, 32); J.n += 1; J.blk0[J.n] = blocks;
    }
    int launch(hipStream_t st)
    {
        hipLaunchKernelGGL(k_t_transposes, dim3(blocks), dim3(256), 0, st, J);
        PTX_LAUNCHED("k_t_transposes");
        return PTX_OK;
    }
};

// ------------------------------------------------------------------------------ proxy attention, train mode (PRE:230-252)
// qkv (B*n, 3C) = [Q | K | V] head-split along the columns, pt (B*L, C) the projected proxies.
//   S1[l][i] = scale Pt[l] . K[i]   P1 = softmax_i   D1 = drop(P1)   PV[l] = sum_i D1[l][i] V[i]
//   S2[i][l] = scale Q[i] . Pt[l]   (padded text proxies filled with -1e9)   P2 = softmax_l   D2 = drop(P2)   O[i] = sum_l D2[i][l] PV[l]
// Saved for the backward: P1 (Z,L,n), PV (Z,L,hd), P2 (Z,n,L), Z = B * heads; the dropout masks are recomputed from the seed
// with the element numbering of those arrays.
struct TAttn {
    const float *qkv, *pt; const uint8_t *mask;
    int B, n, L, heads, C; float scale; Drop1 d1, d2;
    float *P1, *PV, *P2, *O;                       // forward outputs (backward: inputs)
    const float *dO; float *dqkv, *dpt;            // backward
    float *dPVp, *dPtp, *dPVg, *dS1g; int ntile;   // backward scratch: per-token-tile partials, final dPV, dS1
};
constexpr int kLT = 4;          // proxies per work-group in the per-proxy kernels
constexpr int kTT = 64;         // tokens per work-group in the per-token kernels

__device__ __forceinline__ float strided_dot(const float *s, const float *g, size_t stride, int i0, int i1);

template <int HD>
__global__ __launch_bounds__(256) void k_tattn_fwd_a(TAttn a)
{
    extern __shared__ float sm[];
    const int n = a.n, L = a.L, C = a.C;
    float *S = sm, *Pt = sm + kLT * n, *red = Pt + kLT * HD;
    const int z = blockIdx.y, b = z / a.heads, h = z - b * a.heads, l0 = blockIdx.x * kLT, tid = threadIdx.x;
    for (int i = tid; i < kLT * HD; i += 256) {
        const int l = l0 + i / HD, d = i % HD;
        Pt[i] = l < L ? a.pt[(size_t)(b * L + l) * C + h * HD + d] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const float4 *kp = reinterpret_cast<const float4 *>(a.qkv + (size_t)(b * n + i) * 3 * C + C + h * HD);
        float acc[kLT];
#pragma unroll
        for (int l = 0; l < kLT; ++l) acc[l] = 0.0f;
#pragma unroll
        for (int q = 0; q < HD / 4; ++q) {
            const float4 kv = kp[q];
#pragma unroll
            for (int l = 0; l < kLT; ++l) {
                const float4 pv = *reinterpret_cast<const float4 *>(&Pt[l * HD + 4 * q]);
                acc[l] = fmaf(kv.x, pv.x, acc[l]); acc[l] = fmaf(kv.y, pv.y, acc[l]);
                acc[l] = fmaf(kv.z, pv.z, acc[l]); acc[l] = fmaf(kv.w, pv.w, acc[l]);
            }
        }
#pragma unroll
        for (int l = 0; l < kLT; ++l) S[l * n + i] = a.scale * acc[l];
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    if (l0 + wv < L) {
        float *sr = S + wv * n;
        float mx = -INFINITY;
        for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sr[i]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int i = lane; i < n; i += 64) { const float e = expf(sr[i] - mx); sr[i] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        const size_t rbase = ((size_t)z * L + l0 + wv) * n;
        for (int i = lane; i < n; i += 64) {
            const float p = sr[i] * inv;
            a.P1[rbase + i] = p;
            sr[i] = drop_apply(a.d1, p, rbase + i);
        }
    }
    __syncthreads();
    constexpr int PARTS = 256 / (kLT * HD) < 1 ? 1 : 256 / (kLT * HD);
    const int o = tid % (kLT * HD), part = tid / (kLT * HD), lw = o / HD, d = o % HD;
    if (part < PARTS) {
        const int per = (n + PARTS - 1) / PARTS, i0 = part * per, i1 = min(n, i0 + per);
        const float *vp = a.qkv + (size_t)(b * n) * 3 * C + 2 * C + h * HD + d;
        red[part * kLT * HD + o] = strided_dot(S + lw * n, vp, (size_t)3 * C, i0, i1);
    }
    __syncthreads();
    if (part == 0 && l0 + lw < L) {
        float v = red[o];
        if (PARTS == 2) v += red[kLT * HD + o];
        a.PV[((size_t)z * L + l0 + lw) * HD + d] = v;
    }
}

// sum_i s[i] * g[i * stride] over [i0, i1): sixteen independent requests in flight per step (one dependent request per step
// made these loops the whole kernel: 345 round trips, r04)
__device__ __forceinline__ float strided_dot(const float *s, const float *g, size_t stride, int i0, int i1)
{
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    for (int i = i0; i < i1; i += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = g[(size_t)min(i + u, i1 - 1) * stride];
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            c0 = fmaf(i + u < i1 ? s[i + u] : 0.0f, v[u], c0);
            c1 = fmaf(i + u + 1 < i1 ? s[i + u + 1] : 0.0f, v[u + 1], c1);
            c2 = fmaf(i + u + 2 < i1 ? s[i + u + 2] : 0.0f, v[u + 2], c2);
            c3 = fmaf(i + u + 3 < i1 ? s[i + u + 3] : 0.0f, v[u + 3], c3);
        }
    }
    return (c0 + c1) + (c2 + c3);
}

// quad (4 adjacent lanes) reductions
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 1)); return fmaxf(v, __shfl_xor(v, 2)); }
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 1); return v + __shfl_xor(v, 2); }

template <int HD>
__global__ __launch_bounds__(256) void k_tattn_fwd_b(TAttn a)
{
    extern __shared__ float sm[];
    const int n = a.n, L = a.L, C = a.C, LP = L + 1;
    float *Pts = sm, *PVs = sm + L * HD, *S2 = PVs + L * HD;            // S2: [kTT][L + 1]
    const int z = blockIdx.y, b = z / a.heads, h = z - b * a.heads, tid = threadIdx.x, tok = tid >> 2, j = tid & 3;
    for (int i = tid; i < L * HD; i += 256) {
        const int l = i / HD, d = i % HD;
        Pts[i] = a.pt[(size_t)(b * L + l) * C + h * HD + d];
        PVs[i] = a.PV[((size_t)z * L + l) * HD + d];
    }
    const int i = blockIdx.x * kTT + tok;
    const bool valid = i < n;
    const int ic = valid ? i : n - 1;
    float q[HD];
    {
        const float4 *qp = reinterpret_cast<const float4 *>(a.qkv + (size_t)(b * n + ic) * 3 * C + h * HD);
#pragma unroll
        for (int t = 0; t < HD / 4; ++t) { const float4 v = qp[t]; q[4 * t] = v.x; q[4 * t + 1] = v.y; q[4 * t + 2] = v.z; q[4 * t + 3] = v.w; }
    }
    __syncthreads();
    float *srow = S2 + tok * LP;
    float mx = -INFINITY;
    for (int l = j; l < L; l += 4) {
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < HD / 4; ++t) {
            const float4 pv = *reinterpret_cast<const float4 *>(&Pts[l * HD + 4 * t]);
            dot = fmaf(q[4 * t], pv.x, dot); dot = fmaf(q[4 * t + 1], pv.y, dot);
            dot = fmaf(q[4 * t + 2], pv.z, dot); dot = fmaf(q[4 * t + 3], pv.w, dot);
        }
        float s = a.scale * dot;
        if (a.mask && a.mask[b * L + l] == 0) s = -1e9f;
        srow[l] = s; mx = fmaxf(mx, s);
    }
    mx = quad_max(mx);
    float sum = 0.0f;
    for (int l = j; l < L; l += 4) sum += expf(srow[l] - mx);
    sum = quad_sum(sum);
    const float inv = 1.0f / sum;
    const size_t rbase = ((size_t)z * n + ic) * L;
    for (int l = j; l < L; l += 4) {
        const float p = expf(srow[l] - mx) * inv;
        if (valid) a.P2[rbase + l] = p;
        srow[l] = drop_apply(a.d2, p, rbase + l);
    }
    __syncthreads();
    constexpr int DQ = HD / 4;
    float o[DQ];
#pragma unroll
    for (int t = 0; t < DQ; ++t) o[t] = 0.0f;
    for (int l = 0; l < L; ++l) {
        const float dv = srow[l];
#pragma unroll
        for (int t = 0; t < DQ; t += 4) {
            const float4 pv = *reinterpret_cast<const float4 *>(&PVs[l * HD + j * DQ + t]);
            o[t] = fmaf(dv, pv.x, o[t]); o[t + 1] = fmaf(dv, pv.y, o[t + 1]);
            o[t + 2] = fmaf(dv, pv.z, o[t + 2]); o[t + 3] = fmaf(dv, pv.w, o[t + 3]);
        }
    }
    if (valid) {
        float4 *op = reinterpret_cast<float4 *>(a.O + (size_t)(b * n + i) * C + h * HD + j * DQ);
#pragma unroll
        for (int t = 0; t < DQ; t += 4) op[t / 4] = make_float4(o[t], o[t + 1], o[t + 2], o[t + 3]);
    }
}

// backward A: per token tile -- dD2 = dO . PV, soft-max backward over the proxies, dQ, and the tile's contributions to
// dPV[l] = sum_i D2[i][l] dO[i] and dPt[l] = sum_i dS2[i][l] Q[i]
template <int HD>
__global__ __launch_bounds__(256) void k_tattn_bwd_a(TAttn a)
{
    extern __shared__ float sm[];
    const int n = a.n, L = a.L, C = a.C, LP = L + 1;
    constexpr int DQ = HD / 4;
    float *Pts = sm, *PVs = Pts + L * HD, *dOs = PVs + L * HD, *Qs = dOs + kTT * HD, *D2s = Qs + kTT * HD, *dSs = D2s + kTT * LP;
    const int z = blockIdx.y, b = z / a.heads, h = z - b * a.heads, tid = threadIdx.x, tok = tid >> 2, j = tid & 3;
    for (int i = tid; i < L * HD; i += 256) {
        const int l = i / HD, d = i % HD;
        Pts[i] = a.pt[(size_t)(b * L + l) * C + h * HD + d];
        PVs[i] = a.PV[((size_t)z * L + l) * HD + d];
    }
    const int i = blockIdx.x * kTT + tok;
    const bool valid = i < n;
    {
        const float4 *gp = reinterpret_cast<const float4 *>(a.dO + (size_t)(b * n + (valid ? i : 0)) * C + h * HD + j * DQ);
        const float4 *qp = reinterpret_cast<const float4 *>(a.qkv + (size_t)(b * n + (valid ? i : 0)) * 3 * C + h * HD + j * DQ);
#pragma unroll
        for (int t = 0; t < DQ / 4; ++t) {
            const float4 gv = valid ? gp[t] : make_float4(0.f, 0.f, 0.f, 0.f), qv = valid ? qp[t] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(&dOs[tok * HD + j * DQ + 4 * t]) = gv;
            *reinterpret_cast<float4 *>(&Qs[tok * HD + j * DQ + 4 * t]) = qv;
        }
    }
    __syncthreads();
    float *d2row = D2s + tok * LP, *dsrow = dSs + tok * LP;
    const size_t rbase = ((size_t)z * n + (valid ? i : 0)) * L;
    float dot = 0.0f;
    for (int l = j; l < L; l += 4) {
        float dd = 0.0f;
#pragma unroll
        for (int t = 0; t < HD / 4; ++t) {
            const float4 g4 = *reinterpret_cast<const float4 *>(&dOs[tok * HD + 4 * t]);
            const float4 pv = *reinterpret_cast<const float4 *>(&PVs[l * HD + 4 * t]);
            dd = fmaf(g4.x, pv.x, dd); dd = fmaf(g4.y, pv.y, dd); dd = fmaf(g4.z, pv.z, dd); dd = fmaf(g4.w, pv.w, dd);
        }
        const float p2 = valid ? a.P2[rbase + l] : 0.0f;
        const float dp = drop_apply(a.d2, dd, rbase + l);
        d2row[l] = p2; dsrow[l] = dp;
        dot = fmaf(p2, dp, dot);
    }
    dot = quad_sum(dot);
    for (int l = j; l < L; l += 4) {
        const float p2 = d2row[l];
        float ds = p2 * (dsrow[l] - dot);
        if (a.mask && a.mask[b * L + l] == 0) ds = 0.0f;
        dsrow[l] = ds;
        d2row[l] = drop_apply(a.d2, p2, rbase + l);
    }
    __syncthreads();
    if (valid) {
        float o[DQ];
#pragma unroll
        for (int t = 0; t < DQ; ++t) o[t] = 0.0f;
        for (int l = 0; l < L; ++l) {
            const float dv = dsrow[l];
#pragma unroll
            for (int t = 0; t < DQ; t += 4) {
                const float4 pv = *reinterpret_cast<const float4 *>(&Pts[l * HD + j * DQ + t]);
                o[t] = fmaf(dv, pv.x, o[t]); o[t + 1] = fmaf(dv, pv.y, o[t + 1]);
                o[t + 2] = fmaf(dv, pv.z, o[t + 2]); o[t + 3] = fmaf(dv, pv.w, o[t + 3]);
            }
        }
        float4 *op = reinterpret_cast<float4 *>(a.dqkv + (size_t)(b * n + i) * 3 * C + h * HD + j * DQ);
#pragma unroll
        for (int t = 0; t < DQ; t += 4) op[t / 4] = make_float4(a.scale * o[t], a.scale * o[t + 1], a.scale * o[t + 2], a.scale * o[t + 3]);
    }
    const size_t pbase = ((size_t)z * a.ntile + blockIdx.x) * L * HD;
    for (int idx = tid; idx < L * HD; idx += 256) {
        const int l = idx / HD, d = idx % HD;
        float c0 = 0.0f, c1 = 0.0f;
#pragma unroll 8
        for (int t = 0; t < kTT; ++t) {
            c0 = fmaf(D2s[t * LP + l], dOs[t * HD + d], c0);
            c1 = fmaf(dSs[t * LP + l], Qs[t * HD + d], c1);
        }
        a.dPVp[pbase + idx] = c0; a.dPtp[pbase + idx] = c1;
    }
}

// backward B: per head, four proxies -- dPV and the proxy-as-key half of dPt from the tile partials (tile order), dD1 = dPV . V,
// soft-max backward over the tokens (dS1 kept for backward C), dPt += scale sum_i dS1[l][i] K[i]
template <int HD>
__global__ __launch_bounds__(256) void k_tattn_bwd_b(TAttn a)
{
    extern __shared__ float sm[];
    const int n = a.n, L = a.L, C = a.C;
    float *S = sm, *dPVs = sm + kLT * n, *red = dPVs + kLT * HD;
    const int z = blockIdx.y, b = z / a.heads, h = z - b * a.heads, l0 = blockIdx.x * kLT, tid = threadIdx.x;
    float dpt_a = 0.0f;
    if (tid < kLT * HD) {
        const int l = l0 + tid / HD, d = tid % HD;
        float pv = 0.0f;
        if (l < L) {
#pragma unroll 4
            for (int t = 0; t < a.ntile; ++t) {
                const size_t pi = (((size_t)z * a.ntile + t) * L + l) * HD + d;
                pv += a.dPVp[pi]; dpt_a += a.dPtp[pi];
            }
            a.dPVg[((size_t)z * L + l) * HD + d] = pv;
        }
        dPVs[tid] = pv;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const float4 *vp = reinterpret_cast<const float4 *>(a.qkv + (size_t)(b * n + i) * 3 * C + 2 * C + h * HD);
        float acc[kLT];
#pragma unroll
        for (int l = 0; l < kLT; ++l) acc[l] = 0.0f;
#pragma unroll
        for (int q = 0; q < HD / 4; ++q) {
            const float4 kv = vp[q];
#pragma unroll
            for (int l = 0; l < kLT; ++l) {
                const float4 pv = *reinterpret_cast<const float4 *>(&dPVs[l * HD + 4 * q]);
                acc[l] = fmaf(kv.x, pv.x, acc[l]); acc[l] = fmaf(kv.y, pv.y, acc[l]);
                acc[l] = fmaf(kv.z, pv.z, acc[l]); acc[l] = fmaf(kv.w, pv.w, acc[l]);
            }
        }
#pragma unroll
        for (int l = 0; l < kLT; ++l)
            if (l0 + l < L) S[l * n + i] = drop_apply(a.d1, acc[l], ((size_t)z * L + l0 + l) * n + i);
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    if (l0 + wv < L) {
        float *sr = S + wv * n;
        const size_t rbase = ((size_t)z * L + l0 + wv) * n;
        float dot = 0.0f;
        for (int i = lane; i < n; i += 64) dot = fmaf(a.P1[rbase + i], sr[i], dot);
        dot = wave_sum(dot);
        for (int i = lane; i < n; i += 64) {
            const float ds = a.P1[rbase + i] * (sr[i] - dot);
            sr[i] = ds; a.dS1g[rbase + i] = ds;
        }
    } else if (wv < kLT) {
        for (int i = lane; i < n; i += 64) S[wv * n + i] = 0.0f;
    }
    __syncthreads();
    constexpr int PARTS = 256 / (kLT * HD) < 1 ? 1 : 256 / (kLT * HD);
    const int o = tid % (kLT * HD), part = tid / (kLT * HD), lw = o / HD, d = o % HD;
    if (part < PARTS) {
        const int per = (n + PARTS - 1) / PARTS, i0 = part * per, i1 = min(n, i0 + per);
        const float *kp = a.qkv + (size_t)(b * n) * 3 * C + C + h * HD + d;
        red[part * kLT * HD + o] = strided_dot(S + lw * n, kp, (size_t)3 * C, i0, i1);
    }
    __syncthreads();
    if (part == 0 && l0 + lw < L) {
        float v = red[o];
        if (PARTS == 2) v += red[kLT * HD + o];
        a.dpt[(size_t)(b * L + l0 + lw) * C + h * HD + d] = a.scale * (dpt_a + v);
    }
}

// backward C: per token tile -- dK[i] = scale sum_l dS1[l][i] Pt[l],  dV[i] = sum_l D1[l][i] dPV[l]
template <int HD>
__global__ __launch_bounds__(256) void k_tattn_bwd_c(TAttn a)
{
    extern __shared__ float sm[];
    const int n = a.n, L = a.L, C = a.C;
    constexpr int DQ = HD / 4, TP = kTT + 1;
    float *Pts = sm, *dPVs = Pts + L * HD, *dSs = dPVs + L * HD, *D1s = dSs + L * TP;
    const int z = blockIdx.y, b = z / a.heads, h = z - b * a.heads, tid = threadIdx.x, tok = tid >> 2, j = tid & 3;
    for (int i = tid; i < L * HD; i += 256) {
        const int l = i / HD, d = i % HD;
        Pts[i] = a.pt[(size_t)(b * L + l) * C + h * HD + d];
        dPVs[i] = a.dPVg[((size_t)z * L + l) * HD + d];
    }
    for (int idx = tid; idx < L * kTT; idx += 256) {
        const int l = idx / kTT, t = idx % kTT, i = blockIdx.x * kTT + t;
        const size_t gi = ((size_t)z * L + l) * n + i;
        const bool ok = i < n;
        dSs[l * TP + t] = ok ? a.dS1g[gi] : 0.0f;
        D1s[l * TP + t] = ok ? drop_apply(a.d1, a.P1[gi], gi) : 0.0f;
    }
    __syncthreads();
    const int i = blockIdx.x * kTT + tok;
    if (i >= n) return;
    float dk[DQ], dv[DQ];
#pragma unroll
    for (int t = 0; t < DQ; ++t) { dk[t] = 0.0f; dv[t] = 0.0f; }
    for (int l = 0; l < L; ++l) {
        const float s1 = dSs[l * TP + tok], d1 = D1s[l * TP + tok];
#pragma unroll
        for (int t = 0; t < DQ; t += 4) {
            const float4 pv = *reinterpret_cast<const float4 *>(&Pts[l * HD + j * DQ + t]);
            const float4 gv = *reinterpret_cast<const float4 *>(&dPVs[l * HD + j * DQ + t]);
            dk[t] = fmaf(s1, pv.x, dk[t]); dk[t + 1] = fmaf(s1, pv.y, dk[t + 1]);
            dk[t + 2] = fmaf(s1, pv.z, dk[t + 2]); dk[t + 3] = fmaf(s1, pv.w, dk[t + 3]);
            dv[t] = fmaf(d1, gv.x, dv[t]); dv[t + 1] = fmaf(d1, gv.y, dv[t + 1]);
            dv[t + 2] = fmaf(d1, gv.z, dv[t + 2]); dv[t + 3] = fmaf(d1, gv.w, dv[t + 3]);
        }
    }
    float *rowp = a.dqkv + (size_t)(b * n + i) * 3 * C + h * HD + j * DQ;
#pragma unroll
    for (int t = 0; t < DQ; t += 4) {
        *reinterpret_cast<float4 *>(rowp + C + t) = make_float4(a.scale * dk[t], a.scale * dk[t + 1], a.scale * dk[t + 2], a.scale * dk[t + 3]);
        *reinterpret_cast<float4 *>(rowp + 2 * C + t) = make_float4(dv[t], dv[t + 1], dv[t + 2], dv[t + 3]);
    }
}

static bool tattn_ok(int B, int n, int L, int heads, int C)
{
    if (B < 1 || n < 1 || L < 1 || heads < 1 || C % heads != 0) return false;
    const int hd = C / heads;
    return (hd == 32 || hd == 64) && (long)L * hd <= 4096 && n <= 8192 && (long)B * heads <= 65535;
}
static size_t tattn_tmp_floats(int B, int n, int L, int heads, int C)
{
    const size_t Z = (size_t)B * heads, hd = C / heads, nt = (n + kTT - 1) / kTT;
    return 2 * Z * nt * L * hd + Z * L * hd + Z * L * n + 64;
}
template <int HD>
static int set_lds(const void *fn, size_t bytes)
{
    if (bytes > 64 * 1024) PTX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PTX_OK;
}
template <int HD>
static int tattn_fwd_t(const TAttn &a, hipStream_t st)
{
    const int Z = a.B * a.heads;
    const size_t lds_a = ((size_t)kLT * a.n + kLT * HD + 2 * kLT * HD) * 4;
    const size_t lds_b = ((size_t)2 * a.L * HD + (size_t)kTT * (a.L + 1)) * 4;
    PTX_TRY(set_lds<HD>(reinterpret_cast<const void *>(&k_tattn_fwd_a<HD>), lds_a));
    PTX_TRY(set_lds<HD>(reinterpret_cast<const void *>(&k_tattn_fwd_b<HD>), lds_b));
    hipLaunchKernelGGL((k_tattn_fwd_a<HD>), dim3(cdiv(a.L, kLT), Z), dim3(256), lds_a, st, a);
    PTX_LAUNCHED("k_tattn_fwd_a");
    hipLaunchKernelGGL((k_tattn_fwd_b<HD>), dim3(cdiv(a.n, kTT), Z), dim3(256), lds_b, st, a);
    PTX_LAUNCHED("k_tattn_fwd_b");
    return PTX_OK;
}
template <int HD>
static int tattn_bwd_t(const TAttn &a, hipStream_t st)
{
    const int Z = a.B * a.heads;
    const size_t lds_a = ((size_t)2 * a.L * HD + 2 * kTT * HD + (size_t)2 * kTT * (a.L + 1)) * 4;
    const size_t lds_b = ((size_t)kLT * a.n + kLT * HD + 2 * kLT * HD) * 4;
    const size_t lds_c = ((size_t)2 * a.L * HD + (size_t)2 * a.L * (kTT + 1)) * 4;
    PTX_TRY(set_lds<HD>(reinterpret_cast<const void *>(&k_tattn_bwd_a<HD>), lds_a));
    PTX_TRY(set_lds<HD>(reinterpret_cast<const void *>(&k_tattn_bwd_b<HD>), lds_b));
    PTX_TRY(set_lds<HD>(reinterpret_cast<const void *>(&k_tattn_bwd_c<HD>), lds_c));
    hipLaunchKernelGGL((k_tattn_bwd_a<HD>), dim3(a.ntile, Z), dim3(256), lds_a, st, a);
    PTX_LAUNCHED("k_tattn_bwd_a");
    hipLaunchKernelGGL((k_tattn_bwd_b<HD>), dim3(cdiv(a.L, kLT), Z), dim3(256), lds_b, st, a);
    PTX_LAUNCHED("k_tattn_bwd_b");
    hipLaunchKernelGGL((k_tattn_bwd_c<HD>), dim3(a.ntile, Z), dim3(256), lds_c, st, a);
    PTX_LAUNCHED("k_tattn_bwd_c");
    return PTX_OK;
}
static TAttn tattn_args(const float *qkv, const float *pt, const uint8_t *mask, int B, int n, int L, int heads, int C, float p,
                        uint64_t seed)
{
    TAttn a;
    memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.pt = pt; a.mask = mask; a.B = B; a.n = n; a.L = L; a.heads = heads; a.C = C;
    a.scale = 1.0f / sqrtf((float)(C / heads));
    a.d1 = make_drop(p, seed); a.d2 = make_drop(p, seed + 1);
    a.ntile = cdiv(n, kTT);
    return a;
}
static void tattn_carve(TAttn &a, float *tmp)
{
    const size_t Z = (size_t)a.B * a.heads, hd = a.C / a.heads;
    a.dPVp = tmp; a.dPtp = a.dPVp + Z * a.ntile * a.L * hd; a.dPVg = a.dPtp + Z * a.ntile * a.L * hd; a.dS1g = a.dPVg + Z * a.L * hd;
}
static int tattn_fwd(const TAttn &a, hipStream_t st) { return a.C / a.heads == 32 ? tattn_fwd_t<32>(a, st) : tattn_fwd_t<64>(a, st); }
static int tattn_bwd(const TAttn &a, hipStream_t st) { return a.C / a.heads == 32 ? tattn_bwd_t<32>(a, st) : tattn_bwd_t<64>(a, st); }

int launch_t_ln_fwd(const float *x, const float *w, const float *b, int R, int C, float eps, float *y, float *stats, hipStream_t st)
{
    PTX_REQUIRE(x && w && b && y && stats && R >= 1 && C % 64 == 0 && C <= 64 * kMaxQ, "layernorm rows: bad arguments (C=%d)", C);
    const Drop1 none = make_drop(0.0f, 0);
    LnFwdArgs g{x, nullptr, none, none, 1, w, b, nullptr, 1, R, C, eps, nullptr, y, stats};
    hipLaunchKernelGGL(k_t_ln_fwd, dim3(cdiv(R, 4)), dim3(256), 0, st, g);
    PTX_LAUNCHED("k_t_ln_fwd");
    return PTX_OK;
}
int t_ln_chunks(int R) { return cdiv(R, kRowsPerChunk); }
int launch_t_ln_bwd(const float *x, const float *stats, const float *w, const float *dy, int R, int C, float *dx, float *part, hipStream_t st)
{
    PTX_REQUIRE(x && stats && w && dy && dx && part && R >= 1 && C % 64 == 0 && C <= 64 * kMaxQ, "layernorm rows backward: bad arguments");
    const Drop1 none = make_drop(0.0f, 0);
    LnBwdArgs g{x, stats, w, dy, nullptr, none, none, 1, R, C, dx, nullptr, part};
    hipLaunchKernelGGL(k_t_ln_bwd<false>, dim3(cdiv(R, kRowsPerChunk)), dim3(256), (size_t)4 * 2 * C * 4, st, g);
    PTX_LAUNCHED("k_t_ln_bwd");
    return PTX_OK;
}

// ------------------------------------------------------------------------------ the block: buffers
struct BlockBufs {
    // saved by the forward
    float *table, *stats1, *xln, *qkv, *pt, *P1, *PV, *P2, *O, *x1, *stats2, *hln, *hpre, *hact, *x2, *stats3, *g, *tpre, *mr;
    size_t save_total;
};
struct Carve {
    float *base; size_t off;
    float *take(size_t n) { float *p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; }
};
static void block_save_layout(const PtxTrainBlock &a, float *base, BlockBufs &s)
{
    const size_t R = (size_t)a.B * a.n, C = a.C, H = a.H, BL = (size_t)a.B * a.L, Z = (size_t)a.B * a.heads, hd = a.C / a.heads;
    Carve c{base, 0};
    s.table = c.take((size_t)a.n * C); s.stats1 = c.take(2 * R); s.xln = c.take(R * C); s.qkv = c.take(R * 3 * C);
    s.pt = c.take(BL * C); s.P1 = c.take(Z * a.L * a.n); s.PV = c.take(Z * a.L * hd); s.P2 = c.take(Z * a.n * a.L);
    s.O = c.take(R * C); s.x1 = c.take(R * C); s.stats2 = c.take(2 * R); s.hln = c.take(R * C); s.hpre = c.take(R * H);
    s.hact = c.take(R * H); s.x2 = c.take(R * C); s.stats3 = c.take(2 * R); s.g = c.take(R * C); s.tpre = c.take(R * a.nout);
    s.mr = c.take(2 * kBnMaxC);
    s.save_total = c.off;
}
struct BwdBufs {
    float *dtpre, *dg, *dx2, *dh2, *dhact, *dhpre, *dhln, *dx1, *dob, *dO, *dqkv, *dpt, *dxln, *attn, *wT[5], *dtab;
    float *p_head, *p_ln3, *p_gelu, *p_ln2, *p_qkv, *p_ln1, *p_bn, *p_w[6];
    int ks[6];
    size_t total;
};
// weight-gradient products in the order qkv, pp, proj, fc1, fc2 (head: its own kernel)
static void block_bwd_layout(const PtxTrainBlock &a, float *base, BwdBufs &t)
{
    const size_t R = (size_t)a.B * a.n, C = a.C, H = a.H, BL = (size_t)a.B * a.L;
    const size_t chunks = (R + kRowsPerChunk - 1) / kRowsPerChunk;
    Carve c{base, 0};
    t.dtpre = c.take(R * a.nout); t.dg = c.take(R * C); t.dx2 = c.take(R * C); t.dh2 = c.take(R * C); t.dhact = c.take(R * H);
    t.dhpre = c.take(R * H); t.dhln = c.take(R * C); t.dx1 = c.take(R * C); t.dob = c.take(R * C); t.dO = c.take(R * C);
    t.dqkv = c.take(R * 3 * C); t.dpt = c.take(BL * C); t.dxln = c.take(R * C);
    t.attn = c.take(tattn_ok(a.B, a.n, a.L, a.heads, a.C) ? tattn_tmp_floats(a.B, a.n, a.L, a.heads, a.C) : 64);
    t.wT[0] = c.take(3 * C * C); t.wT[1] = c.take(C * C); t.wT[2] = c.take(C * C); t.wT[3] = c.take(C * H); t.wT[4] = c.take(C * H);
    t.dtab = c.take((size_t)a.n * C);      // wT: qkv, proxy_proj, proj, fc1, fc2
    t.p_head = c.take(chunks * a.nout * C); t.p_ln3 = c.take(chunks * 3 * C); t.p_gelu = c.take(chunks * H);
    t.p_ln2 = c.take(chunks * 3 * C); t.p_qkv = c.take(chunks * 3 * C); t.p_ln1 = c.take(chunks * 2 * C);
    t.p_bn = c.take((size_t)kBnParts * kBnMaxC);
    const int Ms[5] = {3 * a.C, a.C, a.C, a.H, a.C}, Ns[5] = {a.C, a.C, a.C, a.C, a.H};
    const int Ks[5] = {(int)R, (int)BL, (int)R, (int)R, (int)R};
    for (int i = 0; i < 5; ++i) {
        t.ks[i] = tn_ksplit(Ms[i], Ns[i], Ks[i]);
        t.p_w[i] = t.ks[i] > 1 ? c.take((size_t)t.ks[i] * Ms[i] * Ns[i]) : nullptr;
    }
    t.total = c.off;
}
static size_t block_fwd_tmp(const PtxTrainBlock &a) { return 2 * (((size_t)a.B * a.n * a.C + 63) / 64 * 64) + 64; }

static int block_check(const PtxTrainBlock *a, bool bwd)
{
    PTX_REQUIRE(a, "ptx_train_block: null argument");
    PTX_REQUIRE(a->B >= 1 && a->n >= 1 && a->L >= 1 && a->C % 64 == 0 && a->C <= 512 && a->H % 64 == 0 && a->H <= 2048 &&
                a->heads >= 1 && a->C % a->heads == 0 && a->nout >= 1 && a->nout <= kHeadMax && a->s >= 1,
                "ptx_train_block: shape B=%d n=%d L=%d C=%d H=%d heads=%d nout=%d", a->B, a->n, a->L, a->C, a->H, a->heads, a->nout);
    PTX_REQUIRE(tattn_ok(a->B, a->n, a->L, a->heads, a->C), "ptx_train_block: attention shape L=%d head_dim=%d outside the fused range",
                a->L, a->C / a->heads);
    PTX_REQUIRE(a->x && a->proxy && a->save && a->tmp, "ptx_train_block: null buffer");
    for (int i = 0; i < PTX_TB_NPARAM; ++i)
        PTX_REQUIRE(a->param[i] || i == PTX_TB_QKV_B, "ptx_train_block: parameter %d is null", i);
    PTX_REQUIRE(a->compute_dtype == 0 || a->compute_dtype == 1, "ptx_train_block: compute_dtype=%d", a->compute_dtype);
    PTX_REQUIRE(a->p_attn >= 0.f && a->p_attn < 1.f && a->p_drop >= 0.f && a->p_drop < 1.f && a->p_path >= 0.f && a->p_path < 1.f,
                "ptx_train_block: drop rates");
    if (bwd) PTX_REQUIRE(a->dout && a->dx && a->dproxy, "ptx_train_block_bwd: null gradient buffer");
    else PTX_REQUIRE(a->out, "ptx_train_block_fwd: null output");
    return PTX_OK;
}

static int nt_gemm(const float *x, const float *w, const float *bias, float *y, int rows, int n_out, int n_in, hipStream_t st, int cdt = 0)
{
    GemmBatch g{}; g.n = 1;
    g.p[0] = GemmProb{x, w, y, bias, nullptr, nullptr, nullptr, rows, n_out, n_in, n_in, n_in, n_out, n_out, 0, 0, EPI_NONE};
    return launch_gemm(g, st, cdt);
}
// dx (rows, n_in) = dy (rows, n_out) @ w (n_out, n_in): NT against the transposed weight
static int dx_gemm(const float *dy, const float *wT, float *dx, int rows, int n_out, int n_in, hipStream_t st, int cdt)
{
    return nt_gemm(dy, wT, nullptr, dx, rows, n_in, n_out, st, cdt);
}
// dw (n_out, n_in) = dy^T x over `rows`: a problem of the block's grouped TN launch; K-sliced partials go to the finalize list
static int dw_add(TnList &tn, FinList &fin, const float *dy, const float *x, float *dw, int rows, int n_out, int n_in, int ks, float *part)
{
    PTX_REQUIRE(tn.tb.n < 6, "ptx_train_block_bwd: too many weight-gradient products");
    if (ks > 1) {
        tn.add(dy, x, part, n_out, n_in, rows, ks, (long)n_out * n_in);
        PTX_REQUIRE(fin.add(part, dw, ks, (long)n_out * n_in, (long)n_out * n_in), "ptx_train_block_bwd: job table full");
    } else tn.add(dy, x, dw, n_out, n_in, rows, 1, 0);
    return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

int ptx_train_block_sizes(const PtxTrainBlock *a, size_t *save_floats, size_t *tmp_fwd_floats, size_t *tmp_bwd_floats)
{
    PTX_REQUIRE(a && a->B >= 1 && a->n >= 1 && a->L >= 1 && a->C >= 64 && a->H >= 64 && a->heads >= 1 && a->nout >= 1,
                "ptx_train_block_sizes: bad shape");
    BlockBufs s; BwdBufs t;
    block_save_layout(*a, nullptr, s);
    block_bwd_layout(*a, nullptr, t);
    if (save_floats) *save_floats = s.save_total;
    if (tmp_fwd_floats) *tmp_fwd_floats = block_fwd_tmp(*a);
    if (tmp_bwd_floats) *tmp_bwd_floats = t.total;
    return PTX_OK;
}

int ptx_train_block_fwd(const PtxTrainBlock *ap, void *stream)
{
    PTX_TRY(block_check(ap, false));
    const PtxTrainBlock &a = *ap;
    hipStream_t st = static_cast<hipStream_t>(stream);
    BlockBufs s;
    block_save_layout(a, a.save, s);
    PTX_REQUIRE(a.save_floats >= s.save_total && a.tmp_floats >= block_fwd_tmp(a), "ptx_train_block_fwd: buffers too small");
    const int R = a.B * a.n, C = a.C, H = a.H, BL = a.B * a.L;
    float *o = a.tmp, *h2 = a.tmp + ((size_t)R * C + 63) / 64 * 64;
    const float *const *P = a.param;
    const Drop1 none = make_drop(0.0f, 0);
    // norm1 + per-slot bias table (PRE:212-217)
    PTX_TRY(ptx_op_slotbias_fwd(P[PTX_TB_PB], P[PTX_TB_PC], P[PTX_TB_PR], a.n, a.s, C, s.table, st));
    {
        LnFwdArgs g{a.x, nullptr, none, none, a.n, P[PTX_TB_LN1_W], P[PTX_TB_LN1_B], s.table, a.n, R, C, a.eps1, nullptr, s.xln, s.stats1};
        hipLaunchKernelGGL(k_t_ln_fwd, dim3(cdiv(R, 4)), dim3(256), 0, st, g);
        PTX_LAUNCHED("k_t_ln_fwd");
    }
    PTX_TRY(nt_gemm(s.xln, P[PTX_TB_QKV_W], P[PTX_TB_QKV_B], s.qkv, R, 3 * C, C, st, a.compute_dtype));
    PTX_TRY(nt_gemm(a.proxy, P[PTX_TB_PP_W], P[PTX_TB_PP_B], s.pt, BL, C, C, st, a.compute_dtype));
    {
        TAttn t = tattn_args(s.qkv, s.pt, a.mask, a.B, a.n, a.L, a.heads, C, a.p_attn, a.seed[0]);
        t.P1 = s.P1; t.PV = s.PV; t.P2 = s.P2; t.O = s.O;
        PTX_TRY(tattn_fwd(t, st));
    }
    PTX_TRY(nt_gemm(s.O, P[PTX_TB_PROJ_W], P[PTX_TB_PROJ_B], o, R, C, C, st, a.compute_dtype));
    {   // x1 = x + DropPath(Dropout(o)); hln = norm2(x1)
        LnFwdArgs g{a.x, o, make_drop(a.p_drop, a.seed[1]), make_drop(a.p_path, a.seed[2]), a.n, P[PTX_TB_LN2_W], P[PTX_TB_LN2_B],
                    nullptr, 1, R, C, a.eps2, s.x1, s.hln, s.stats2};
        hipLaunchKernelGGL(k_t_ln_fwd, dim3(cdiv(R, 4)), dim3(256), 0, st, g);
        PTX_LAUNCHED("k_t_ln_fwd");
    }
    PTX_TRY(nt_gemm(s.hln, P[PTX_TB_FC1_W], P[PTX_TB_FC1_B], s.hpre, R, H, C, st, a.compute_dtype));
    {
        const long n4 = (long)R * H / 4;
        hipLaunchKernelGGL(k_t_gelu_drop, dim3((unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256)), dim3(256), 0, st, s.hpre, n4,
                           make_drop(a.p_drop, a.seed[3]), s.hact);
        PTX_LAUNCHED("k_t_gelu_drop");
    }
    PTX_TRY(nt_gemm(s.hact, P[PTX_TB_FC2_W], P[PTX_TB_FC2_B], h2, R, C, H, st, a.compute_dtype));
    {   // x2 = x1 + DropPath(Dropout(h2)); g = trailing LayerNorm(x2)
        LnFwdArgs g{s.x1, h2, make_drop(a.p_drop, a.seed[4]), make_drop(a.p_path, a.seed[5]), a.n, P[PTX_TB_LN3_W], P[PTX_TB_LN3_B],
                    nullptr, 1, R, C, a.eps3, s.x2, s.g, s.stats3};
        hipLaunchKernelGGL(k_t_ln_fwd, dim3(cdiv(R, 4)), dim3(256), 0, st, g);
        PTX_LAUNCHED("k_t_ln_fwd");
    }
    PTX_TRY(nt_gemm(s.g, P[PTX_TB_HEAD_W], P[PTX_TB_HEAD_B], s.tpre, R, a.nout, C, st));
    {
        double *bnp = reinterpret_cast<double *>(h2);            // 2 * kBnParts * kBnMaxC doubles of the free scratch rows
        hipLaunchKernelGGL(k_t_bn_part<0>, dim3(kBnParts), dim3(256), 0, st, s.tpre, (const float *)nullptr, (const float *)nullptr, R, a.nout, bnp);
        hipLaunchKernelGGL(k_t_bn_small_fwd, dim3(kBnParts), dim3(256), 0, st, s.tpre, bnp, kBnParts, R, a.nout, P[PTX_TB_BN_W],
                           P[PTX_TB_BN_B], a.bn_eps, a.bn_momentum, a.bn_run_mean, a.bn_run_var, a.out, s.mr);
    }
    PTX_LAUNCHED("k_t_bn_small_fwd");
    return PTX_OK;
}

int ptx_train_block_bwd(const PtxTrainBlock *ap, void *stream)
{
    PTX_TRY(block_check(ap, true));
    const PtxTrainBlock &a = *ap;
    hipStream_t st = static_cast<hipStream_t>(stream);
    BlockBufs s; BwdBufs t;
    block_save_layout(a, a.save, s);
    block_bwd_layout(a, a.tmp, t);
    PTX_REQUIRE(a.save_floats >= s.save_total && a.tmp_floats >= t.total, "ptx_train_block_bwd: buffers too small");
    const int R = a.B * a.n, C = a.C, H = a.H, BL = a.B * a.L, chunks = cdiv(R, kRowsPerChunk);
    const float *const *P = a.param;
    float *const *G = a.grad;
    for (int i = 0; i < PTX_TB_NPARAM; ++i) PTX_REQUIRE(G[i] || P[i] == nullptr, "ptx_train_block_bwd: gradient buffer %d is null", i);
    const Drop1 none = make_drop(0.0f, 0);
    FinList fin;
    TnList tn;
    {
        TrList tr;
        tr.add(P[PTX_TB_QKV_W], t.wT[0], 3 * C, C); tr.add(P[PTX_TB_PP_W], t.wT[1], C, C); tr.add(P[PTX_TB_PROJ_W], t.wT[2], C, C);
        tr.add(P[PTX_TB_FC1_W], t.wT[3], H, C); tr.add(P[PTX_TB_FC2_W], t.wT[4], C, H);
        PTX_TRY(tr.launch(st));
    }
    // BatchNorm1d + Linear head
    {
        double *bnp = reinterpret_cast<double *>(t.dg);          // free until k_t_head_bwd writes it
        hipLaunchKernelGGL(k_t_bn_part<1>, dim3(kBnParts), dim3(256), 0, st, s.tpre, a.dout, s.mr, R, a.nout, bnp);
        hipLaunchKernelGGL(k_t_bn_small_bwd, dim3(kBnParts), dim3(256), 0, st, s.tpre, s.mr, P[PTX_TB_BN_W], a.dout, bnp, kBnParts, R,
                           a.nout, t.dtpre, G[PTX_TB_BN_W], G[PTX_TB_BN_B], t.p_bn);
        fin.add(t.p_bn, G[PTX_TB_HEAD_B], kBnParts, a.nout, a.nout);
    }
    PTX_LAUNCHED("k_t_bn_small_bwd");
    {
        const size_t lds = ((size_t)4 * a.nout * C + (size_t)a.nout * C) * 4;
        PTX_TRY(set_lds<0>(reinterpret_cast<const void *>(&k_t_head_bwd), lds));
        hipLaunchKernelGGL(k_t_head_bwd, dim3(chunks), dim3(256), lds, st, t.dtpre, (const float *)nullptr, s.g, P[PTX_TB_HEAD_W], R, C, a.nout, t.dg, t.p_head);
        PTX_LAUNCHED("k_t_head_bwd");
        fin.add(t.p_head, G[PTX_TB_HEAD_W], chunks, (long)a.nout * C, (long)a.nout * C);
    }
    // trailing LayerNorm; dh2 = gradient of fc2's output (through DropPath and Dropout)
    {
        LnBwdArgs g{s.x2, s.stats3, P[PTX_TB_LN3_W], t.dg, nullptr, make_drop(a.p_drop, a.seed[4]), make_drop(a.p_path, a.seed[5]), a.n,
                    R, C, t.dx2, t.dh2, t.p_ln3};
        hipLaunchKernelGGL(k_t_ln_bwd<true>, dim3(chunks), dim3(256), (size_t)4 * 3 * C * 4, st, g);
        PTX_LAUNCHED("k_t_ln_bwd");
        fin.add(t.p_ln3, G[PTX_TB_LN3_W], chunks, C, 3l * C);
        fin.add(t.p_ln3 + C, G[PTX_TB_LN3_B], chunks, C, 3l * C);
        fin.add(t.p_ln3 + 2 * C, G[PTX_TB_FC2_B], chunks, C, 3l * C);
    }
    // fc2
    PTX_TRY(dx_gemm(t.dh2, t.wT[4], t.dhact, R, C, H, st, a.compute_dtype));
    PTX_TRY(dw_add(tn, fin, t.dh2, s.hact, G[PTX_TB_FC2_W], R, C, H, t.ks[4], t.p_w[4]));
    // GELU + Dropout
    hipLaunchKernelGGL(k_t_gelu_bwd, dim3(chunks), dim3(256), (size_t)4 * H * 4, st, s.hpre, t.dhact, make_drop(a.p_drop, a.seed[3]), R, H,
                       t.dhpre, t.p_gelu);
    PTX_LAUNCHED("k_t_gelu_bwd");
    fin.add(t.p_gelu, G[PTX_TB_FC1_B], chunks, H, H);
    // fc1
    PTX_TRY(dx_gemm(t.dhpre, t.wT[3], t.dhln, R, H, C, st, a.compute_dtype));
    PTX_TRY(dw_add(tn, fin, t.dhpre, s.hln, G[PTX_TB_FC1_W], R, H, C, t.ks[3], t.p_w[3]));
    // norm2 + residual; dob = gradient of proj's output
    {
        LnBwdArgs g{s.x1, s.stats2, P[PTX_TB_LN2_W], t.dhln, t.dx2, make_drop(a.p_drop, a.seed[1]), make_drop(a.p_path, a.seed[2]), a.n,
                    R, C, t.dx1, t.dob, t.p_ln2};
        hipLaunchKernelGGL(k_t_ln_bwd<true>, dim3(chunks), dim3(256), (size_t)4 * 3 * C * 4, st, g);
        PTX_LAUNCHED("k_t_ln_bwd");
        fin.add(t.p_ln2, G[PTX_TB_LN2_W], chunks, C, 3l * C);
        fin.add(t.p_ln2 + C, G[PTX_TB_LN2_B], chunks, C, 3l * C);
        fin.add(t.p_ln2 + 2 * C, G[PTX_TB_PROJ_B], chunks, C, 3l * C);
    }
    // proj
    PTX_TRY(dx_gemm(t.dob, t.wT[2], t.dO, R, C, C, st, a.compute_dtype));
    PTX_TRY(dw_add(tn, fin, t.dob, s.O, G[PTX_TB_PROJ_W], R, C, C, t.ks[2], t.p_w[2]));
    // proxy attention
    {
        TAttn ta = tattn_args(s.qkv, s.pt, a.mask, a.B, a.n, a.L, a.heads, C, a.p_attn, a.seed[0]);
        ta.P1 = s.P1; ta.PV = s.PV; ta.P2 = s.P2; ta.dO = t.dO; ta.dqkv = t.dqkv; ta.dpt = t.dpt;
        tattn_carve(ta, t.attn);
        PTX_TRY(tattn_bwd(ta, st));
    }
    // proxy_proj
    PTX_TRY(dx_gemm(t.dpt, t.wT[1], a.dproxy, BL, C, C, st, a.compute_dtype));
    PTX_TRY(dw_add(tn, fin, t.dpt, a.proxy, G[PTX_TB_PP_W], BL, C, C, t.ks[1], t.p_w[1]));
    fin.add(t.dpt, G[PTX_TB_PP_B], BL, C, C);
    // qkv
    if (P[PTX_TB_QKV_B]) {
        hipLaunchKernelGGL(k_t_colpart, dim3(chunks), dim3(256), (size_t)4 * 3 * C * 4, st, t.dqkv, R, 3 * C, t.p_qkv);
        PTX_LAUNCHED("k_t_colpart");
        fin.add(t.p_qkv, G[PTX_TB_QKV_B], chunks, 3l * C, 3l * C);
    }
    PTX_TRY(dx_gemm(t.dqkv, t.wT[0], t.dxln, R, 3 * C, C, st, a.compute_dtype));
    PTX_TRY(dw_add(tn, fin, t.dqkv, s.xln, G[PTX_TB_QKV_W], R, 3 * C, C, t.ks[0], t.p_w[0]));
    // norm1 + residual -> dx;  slot-bias table gradient = sum over the scenes of dxln
    {
        LnBwdArgs g{a.x, s.stats1, P[PTX_TB_LN1_W], t.dxln, t.dx1, none, none, a.n, R, C, a.dx, nullptr, t.p_ln1, a.dx_add};
        hipLaunchKernelGGL(k_t_ln_bwd<false>, dim3(chunks), dim3(256), (size_t)4 * 2 * C * 4, st, g);
        PTX_LAUNCHED("k_t_ln_bwd");
        fin.add(t.p_ln1, G[PTX_TB_LN1_W], chunks, C, 2l * C);
        fin.add(t.p_ln1 + C, G[PTX_TB_LN1_B], chunks, C, 2l * C);
    }
    fin.add(t.dxln, t.dtab, a.B, (long)a.n * C, (long)a.n * C);
    PTX_TRY(tn.launch(st, a.compute_dtype));                                  // every weight gradient of the block
    PTX_TRY(fin.launch(st));
    PTX_TRY(ptx_op_slotbias_bwd(t.dtab, a.n, a.s, C, G[PTX_TB_PB], G[PTX_TB_PC], G[PTX_TB_PR], st));
    return PTX_OK;
}

size_t ptx_op_head_bwd_tmp_floats(int R, int C, int nout) { return (size_t)cdiv(R, kRowsPerChunk) * nout * C + 64; }

int ptx_op_head_bwd(const float *dt, const float *coef, const float *x, const float *w, int R, int C, int nout, float *dx, float *dw,
                    float *tmp, size_t tmp_floats, void *stream)
{
    PTX_REQUIRE(dt && x && w && dx && dw && tmp && R >= 1 && C % 64 == 0 && C <= 64 * kMaxQ && nout >= 1 && nout <= kHeadMax &&
                tmp_floats >= ptx_op_head_bwd_tmp_floats(R, C, nout), "ptx_op_head_bwd: bad arguments (R=%d C=%d nout=%d)", R, C, nout);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = cdiv(R, kRowsPerChunk);
    const size_t lds = ((size_t)4 * nout * C + (size_t)nout * C) * 4;
    PTX_TRY(set_lds<0>(reinterpret_cast<const void *>(&k_t_head_bwd), lds));
    hipLaunchKernelGGL(k_t_head_bwd, dim3(chunks), dim3(256), lds, st, dt, coef, x, w, R, C, nout, dx, tmp);
    PTX_LAUNCHED("k_t_head_bwd");
    FinList fin;
    fin.add(tmp, dw, chunks, (long)nout * C, (long)nout * C);
    return fin.launch(st);
}

size_t ptx_train_attn_tmp_floats(int B, int n, int L, int heads, int C)
{
    return tattn_ok(B, n, L, heads, C) ? tattn_tmp_floats(B, n, L, heads, C) : 0;
}

int ptx_train_attn_fwd(const float *qkv, const float *pt, const uint8_t *mask, int B, int n, int L, int heads, int C, float p_drop,
                       uint64_t seed, float *P1, float *PV, float *P2, float *o, void *stream)
{
    PTX_REQUIRE(qkv && pt && P1 && PV && P2 && o && p_drop >= 0.0f && p_drop < 1.0f, "ptx_train_attn_fwd: bad arguments");
    PTX_REQUIRE(tattn_ok(B, n, L, heads, C), "ptx_train_attn_fwd: L=%d head_dim=%d outside the fused range", L, heads ? C / heads : 0);
    TAttn a = tattn_args(qkv, pt, mask, B, n, L, heads, C, p_drop, seed);
    a.P1 = P1; a.PV = PV; a.P2 = P2; a.O = o;
    return tattn_fwd(a, static_cast<hipStream_t>(stream));
}

int ptx_train_attn_bwd(const float *qkv, const float *pt, const uint8_t *mask, int B, int n, int L, int heads, int C, float p_drop,
                       uint64_t seed, const float *P1, const float *PV, const float *P2, const float *dO, float *dqkv, float *dpt,
                       float *tmp, size_t tmp_floats, void *stream)
{
    PTX_REQUIRE(qkv && pt && P1 && PV && P2 && dO && dqkv && dpt && tmp, "ptx_train_attn_bwd: bad arguments");
    PTX_REQUIRE(tattn_ok(B, n, L, heads, C) && tmp_floats >= tattn_tmp_floats(B, n, L, heads, C), "ptx_train_attn_bwd: shape / scratch");
    TAttn a = tattn_args(qkv, pt, mask, B, n, L, heads, C, p_drop, seed);
    a.P1 = const_cast<float *>(P1); a.PV = const_cast<float *>(PV); a.P2 = const_cast<float *>(P2);
    a.dO = dO; a.dqkv = dqkv; a.dpt = dpt;
    tattn_carve(a, tmp);
    return tattn_bwd(a, static_cast<hipStream_t>(stream));
}

}  // extern "C"
