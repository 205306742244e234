// AttentionPool2d in train mode, forward and backward, without materialising the pixel tokens (PRE:144-177, 338).
//
// The reference maps every pixel of every view to a token (Conv2d(in_dim, C, 1)), prepends the mean token, adds the positional
// embedding, projects all tokens to keys and values and attends with ONE query (token 0).  Written out, the training step
// spent 1.3 of its 3.9 ms there (r04: three 7 GFLOP products for the tokens and their two gradients, two 3.5 GFLOP products
// for keys / values and four for their gradients, all through the generic strided kernel).  Because only token 0 queries,
// every one of those products collapses onto `heads` vectors per image (the same algebra as the eval path's folded pool,
// csrc/imgpool.hip, extended to the gradients):
//
//   t0 = wc xbar + bc + pos0,  q = wq t0 + bq,  w'_h = scale wk_h^T q_h  (C),  e'_h = wc^T w'_h  (in_dim)
//   S[h,1+p] = e'_h . x_p + w'_h . (bc + pos_{1+p}) + scale q_h . bk_h,   S[h,0] = w'_h . t0 + scale q_h . bk_h,   P = softmax_t S
//   g_h = sum_t P[h,t] tok_t = wc (sum_p P[h,1+p] x_p) + sum_p P[h,1+p] (bc + pos_{1+p}) + P[h,0] t0,   o_h = wv_h g_h + bv_h
//
// and backwards, with dg_h = wv_h^T do_h:  dP[h,t] = dg_h . tok_t (again a score pass: (wc^T dg_h) . x_p + ...), dS the soft-max
// backward, u_h = sum_t dS[h,t] tok_t (again a pooling pass), dq_h = scale wk_h u_h, dwk_h = scale q_h (x) u_h, dwv_h = do_h (x) g_h,
// dtok_t = sum_h dS[h,t] w'_h + P[h,t] dg_h -- rank 2 * heads per image -- so that
//   dwc = A^T Y,  dpos[1:] = W[:,1:]^T A,  dx_p = sum_j W[j][1+p] (A wc)[j] + wc^T dt0 / hw
// with A = [w' ; dg ; dt0] (rows per (image, head)), W = [dS ; P], Y = [sum_p dS x_p ; sum_p P x_p ; xbar].
// Two streaming passes over the image features forward, two backward plus one write of their gradient; every other product
// has a few hundred rows.  scratch/fold_proto.py checks this algebra in float64 against autograd of the plain form.
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace ptx {

template <int DT>
__device__ __forceinline__ float ti_load(const void *base, size_t off)
{
    if (DT == 0) return static_cast<const float *>(base)[off];
    const unsigned short u = static_cast<const unsigned short *>(base)[off];
    if (DT == 1) return __uint_as_float((unsigned int)u << 16);
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
template <int DT>
__device__ __forceinline__ void ti_store(void *base, size_t off, float v)
{
    if (DT == 0) { static_cast<float *>(base)[off] = v; return; }
    if (DT == 1) {                                           // round to nearest even, like tensor.to(torch.bfloat16)
        unsigned int u = __float_as_uint(v);
        if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
        static_cast<unsigned short *>(base)[off] = (unsigned short)(u >> 16);
        return;
    }
    const _Float16 h = (_Float16)v;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    static_cast<unsigned short *>(base)[off] = u;
}

constexpr int kTiHeads = 8;          // heads per image (PRE:305: num_heads = 8); the kernels below are written for exactly 8
constexpr int kTiSlots = 4;          // pixels per lane in the pooling passes: hw <= 256

// xbar[img][cin] = mean_p x[img][cin][p]; one wave per row
template <int DT>
__global__ __launch_bounds__(256) void k_ti_mean(const void *__restrict__ x, long rows, int hw, float *__restrict__ out)
{
    const long row = blockIdx.x * 4l + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.0f;
    for (int p = lane; p < hw; p += 64) s += ti_load<DT>(x, (size_t)row * hw + p);
    s = wave_sum(s);
    if (lane == 0) out[row] = s / (float)hw;
}

// posb[t] = pos[t] + bc
__global__ void k_ti_posb(const float *__restrict__ pos, const float *__restrict__ bc, int T, int C, float *__restrict__ posb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T * C) posb[i] = pos[i] + bc[i % C];
}

// One image per work-group (512 threads = 8 waves = 8 heads):
//   scores  Z[h][1+p] = V[h] . x_p + zb[h][1+p] (+ zc[h]),  Z[h][0] = A[h] . t0 (+ zc[h])      V (8, Cin), A (8, C) rows of this image
//   MODE 0 (forward):  zc[h] = scale q_h . bk_h;  P = softmax_t Z  -> Wout;   pooled[h] = sum_p P[h][1+p] x_p
//   MODE 1 (backward): Z = dP;  dS = P (dP - sum_t P dP) -> Wout, sig[h] = sum_t dS;  pooled[h] = sum_p dS[h][1+p] x_p
struct TiPool {
    const void *x; int nimg, Cin, hw, C; float scale;
    const float *V, *A, *zb, *t0, *q, *bk, *Pin;
    float *Wout, *pooled, *sig;
};
template <int DT, int MODE>
__global__ __launch_bounds__(512) void k_ti_pool(TiPool a)
{
    extern __shared__ float sm[];
    const int Cin = a.Cin, hw = a.hw, T = hw + 1, C = a.C, img = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    float *Vt = sm;                               // [Cin][8]
    float *Zp = Vt + (size_t)Cin * 8;             // [2][8][256] partial scores
    float *Wl = Zp + 2 * 8 * 256;                 // [8][T]: the weights of the pooling pass
    __shared__ float s_z0[kTiHeads];
    const size_t xbase = (size_t)img * Cin * hw;
    for (int i = tid; i < Cin * 8; i += 512) { const int h = i / Cin, c = i - h * Cin; Vt[c * 8 + h] = a.V[((size_t)img * 8 + h) * Cin + c]; }
    {   // wave h: Z[h][0]
        const float *Ah = a.A + ((size_t)img * 8 + wv) * C, *t0 = a.t0 + (size_t)img * C;
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s = fmaf(Ah[c], t0[c], s);
        if (MODE == 0) {
            const int hd = C / 8;
            float qb = 0.0f;
            for (int d = lane; d < hd; d += 64) qb = fmaf(a.q[(size_t)img * C + wv * hd + d], a.bk[wv * hd + d], qb);
            s = fmaf(a.scale, qb, s);              // both lanes' partial sums carry their share; summed below
        }
        s = wave_sum(s);
        if (lane == 0) s_z0[wv] = s;
    }
    __syncthreads();
    // scores: thread (p, half of the channels)
    {
        const int p = tid & 255, part = tid >> 8, c0 = part * (Cin / 2), c1 = part == 0 ? Cin / 2 : Cin;
        float acc[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) acc[h] = 0.0f;
        if (p < hw) {
            // 16 rows in flight per thread: with one work-group per image on half of the CUs this pass waits for memory, not for issue
            // slots (r05: 8 in flight = 16 KB per CU made the two passes of this kernel 90-100 us for 55 MB)
            for (int c = c0; c < c1; c += 16) {
                float xv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) xv[u] = ti_load<DT>(a.x, xbase + (size_t)min(c + u, c1 - 1) * hw + p);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (c + u < c1) {
                        const float4 v0 = *reinterpret_cast<const float4 *>(&Vt[(c + u) * 8]);
                        const float4 v1 = *reinterpret_cast<const float4 *>(&Vt[(c + u) * 8 + 4]);
                        acc[0] = fmaf(v0.x, xv[u], acc[0]); acc[1] = fmaf(v0.y, xv[u], acc[1]);
                        acc[2] = fmaf(v0.z, xv[u], acc[2]); acc[3] = fmaf(v0.w, xv[u], acc[3]);
                        acc[4] = fmaf(v1.x, xv[u], acc[4]); acc[5] = fmaf(v1.y, xv[u], acc[5]);
                        acc[6] = fmaf(v1.z, xv[u], acc[6]); acc[7] = fmaf(v1.w, xv[u], acc[7]);
                    }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) Zp[(part * 8 + h) * 256 + p] = acc[h];
    }
    __syncthreads();
    // wave h: the row of head h over the T tokens
    {
        const int h = wv;
        const size_t rb = ((size_t)img * 8 + h) * T;
        const float z0 = s_z0[h];
        // with MODE 0 the constant zc[h] is inside z0 already; the pixel scores add it here
        float zc = 0.0f;
        if (MODE == 0) {
            const int hd = C / 8;
            float qb = 0.0f;
            for (int d = lane; d < hd; d += 64) qb = fmaf(a.q[(size_t)img * C + h * hd + d], a.bk[h * hd + d], qb);
            zc = a.scale * wave_sum(qb);
        }
        float *wl = Wl + h * T;
        for (int t = lane; t < T; t += 64)
            wl[t] = t == 0 ? z0 : (Zp[h * 256 + t - 1] + Zp[(8 + h) * 256 + t - 1]) + a.zb[rb + t] + zc;
        if (MODE == 0) {
            float mx = -INFINITY;
            for (int t = lane; t < T; t += 64) mx = fmaxf(mx, wl[t]);
            mx = wave_max(mx);
            float sum = 0.0f;
            for (int t = lane; t < T; t += 64) { const float e = expf(wl[t] - mx); wl[t] = e; sum += e; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int t = lane; t < T; t += 64) { const float pv = wl[t] * inv; wl[t] = pv; a.Wout[rb + t] = pv; }
        } else {
            float dot = 0.0f;
            for (int t = lane; t < T; t += 64) dot = fmaf(a.Pin[rb + t], wl[t], dot);
            dot = wave_sum(dot);
            float sg = 0.0f;
            for (int t = lane; t < T; t += 64) { const float ds = a.Pin[rb + t] * (wl[t] - dot); wl[t] = ds; a.Wout[rb + t] = ds; sg += ds; }
            sg = wave_sum(sg);
            if (lane == 0) a.sig[(size_t)img * 8 + h] = sg;
        }
    }
    __syncthreads();
    // pooling: wave w owns channels 4 w .. 4 w + 3, then + 32, ...; lanes own pixels lane + 64 s.  Four channels per trip = 16 loads
    // in flight per wave, and their 4 x 8 per-lane partial sums are reduced TOGETHER: a halving butterfly (each exchange keeps half of
    // the values: 16 + 8 + 4 + 2 + 1 + 1 = 32 lane exchanges instead of 32 x 6) that leaves value (lane >> 1) complete in both lanes
    // of a pair; pairing order = wave_sum's (32, 16, 8, 4, 2, 1)
    {
        constexpr int UN = 4;
        float wt[kTiSlots][8];
#pragma unroll
        for (int s = 0; s < kTiSlots; ++s)
#pragma unroll
            for (int h = 0; h < 8; ++h) { const int p = lane + 64 * s; wt[s][h] = p < hw ? Wl[h * T + 1 + p] : 0.0f; }
        for (int c = wv * UN; c < Cin; c += 8 * UN) {
            float xv[UN][kTiSlots];
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int s = 0; s < kTiSlots; ++s) {
                    const int p = lane + 64 * s;
                    xv[u][s] = ti_load<DT>(a.x, xbase + (size_t)min(c + u, Cin - 1) * hw + min(p, hw - 1));
                }
            float v[UN * 8];
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    float s = 0.0f;
#pragma unroll
                    for (int sl = 0; sl < kTiSlots; ++sl) s = fmaf(wt[sl][h], xv[u][sl], s);
                    v[u * 8 + h] = s;
                }
            // (written out step by step: with the step count as a loop variable the compiler indexes v[] dynamically -- 2 000 compare /
            // select instructions per trip)
#define PTX_TI_BFLY(n_, bit_)                                                                        \
            {                                                                                        \
                const bool up = (lane & (bit_)) != 0;                                                \
                _Pragma("unroll") for (int i = 0; i < (n_); ++i) {                                   \
                    const float send = up ? v[i] : v[i + (n_)], keep = up ? v[i + (n_)] : v[i];      \
                    v[i] = keep + __shfl_xor(send, (bit_));                                          \
                }                                                                                    \
            }
            static_assert(UN == 4, "the exchange steps below are written for 32 values");
            PTX_TI_BFLY(16, 32) PTX_TI_BFLY(8, 16) PTX_TI_BFLY(4, 8) PTX_TI_BFLY(2, 4) PTX_TI_BFLY(1, 2)
#undef PTX_TI_BFLY
            const float tot = v[0] + __shfl_xor(v[0], 1);
            const int idx = lane >> 1, u = idx >> 3, h = idx & 7;
            if ((lane & 1) == 0 && c + u < Cin) a.pooled[((size_t)img * 8 + h) * Cin + c + u] = tot;
        }
    }
}

// g[k][c] += W[k][0] * t0[img][c]  (k = img * 8 + h);  optionally out[img][c] = bias[c] (the prefill of the product that follows)
__global__ void k_ti_fix(float *__restrict__ g, const float *__restrict__ W, int T, const float *__restrict__ t0, int nimg, int C,
                         float *__restrict__ out, const float *__restrict__ bias)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)nimg * 8 * C) return;
    const long k = i / C; const int c = (int)(i - k * C); const long img = k / 8;
    g[i] = fmaf(W[k * T], t0[img * C + c], g[i]);
    if (out && i < (long)nimg * C) out[i] = bias[i % C];
}

// dt0[img][c] = sum_h dS[k][0] w'[k][c] + P[k][0] dg[k][c];   dbcv[img][c] = sum_h (sig[k] - dS[k][0]) w'[k][c] + (1 - P[k][0]) dg[k][c]
__global__ void k_ti_dt0(const float *__restrict__ dS, const float *__restrict__ P, int T, const float *__restrict__ sig,
                         const float *__restrict__ w, const float *__restrict__ dg, int nimg, int C, float *__restrict__ dt0,
                         float *__restrict__ dbcv)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)nimg * C) return;
    const long img = i / C; const int c = (int)(i - img * C);
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const long k = img * 8 + h;
        const float ds0 = dS[k * T], p0 = P[k * T], wv = w[k * C + c], gv = dg[k * C + c];
        a = fmaf(ds0, wv, a); a = fmaf(p0, gv, a);
        b = fmaf(sig[k] - ds0, wv, b); b = fmaf(1.0f - p0, gv, b);
    }
    dt0[i] = a; dbcv[i] = b;
}

// dbk[h * hd + d] = scale sum_img q[img][h * hd + d] sig[img][h]  (zero up to rounding: the soft-max gradient sums to zero)
__global__ __launch_bounds__(256) void k_ti_dbk(const float *__restrict__ q, const float *__restrict__ sig, int nimg, int C, float scale,
                                                float *__restrict__ dbk)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per column, lanes over the images
    if (c >= C) return;
    const int h = c / (C / 8);
    float s = 0.0f;
    for (int i = lane; i < nimg; i += 64) s = fmaf(q[(size_t)i * C + c], sig[(size_t)i * 8 + h], s);
    s = wave_sum(s);
    if (lane == 0) dbk[c] = scale * s;
}

// dx[img][c][p] = sum_h dS[k][1+p] e'[k][c] + P[k][1+p] Bv[k][c]  +  b0[img][c] / hw;   grid (Cin / 64, nimg), wave: 16 channels
template <int DT>
__global__ __launch_bounds__(256) void k_ti_dx(const float *__restrict__ dS, const float *__restrict__ P, const float *__restrict__ e,
                                               const float *__restrict__ Bv, const float *__restrict__ b0, int Cin, int hw,
                                               void *__restrict__ dx)
{
    const int img = blockIdx.y, T = hw + 1, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float wd[kTiSlots][8], wp[kTiSlots][8];
#pragma unroll
    for (int s = 0; s < kTiSlots; ++s)
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int p = lane + 64 * s;
            const size_t i = ((size_t)img * 8 + h) * T + 1 + min(p, hw - 1);
            wd[s][h] = dS[i]; wp[s][h] = P[i];
        }
    const float inv = 1.0f / (float)hw;
    for (int cc = 0; cc < 16; ++cc) {
        const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 64 + wv * 16 + cc);
        if (c >= Cin) break;
        float ev[8], bv[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) { ev[h] = e[((size_t)img * 8 + h) * Cin + c]; bv[h] = Bv[((size_t)img * 8 + h) * Cin + c]; }
        const float base = b0[(size_t)img * Cin + c] * inv;
#pragma unroll
        for (int s = 0; s < kTiSlots; ++s) {
            const int p = lane + 64 * s;
            float v = base;
#pragma unroll
            for (int h = 0; h < 8; ++h) { v = fmaf(wd[s][h], ev[h], v); v = fmaf(wp[s][h], bv[h], v); }
            if (p < hw) ti_store<DT>(dx, ((size_t)img * Cin + c) * hw + p, v);
        }
    }
}

// out[c] = sum_p part[p * stride + c] in double, p in order (bias gradients over the images, K slices of dwc / dpos)
struct TiSum { const float *part; float *out; int nparts; long ncols, stride; };
struct TiSums { TiSum j[12]; int blk0[13]; int n; };
__global__ __launch_bounds__(256) void k_ti_sums(TiSums J)
{
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.blk0[k + 1]) ++k;
    const TiSum jb = J.j[k];
    const long c = ((long)blockIdx.x - J.blk0[k]) * 256 + threadIdx.x;
    if (c >= jb.ncols) return;
    double t = 0.0;
    int p = 0;
    for (; p + 3 < jb.nparts; p += 4) {
        const float a0 = jb.part[(size_t)p * jb.stride + c], a1 = jb.part[(size_t)(p + 1) * jb.stride + c];
        const float a2 = jb.part[(size_t)(p + 2) * jb.stride + c], a3 = jb.part[(size_t)(p + 3) * jb.stride + c];
        t += (double)a0; t += (double)a1; t += (double)a2; t += (double)a3;
    }
    for (; p < jb.nparts; ++p) t += (double)jb.part[(size_t)p * jb.stride + c];
    jb.out[c] = (float)t;
}
struct TiSumList {
    TiSums J; int blocks;
    TiSumList() { J.n = 0; blocks = 0; J.blk0[0] = 0; }
    void add(const float *part, float *out, int nparts, long ncols, long stride)
    {
        if (!out || J.n >= 12) return;
        J.j[J.n] = TiSum{part, out, nparts, ncols, stride};
        blocks += (int)((ncols + 255) / 256); J.n += 1; J.blk0[J.n] = blocks;
    }
    int launch(hipStream_t st)
    {
        if (J.n == 0) return PTX_OK;
        hipLaunchKernelGGL(k_ti_sums, dim3(blocks), dim3(256), 0, st, J);
        PTX_LAUNCHED("k_ti_sums");
        return PTX_OK;
    }
};

// ---------------------------------------------------------------------------------------------- host side
struct TiBufs {
    float *t0, *q, *g, *posb, *Acat, *Ycat, *Wcat, *Bcat, *sig, *o, *y, *stats;      // save
    size_t save_total;
};
struct TiCarve { float *base; size_t off; float *take(size_t n) { float *p = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return p; } };
static void ti_layout(const PtxTrainImgPool &a, float *base, TiBufs &s)
{
    const size_t nimg = a.nimg, NH = nimg * kTiHeads, C = a.C, Cin = a.Cin, T = a.hw + 1;
    TiCarve c{base, 0};
    s.t0 = c.take(nimg * C); s.q = c.take(nimg * C); s.g = c.take(NH * C); s.posb = c.take(T * C);
    s.Acat = c.take((2 * NH + 2 * nimg) * C);         // [w' ; dg ; dt0 ; dbcv]
    s.Ycat = c.take((2 * NH + nimg) * Cin);           // [sum_p dS x ; sum_p P x ; xbar]
    s.Wcat = c.take(2 * NH * T);                      // [dS ; P]
    s.Bcat = c.take((2 * NH + nimg) * Cin);           // [e' ; wc^T dg ; wc^T dt0]
    s.sig = c.take(NH);
    s.o = c.take(nimg * C); s.y = c.take(nimg * C); s.stats = c.take(2 * nimg);       // the c_proj + norm_img tail
    s.save_total = c.off;
}
static int ti_ksplit(int M, int N, int K)
{
    const int tiles = cdiv(M, 64) * cdiv(N, 64);
    if (K < 512) return 1;
    int ks = 1024 / tiles; if (ks < 1) ks = 1;
    const int kb = K / 128;
    ks = ks < kb ? ks : kb;
    return ks < 1 ? 1 : (ks > 64 ? 64 : ks);
}
static size_t ti_tmp_fwd(const PtxTrainImgPool &a) { return (size_t)a.nimg * kTiHeads * (a.hw + 1) + 64; }
struct TiBwd { float *dps, *u, *dq, *p_wc, *p_pos, *dy, *dotail, *p_ln; int ks_wc, ks_pos; size_t total; };
static void ti_bwd_layout(const PtxTrainImgPool &a, float *base, TiBwd &t)
{
    const size_t nimg = a.nimg, NH = nimg * kTiHeads, C = a.C, Cin = a.Cin, T = a.hw + 1;
    TiCarve c{base, 0};
    t.dps = c.take(NH * T); t.u = c.take(NH * C); t.dq = c.take(nimg * C);
    t.ks_wc = ti_ksplit(a.C, a.Cin, (int)(2 * NH + nimg)); t.ks_pos = ti_ksplit(a.hw, a.C, (int)(2 * NH));
    t.p_wc = c.take((size_t)t.ks_wc * C * Cin); t.p_pos = c.take((size_t)t.ks_pos * a.hw * C);
    t.dy = c.take(nimg * C); t.dotail = c.take(nimg * C); t.p_ln = c.take((size_t)t_ln_chunks((int)nimg) * 2 * C);
    t.total = c.off;
}
static int ti_check(const PtxTrainImgPool *a)
{
    PTX_REQUIRE(a, "ptx_train_imgpool: null argument");
    PTX_REQUIRE(a->nimg >= 1 && a->nimg <= 65535 && a->Cin >= 8 && a->Cin % 8 == 0 && a->Cin <= 2048 && a->hw >= 1 && a->hw <= 256 &&
                a->C % 64 == 0 && a->C <= 512 && a->heads == kTiHeads && a->img_dtype >= 0 && a->img_dtype <= 2,
                "ptx_train_imgpool: shape nimg=%d Cin=%d hw=%d C=%d heads=%d", a->nimg, a->Cin, a->hw, a->C, a->heads);
    PTX_REQUIRE(a->img && a->wc && a->bc && a->pos && a->wq && a->bq && a->wk && a->bk && a->wv && a->bv && a->save && a->tmp,
                "ptx_train_imgpool: null buffer");
    return PTX_OK;
}
// plain strided product C (+)= alpha A B through the generic kernel
struct Bg { const float *A; long a_rs, a_cs, a_s2; const float *B; long b_rs, b_cs, b_s2; float *C; long c_rs, c_cs, c_s2; };
static int bg(const Bg &g, int M, int N, int K, int batch, float alpha, int accumulate, hipStream_t st, int ksplit = 1, long c_sk = 0)
{
    return ptx_op_gemm(g.A, g.B, g.C, M, N, K, g.a_rs, g.a_cs, g.b_rs, g.b_cs, g.c_rs, g.c_cs, batch, batch, 0, g.a_s2, 0, g.b_s2, 0, g.c_s2,
                       0, 0, alpha, accumulate, ksplit, c_sk, st);
}
static int ti_nt(const float *x, const float *w, const float *bias, float *y, int rows, int n_out, int n_in, hipStream_t st)
{
    GemmBatch g{}; g.n = 1;
    g.p[0] = GemmProb{x, w, y, bias, nullptr, nullptr, nullptr, rows, n_out, n_in, n_in, n_in, n_out, n_out, 0, 0, EPI_NONE};
    return launch_gemm(g, st);
}
template <int MODE>
static int ti_pool_launch(const TiPool &p, int dt, hipStream_t st)
{
    const size_t lds = ((size_t)p.Cin * 8 + 2 * 8 * 256 + 8 * (size_t)(p.hw + 1)) * 4;
    const void *fn = dt == 0 ? reinterpret_cast<const void *>(&k_ti_pool<0, MODE>)
                   : dt == 1 ? reinterpret_cast<const void *>(&k_ti_pool<1, MODE>) : reinterpret_cast<const void *>(&k_ti_pool<2, MODE>);
    if (lds > 64 * 1024) PTX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dt == 0) hipLaunchKernelGGL((k_ti_pool<0, MODE>), dim3(p.nimg), dim3(512), lds, st, p);
    else if (dt == 1) hipLaunchKernelGGL((k_ti_pool<1, MODE>), dim3(p.nimg), dim3(512), lds, st, p);
    else hipLaunchKernelGGL((k_ti_pool<2, MODE>), dim3(p.nimg), dim3(512), lds, st, p);
    PTX_LAUNCHED("k_ti_pool");
    return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

int ptx_train_imgpool_sizes(const PtxTrainImgPool *a, size_t *save_floats, size_t *tmp_fwd_floats, size_t *tmp_bwd_floats)
{
    PTX_REQUIRE(a && a->nimg >= 1 && a->Cin >= 1 && a->hw >= 1 && a->C >= 64, "ptx_train_imgpool_sizes: bad shape");
    TiBufs s; TiBwd t;
    ti_layout(*a, nullptr, s);
    ti_bwd_layout(*a, nullptr, t);
    if (save_floats) *save_floats = s.save_total;
    if (tmp_fwd_floats) *tmp_fwd_floats = ti_tmp_fwd(*a);
    if (tmp_bwd_floats) *tmp_bwd_floats = t.total;
    return PTX_OK;
}

int ptx_train_imgpool_fwd(const PtxTrainImgPool *ap, void *stream)
{
    PTX_TRY(ti_check(ap));
    const PtxTrainImgPool &a = *ap;
    const bool tail = a.cw != nullptr;
    PTX_REQUIRE(tail ? (a.cb && a.lnw && a.lnb && a.proxy) : a.o != nullptr, "ptx_train_imgpool_fwd: null output / tail parameter");
    hipStream_t st = static_cast<hipStream_t>(stream);
    TiBufs s;
    ti_layout(a, a.save, s);
    PTX_REQUIRE(a.save_floats >= s.save_total && a.tmp_floats >= ti_tmp_fwd(a), "ptx_train_imgpool_fwd: buffers too small");
    const int nimg = a.nimg, NH = nimg * kTiHeads, C = a.C, Cin = a.Cin, hw = a.hw, T = hw + 1, hd = C / kTiHeads;
    const float scale = 1.0f / sqrtf((float)hd);
    float *w = s.Acat, *Ypool = s.Ycat + (size_t)NH * Cin, *xbar = s.Ycat + (size_t)2 * NH * Cin, *P = s.Wcat + (size_t)NH * T, *e = s.Bcat;
    float *sp = a.tmp;
    float *o = tail ? s.o : a.o;
    {
        const long rows = (long)nimg * Cin;
        const dim3 grid((unsigned)((rows + 3) / 4));
        if (a.img_dtype == 0) hipLaunchKernelGGL(k_ti_mean<0>, grid, dim3(256), 0, st, a.img, rows, hw, xbar);
        else if (a.img_dtype == 1) hipLaunchKernelGGL(k_ti_mean<1>, grid, dim3(256), 0, st, a.img, rows, hw, xbar);
        else hipLaunchKernelGGL(k_ti_mean<2>, grid, dim3(256), 0, st, a.img, rows, hw, xbar);
        PTX_LAUNCHED("k_ti_mean");
    }
    hipLaunchKernelGGL(k_ti_posb, dim3(cdiv(T * C, 256)), dim3(256), 0, st, a.pos, a.bc, T, C, s.posb);
    PTX_LAUNCHED("k_ti_posb");
    PTX_TRY(ti_nt(xbar, a.wc, s.posb, s.t0, nimg, C, Cin, st));                    // t0 = wc xbar + bc + pos0
    PTX_TRY(ti_nt(s.t0, a.wq, a.bq, s.q, nimg, C, C, st));                         // q
    // w'[img][h][:] = scale sum_d q[img][h hd + d] wk[h hd + d][:]
    PTX_TRY(bg(Bg{s.q, C, 1, hd, a.wk, C, 1, (long)hd * C, w, (long)kTiHeads * C, 1, C}, nimg, C, hd, kTiHeads, scale, 0, st));
    PTX_TRY(bg(Bg{w, C, 1, 0, a.wc, Cin, 1, 0, e, Cin, 1, 0}, NH, Cin, C, 1, 1.0f, 0, st));              // e' = w' wc
    PTX_TRY(bg(Bg{w, C, 1, 0, s.posb, 1, C, 0, sp, T, 1, 0}, NH, T, C, 1, 1.0f, 0, st));                 // w' . (bc + pos_t)
    {
        TiPool p{a.img, nimg, Cin, hw, C, scale, e, w, sp, s.t0, s.q, a.bk, nullptr, P, Ypool, nullptr};
        PTX_TRY(ti_pool_launch<0>(p, a.img_dtype, st));
    }
    PTX_TRY(ti_nt(Ypool, a.wc, nullptr, s.g, NH, C, Cin, st));                                            // wc (sum_p P x_p)
    PTX_TRY(bg(Bg{P + 1, T, 1, 0, s.posb + C, C, 1, 0, s.g, C, 1, 0}, NH, C, hw, 1, 1.0f, 1, st));       // + sum_p P (bc + pos)
    hipLaunchKernelGGL(k_ti_fix, dim3(cdiv(NH * C, 256)), dim3(256), 0, st, s.g, P, T, s.t0, nimg, C, o, a.bv);
    PTX_LAUNCHED("k_ti_fix");
    // o[img][h hd + d] = bv + wv[h hd + d] . g[img][h]
    PTX_TRY(bg(Bg{s.g, (long)kTiHeads * C, 1, C, a.wv, 1, C, (long)hd * C, o, C, 1, hd}, nimg, hd, C, kTiHeads, 1.0f, 1, st));
    if (tail) {                                                                  // c_proj, norm_img
        PTX_TRY(ti_nt(o, a.cw, a.cb, s.y, nimg, C, C, st));
        PTX_TRY(launch_t_ln_fwd(s.y, a.lnw, a.lnb, nimg, C, a.ln_eps, a.proxy, s.stats, st));
    }
    return PTX_OK;
}

int ptx_train_imgpool_bwd(const PtxTrainImgPool *ap, void *stream)
{
    PTX_TRY(ti_check(ap));
    const PtxTrainImgPool &a = *ap;
    const bool tail = a.cw != nullptr;
    PTX_REQUIRE((tail ? (a.dproxy && a.dcw && a.dcb && a.dlnw && a.dlnb && a.lnw) : a.dout != nullptr) && a.dwc && a.dbc && a.dpos && a.dwq &&
                a.dbq && a.dwk && a.dbk && a.dwv && a.dbv, "ptx_train_imgpool_bwd: null gradient buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    TiBufs s; TiBwd t;
    ti_layout(a, a.save, s);
    ti_bwd_layout(a, a.tmp, t);
    PTX_REQUIRE(a.save_floats >= s.save_total && a.tmp_floats >= t.total, "ptx_train_imgpool_bwd: buffers too small");
    const int nimg = a.nimg, NH = nimg * kTiHeads, C = a.C, Cin = a.Cin, hw = a.hw, T = hw + 1, hd = C / kTiHeads;
    const float scale = 1.0f / sqrtf((float)hd);
    float *w = s.Acat, *dg = s.Acat + (size_t)NH * C, *dt0 = s.Acat + (size_t)2 * NH * C, *dbcv = dt0 + (size_t)nimg * C;
    float *Yd = s.Ycat, *dS = s.Wcat, *P = s.Wcat + (size_t)NH * T, *e = s.Bcat, *Bv = s.Bcat + (size_t)NH * Cin;
    float *b0 = s.Bcat + (size_t)2 * NH * Cin;
    TiSumList sums;
    const float *dout = a.dout;
    if (tail) {                                                                  // norm_img, c_proj
        PTX_TRY(launch_t_ln_bwd(s.y, s.stats, a.lnw, a.dproxy, nimg, C, t.dy, t.p_ln, st));
        const int ch = t_ln_chunks(nimg);
        sums.add(t.p_ln, a.dlnw, ch, C, 2l * C);
        sums.add(t.p_ln + C, a.dlnb, ch, C, 2l * C);
        sums.add(t.dy, a.dcb, nimg, C, C);
        PTX_TRY(bg(Bg{t.dy, C, 1, 0, a.cw, C, 1, 0, t.dotail, C, 1, 0}, nimg, C, C, 1, 1.0f, 0, st));             // do = dy cw
        PTX_TRY(bg(Bg{t.dy, 1, C, 0, s.o, C, 1, 0, a.dcw, C, 1, 0}, C, C, nimg, 1, 1.0f, 0, st));                  // dcw = dy (x) o
        dout = t.dotail;
    }
    // dg[img][h][:] = sum_d do[img][h hd + d] wv[h hd + d][:];  dwv_h = do_h (x) g_h over the images
    PTX_TRY(bg(Bg{dout, C, 1, hd, a.wv, C, 1, (long)hd * C, dg, (long)kTiHeads * C, 1, C}, nimg, C, hd, kTiHeads, 1.0f, 0, st));
    PTX_TRY(bg(Bg{dout, 1, C, hd, s.g, (long)kTiHeads * C, 1, C, a.dwv, C, 1, (long)hd * C}, hd, C, nimg, kTiHeads, 1.0f, 0, st));
    PTX_TRY(bg(Bg{dg, C, 1, 0, a.wc, Cin, 1, 0, Bv, Cin, 1, 0}, NH, Cin, C, 1, 1.0f, 0, st));             // wc^T dg
    PTX_TRY(bg(Bg{dg, C, 1, 0, s.posb, 1, C, 0, t.dps, T, 1, 0}, NH, T, C, 1, 1.0f, 0, st));              // dg . (bc + pos_t)
    {
        TiPool p{a.img, nimg, Cin, hw, C, scale, Bv, dg, t.dps, s.t0, s.q, a.bk, P, dS, Yd, s.sig};
        PTX_TRY(ti_pool_launch<1>(p, a.img_dtype, st));
    }
    // u_h = sum_t dS[h,t] tok_t
    PTX_TRY(ti_nt(Yd, a.wc, nullptr, t.u, NH, C, Cin, st));
    PTX_TRY(bg(Bg{dS + 1, T, 1, 0, s.posb + C, C, 1, 0, t.u, C, 1, 0}, NH, C, hw, 1, 1.0f, 1, st));
    hipLaunchKernelGGL(k_ti_fix, dim3(cdiv(NH * C, 256)), dim3(256), 0, st, t.u, dS, T, s.t0, nimg, C, (float *)nullptr, (const float *)nullptr);
    PTX_LAUNCHED("k_ti_fix");
    // dq_h = scale wk_h u_h  (the bk_h sum_t dS term is a rounding-level zero and is left out);  dwk_h = scale q_h (x) u_h
    PTX_TRY(bg(Bg{t.u, (long)kTiHeads * C, 1, C, a.wk, 1, C, (long)hd * C, t.dq, C, 1, hd}, nimg, hd, C, kTiHeads, scale, 0, st));
    PTX_TRY(bg(Bg{s.q, 1, C, hd, t.u, (long)kTiHeads * C, 1, C, a.dwk, C, 1, (long)hd * C}, hd, C, nimg, kTiHeads, scale, 0, st));
    hipLaunchKernelGGL(k_ti_dbk, dim3(cdiv(C, 4)), dim3(256), 0, st, s.q, s.sig, nimg, C, scale, a.dbk);
    PTX_LAUNCHED("k_ti_dbk");
    // token 0
    hipLaunchKernelGGL(k_ti_dt0, dim3(cdiv(nimg * C, 256)), dim3(256), 0, st, dS, P, T, s.sig, w, dg, nimg, C, dt0, dbcv);
    PTX_LAUNCHED("k_ti_dt0");
    PTX_TRY(bg(Bg{t.dq, C, 1, 0, a.wq, C, 1, 0, dt0, C, 1, 0}, nimg, C, C, 1, 1.0f, 1, st));              // + wq^T dq
    PTX_TRY(bg(Bg{t.dq, 1, C, 0, s.t0, C, 1, 0, a.dwq, C, 1, 0}, C, C, nimg, 1, 1.0f, 0, st));            // dwq = dq (x) t0
    // gradients of the conv / positional parameters and of the features from the rank-(2 heads + 1) form
    {
        const int K = 2 * NH + nimg;
        if (t.ks_wc > 1) {
            PTX_TRY(bg(Bg{s.Acat, 1, C, 0, s.Ycat, Cin, 1, 0, t.p_wc, Cin, 1, 0}, C, Cin, K, 1, 1.0f, 0, st, t.ks_wc, (long)C * Cin));
            sums.add(t.p_wc, a.dwc, t.ks_wc, (long)C * Cin, (long)C * Cin);
        } else PTX_TRY(bg(Bg{s.Acat, 1, C, 0, s.Ycat, Cin, 1, 0, a.dwc, Cin, 1, 0}, C, Cin, K, 1, 1.0f, 0, st));
    }
    {
        const int K = 2 * NH;
        if (t.ks_pos > 1) {
            PTX_TRY(bg(Bg{s.Wcat + 1, 1, T, 0, s.Acat, C, 1, 0, t.p_pos, C, 1, 0}, hw, C, K, 1, 1.0f, 0, st, t.ks_pos, (long)hw * C));
            sums.add(t.p_pos, a.dpos + C, t.ks_pos, (long)hw * C, (long)hw * C);
        } else PTX_TRY(bg(Bg{s.Wcat + 1, 1, T, 0, s.Acat, C, 1, 0, a.dpos + C, C, 1, 0}, hw, C, K, 1, 1.0f, 0, st));
    }
    sums.add(dt0, a.dpos, nimg, C, C);
    sums.add(dt0, a.dbc, 2 * nimg, C, C);             // dt0 rows followed by the dbcv rows
    sums.add(t.dq, a.dbq, nimg, C, C);
    sums.add(dout, a.dbv, nimg, C, C);
    PTX_TRY(sums.launch(st));
    if (a.dimg) {
        PTX_TRY(bg(Bg{dt0, C, 1, 0, a.wc, Cin, 1, 0, b0, Cin, 1, 0}, nimg, Cin, C, 1, 1.0f, 0, st));      // wc^T dt0
        const dim3 grid(cdiv(Cin, 64), nimg);
        if (a.img_dtype == 0) hipLaunchKernelGGL(k_ti_dx<0>, grid, dim3(256), 0, st, dS, P, e, Bv, b0, Cin, hw, a.dimg);
        else if (a.img_dtype == 1) hipLaunchKernelGGL(k_ti_dx<1>, grid, dim3(256), 0, st, dS, P, e, Bv, b0, Cin, hw, a.dimg);
        else hipLaunchKernelGGL(k_ti_dx<2>, grid, dim3(256), 0, st, dS, P, e, Bv, b0, Cin, hw, a.dimg);
        PTX_LAUNCHED("k_ti_dx");
    }
    return PTX_OK;
}

}  // extern "C"
