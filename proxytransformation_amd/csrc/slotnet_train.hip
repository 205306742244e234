// Train-mode OffsetNetwork / SimplifiedPointNet body (PRE:87-102, 126-140) without materialising the slot activations:
//   slot inputs [rel | p] (6) -> Conv2d(6,C,1) -> BatchNorm2d over ALL B*M*K slots (batch statistics) -> ReLU -> mean / max_K
// r02 ran this as separate passes over the (B*M*K, C) activations (311 040 x 256 floats = 318 MB for the offset network of
// the shipped configuration): GEMM, bias, two column reductions, normalise, pool -- and twice that many on the way back
// (~2.4 of the training step's 10 ms, all of it HBM traffic).  The convolution contracts SIX inputs: recomputing a slot's C
// channels from its 24 input bytes is cheaper than reading them back once, so every pass here recomputes them in registers
// (one wave per cluster, a lane owns channels lane + 64 q, the K slots are broadcast with v_readlane like k_slot_net) and
// only per-channel sums leave the wave:
//   forward   stats (sum)  -> stats (centred squares) -> [k_bn_stats] -> apply + pool            3 passes over 7.5 MB
//   backward  (dbeta, dgamma) -> (dW, db, dcentre)                                               2 passes
// Per-channel sums run over every slot of the batch: each lane accumulates in double, waves are combined in a fixed
// order (work-group partials, then one finalising work-group): deterministic, and as accurate as the column reductions
// they replace.
#include <cstdlib>

#include "common.h"

namespace ptx {

constexpr int kSnBlocks = 256;          // work-groups of the reduction passes (4 waves each, clusters dealt round-robin)

struct SnArgs {
    const float *center, *cluster; long nclus; int K, C;
    const float *conv_w, *conv_b;                   // (C,6), (C)
    const float *bn_w, *bn_b, *mean_rstd;           // (C), (C), (2,C)
    const float *dout; const int32_t *arg;          // backward: (nclus,C) gradient of the pooled output, first-arg-max slots
    const float *dbeta, *dgamma;                    // backward pass 2
    float *out; int32_t *arg_out;                   // forward: pooled (nclus,C), arg-max (max pooling)
    float *dcenter;                                 // backward pass 2: (nclus,3) or null
    double *part;                                   // (kSnBlocks, width) work-group partials
    int maxpool; float inv_total;                   // 1 / (nclus * K)
};

// this lane's slot (k = lane < K): inputs x[6] and the padding flag (PRE:93-99 / 131-137)
__device__ __forceinline__ bool sn_slot(const SnArgs &a, long cl, int lane, float (&x)[6])
{
    bool pad = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = 0.0f;
    if (lane < a.K) {
        const float *pk = a.cluster + (cl * a.K + lane) * 3, *c = a.center + cl * 3;
        const float px = pk[0], py = pk[1], pz = pk[2];
        pad = px == 0.0f && py == 0.0f && pz == 0.0f;
        x[0] = pad ? 0.0f : px - c[0]; x[1] = pad ? 0.0f : py - c[1]; x[2] = pad ? 0.0f : pz - c[2];
        x[3] = px; x[4] = py; x[5] = pz;
    }
    return pad;
}

template <int Q>
struct SnWeights { float wt[Q][6], bs[Q]; };
template <int Q>
__device__ __forceinline__ void sn_load_weights(const SnArgs &a, int lane, SnWeights<Q> &w)
{
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c = lane + 64 * q;
#pragma unroll
        for (int i = 0; i < 6; ++i) w.wt[q][i] = a.conv_w[c * 6 + i];
        w.bs[q] = a.conv_b[c];
    }
}
// h[q] of slot k: the six inputs come from lane k
template <int Q>
__device__ __forceinline__ void sn_conv(const SnWeights<Q> &w, const float (&x)[6], int k, float (&s)[6], float (&h)[Q])
{
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[i]), k));
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        float v = w.bs[q];
#pragma unroll
        for (int i = 0; i < 6; ++i) v = fmaf(w.wt[q][i], s[i], v);
        h[q] = v;
    }
}

// work-group partials: part[block][j * C + channel] for the NV per-channel accumulators acc[j][q]
template <int Q, int NV>
__device__ __forceinline__ void sn_store_partials(const SnArgs &a, double (&acc)[NV][Q], double *lds)
{
    const int lane = lane_id(), wv = threadIdx.x >> 6, C = 64 * Q;
    // lds: [4 waves][NV * C] doubles
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) lds[(size_t)wv * NV * C + j * C + lane + 64 * q] = acc[j][q];
    __syncthreads();
    for (int i = threadIdx.x; i < NV * C; i += 256)
        a.part[(size_t)blockIdx.x * NV * C + i] = ((lds[i] + lds[(size_t)NV * C + i]) + lds[(size_t)2 * NV * C + i]) + lds[(size_t)3 * NV * C + i];
}

// sum over the work-group partials of output i: 16 lanes take the blocks round-robin (16 loads in flight per output instead
// of one dependent chain of 256), combined in lane order -- fixed order
__device__ __forceinline__ double sn_finish_sum(const double *__restrict__ part, int nblocks, int width, int i)
{
    __shared__ double red[16][17];
    const int il = threadIdx.x & 15, sub = threadIdx.x >> 4;
    double t = 0.0;
    if (i < width)
        for (int b = sub; b < nblocks; b += 16) t += part[(size_t)b * width + i];
    red[sub][il] = t;
    __syncthreads();
    double s = 0.0;
    if (sub == 0)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][il];
    return s;
}
// out[i] = scale * sum over the work-group partials
__global__ __launch_bounds__(256) void k_sn_finish(const double *__restrict__ part, int nblocks, int width, float scale,
                                                   float *__restrict__ out)
{
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s = sn_finish_sum(part, nblocks, width, i);
    if ((threadIdx.x >> 4) == 0 && i < width) out[i] = (float)(s * (double)scale);
}

// Batch statistics of the convolution's output WITHOUT evaluating it: h_c = w_c . x + b_c is linear in the six slot inputs, so
//   mean_c = w_c . mu + b_c,   var_c = w_c^T Cov w_c      with mu = E[x] (6), Cov = E[x x^T] - mu mu^T (6 x 6) over all slots
// (padded slots and their zero inputs included, as BatchNorm2d sees them).  One pass over the 24-byte slots accumulating the
// 6 + 21 moments in double replaces the two passes that recomputed all C channels of every slot (r04: 2 x 30 us + 2 x 15 us
// and four finishing launches per step); k_sn_moments_finish turns them into mean / rstd and updates the running statistics.
constexpr int kSnMom = 27;
__global__ __launch_bounds__(256) void k_sn_moments(SnArgs a)
{
    __shared__ double red[4][kSnMom];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    double acc[kSnMom];
#pragma unroll
    for (int i = 0; i < kSnMom; ++i) acc[i] = 0.0;
    for (long cl = gw; cl < a.nclus; cl += (long)gridDim.x * 4) {
        float x[6];
        sn_slot(a, cl, lane, x);
        int t = 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            acc[i] += (double)x[i];
#pragma unroll
            for (int j = i; j < 6; ++j) { acc[t] += (double)x[i] * (double)x[j]; ++t; }
        }
    }
#pragma unroll
    for (int i = 0; i < kSnMom; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kSnMom) a.part[(size_t)blockIdx.x * kSnMom + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
__global__ __launch_bounds__(512) void k_sn_moments_finish(const double *__restrict__ part, int nblocks, const float *__restrict__ conv_w,
                                                           const float *__restrict__ conv_b, int C, double total, float eps, float momentum,
                                                           float *__restrict__ mean_rstd, float *__restrict__ run_mean, float *__restrict__ run_var)
{
    __shared__ double mom[kSnMom];
    __shared__ double red[16][32];
    {   // 16 interleaved slices of the work-group partials per moment, combined in slice order
        const int i = threadIdx.x & 31, sl = threadIdx.x >> 5;
        double t = 0.0;
        if (i < kSnMom)
            for (int b = sl; b < nblocks; b += 16) t += part[(size_t)b * kSnMom + i];
        red[sl][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < kSnMom) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
        mom[threadIdx.x] = t / total;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 512) {
        double w[6], mean = (double)conv_b[c], var = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { w[i] = (double)conv_w[c * 6 + i]; mean += w[i] * mom[i]; }
        int t = 6;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) { const double cov = mom[t] - mom[i] * mom[j]; var += (i == j ? 1.0 : 2.0) * w[i] * w[j] * cov; ++t; }
        const float vf = var > 0.0 ? (float)var : 0.0f;                   // biased: what normalises (PRE:74, 114)
        mean_rstd[c] = (float)mean;
        mean_rstd[C + c] = 1.0f / sqrtf(vf + eps);
        if (run_mean) {
            const float unb = total > 1.0 ? (float)(var * total / (total - 1.0)) : vf;
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
        }
    }
}

// y = relu(bn(h)); pooled over the K slots (mean, or max with the FIRST arg-max like torch.max)
template <int Q>
__global__ __launch_bounds__(256) void k_sn_apply(SnArgs a)
{
    const int lane = lane_id(), C = 64 * Q;
    const long cl = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (cl >= a.nclus) return;
    SnWeights<Q> w;
    sn_load_weights<Q>(a, lane, w);
    float mu[Q], rs[Q], ga[Q], be[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c = lane + 64 * q;
        mu[q] = a.mean_rstd[c]; rs[q] = a.mean_rstd[C + c]; ga[q] = a.bn_w[c]; be[q] = a.bn_b[c];
    }
    float x[6], s[6], h[Q], best[Q];
    int bi[Q];
    sn_slot(a, cl, lane, x);
#pragma unroll
    for (int q = 0; q < Q; ++q) { best[q] = a.maxpool ? -INFINITY : 0.0f; bi[q] = 0; }
    for (int k = 0; k < a.K; ++k) {
        sn_conv<Q>(w, x, k, s, h);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float y = fmaxf((h[q] - mu[q]) * rs[q] * ga[q] + be[q], 0.0f);
            if (a.maxpool) { if (y > best[q]) { best[q] = y; bi[q] = k; } }
            else best[q] += y;
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c = lane + 64 * q;
        a.out[cl * C + c] = a.maxpool ? best[q] : best[q] / (float)a.K;
        if (a.maxpool) a.arg_out[cl * C + c] = bi[q];
    }
}

// gradient reaching slot k's activation: dy through the pooling, then through the ReLU
template <int Q>
__device__ __forceinline__ void sn_slot_grad(const SnArgs &a, int k, const float (&h)[Q], const float (&mu)[Q], const float (&rs)[Q],
                                             const float (&ga)[Q], const float (&be)[Q], const float (&dout)[Q],
                                             const int (&am)[Q], float (&g)[Q], float (&xh)[Q])
{
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        xh[q] = (h[q] - mu[q]) * rs[q];
        const float y = (h[q] - mu[q]) * rs[q] * ga[q] + be[q];
        const float dy = a.maxpool ? (am[q] == k ? dout[q] : 0.0f) : dout[q] / (float)a.K;
        g[q] = y > 0.0f ? dy : 0.0f;
    }
}

// backward pass 1: dbeta = sum g, dgamma = sum g * xhat
template <int Q>
__global__ __launch_bounds__(256) void k_sn_bwd_stats(SnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sn_lds[];
    const int lane = lane_id(), C = 64 * Q;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    SnWeights<Q> w;
    sn_load_weights<Q>(a, lane, w);
    float mu[Q], rs[Q], ga[Q], be[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c = lane + 64 * q;
        mu[q] = a.mean_rstd[c]; rs[q] = a.mean_rstd[C + c]; ga[q] = a.bn_w[c]; be[q] = a.bn_b[c];
    }
    double acc[2][Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { acc[0][q] = 0.0; acc[1][q] = 0.0; }
    for (long cl = gw; cl < a.nclus; cl += (long)gridDim.x * 4) {
        float x[6], s[6], h[Q], g[Q], xh[Q], dout[Q], l0[Q], l1[Q];
        int am[Q];
        sn_slot(a, cl, lane, x);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            dout[q] = a.dout[cl * C + lane + 64 * q];
            am[q] = a.maxpool ? a.arg[cl * C + lane + 64 * q] : 0;
            l0[q] = 0.0f; l1[q] = 0.0f;
        }
        for (int k = 0; k < a.K; ++k) {
            sn_conv<Q>(w, x, k, s, h);
            sn_slot_grad<Q>(a, k, h, mu, rs, ga, be, dout, am, g, xh);
#pragma unroll
            for (int q = 0; q < Q; ++q) { l0[q] += g[q]; l1[q] = fmaf(g[q], xh[q], l1[q]); }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) { acc[0][q] += (double)l0[q]; acc[1][q] += (double)l1[q]; }
    }
    sn_store_partials<Q, 2>(a, acc, sn_lds);
}

// backward pass 2: dh = gamma rstd (g - dbeta / R - xhat dgamma / R);  db = sum dh, dW[c][i] = sum dh x_i;
// dcentre = - sum over non-padded slots of W[:, 0..2]^T dh  (rel = p - c; the points carry no gradient)
template <int Q>
__global__ __launch_bounds__(256) void k_sn_bwd_dx(SnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sn_lds[];
    const int lane = lane_id(), C = 64 * Q;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    SnWeights<Q> w;
    sn_load_weights<Q>(a, lane, w);
    float mu[Q], rs[Q], ga[Q], be[Q], db_r[Q], dg_r[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c = lane + 64 * q;
        mu[q] = a.mean_rstd[c]; rs[q] = a.mean_rstd[C + c]; ga[q] = a.bn_w[c]; be[q] = a.bn_b[c];
        db_r[q] = a.dbeta[c] * a.inv_total; dg_r[q] = a.dgamma[c] * a.inv_total;
    }
    double acc[7][Q];           // [0]: db, [1 + i]: dW[:, i]
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[j][q] = 0.0;
    for (long cl = gw; cl < a.nclus; cl += (long)gridDim.x * 4) {
        float x[6], s[6], h[Q], g[Q], xh[Q], dout[Q], loc[7][Q], tc[Q];
        int am[Q];
        const bool pad = sn_slot(a, cl, lane, x);
        const unsigned long long padmask = __ballot(pad);           // bit k: slot k is padding (lanes >= K count as padding)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            dout[q] = a.dout[cl * C + lane + 64 * q];
            am[q] = a.maxpool ? a.arg[cl * C + lane + 64 * q] : 0;
            tc[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < 7; ++j) loc[j][q] = 0.0f;
        }
        for (int k = 0; k < a.K; ++k) {
            sn_conv<Q>(w, x, k, s, h);
            sn_slot_grad<Q>(a, k, h, mu, rs, ga, be, dout, am, g, xh);
            const bool real = ((padmask >> k) & 1ull) == 0ull;     // wave-uniform
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float dh = ga[q] * rs[q] * (g[q] - db_r[q] - xh[q] * dg_r[q]);
                loc[0][q] += dh;
#pragma unroll
                for (int i = 0; i < 6; ++i) loc[1 + i][q] = fmaf(dh, s[i], loc[1 + i][q]);
                if (real) tc[q] += dh;
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[j][q] += (double)loc[j][q];
        if (a.dcenter != nullptr) {
            float d[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int i = 0; i < 3; ++i) d[i] = fmaf(w.wt[q][i], tc[q], d[i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) d[i] = wave_sum(d[i]);
            if (lane < 3) a.dcenter[cl * 3 + lane] = -(lane == 0 ? d[0] : lane == 1 ? d[1] : d[2]);
        }
    }
    sn_store_partials<Q, 7>(a, acc, sn_lds);
}

// dW (C,6) / db (C) from the partials [block][j * C + c]
__global__ __launch_bounds__(256) void k_sn_finish_w(const double *__restrict__ part, int nblocks, int C, float *__restrict__ dw,
                                                     float *__restrict__ db)
{
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s = sn_finish_sum(part, nblocks, 7 * C, i);
    if ((threadIdx.x >> 4) != 0 || i >= 7 * C) return;
    const int j = i / C, c = i - j * C;
    if (j == 0) db[c] = (float)s;
    else dw[c * 6 + (j - 1)] = (float)s;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

size_t ptx_op_slotnet_scratch_bytes(int C) { return (size_t)kSnBlocks * 7 * (size_t)C * sizeof(double); }

int ptx_op_slotnet_fwd(const float *center, const float *cluster, long nclus, int K, int C, const float *conv_w,
                       const float *conv_b, const float *bn_w, const float *bn_b, float eps, float momentum, float *run_mean,
                       float *run_var, int maxpool, float *out, int32_t *arg, float *mean_rstd, float *stat_tmp, void *scratch,
                       size_t scratch_bytes, void *stream)
{
    PTX_REQUIRE(center && cluster && conv_w && conv_b && bn_w && bn_b && out && mean_rstd && stat_tmp && scratch &&
                (!maxpool || arg), "ptx_op_slotnet_fwd: null argument");
    PTX_REQUIRE(nclus >= 1 && K >= 1 && K <= 63 && (C == 256 || C == 512), "ptx_op_slotnet_fwd: nclus=%ld K=%d C=%d", nclus, K, C);
    PTX_REQUIRE(scratch_bytes >= ptx_op_slotnet_scratch_bytes(C), "ptx_op_slotnet_fwd: scratch too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    SnArgs a{};
    a.center = center; a.cluster = cluster; a.nclus = nclus; a.K = K; a.C = C; a.conv_w = conv_w; a.conv_b = conv_b;
    a.bn_w = bn_w; a.bn_b = bn_b; a.mean_rstd = mean_rstd; a.out = out; a.arg_out = arg; a.part = static_cast<double *>(scratch);
    a.maxpool = maxpool; a.inv_total = 1.0f / (float)((double)nclus * K);
    // (r03 took both statistics from the C channels in two passes over the slots: 2 x 30 us + finishing launches; r04: the 6 + 21
    //  moments of the slot inputs in one pass, train.py / DESIGN.md 5.4)
    (void)stat_tmp;
    hipLaunchKernelGGL(k_sn_moments, dim3(kSnBlocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_sn_moments_finish, dim3(1), dim3(512), 0, st, a.part, kSnBlocks, conv_w, conv_b, C, (double)nclus * K, eps,
                       momentum, mean_rstd, run_mean, run_var);
    PTX_LAUNCHED("k_sn_moments");
    if (C == 256) hipLaunchKernelGGL(k_sn_apply<4>, dim3((unsigned)((nclus + 3) / 4)), dim3(256), 0, st, a);
    else          hipLaunchKernelGGL(k_sn_apply<8>, dim3((unsigned)((nclus + 3) / 4)), dim3(256), 0, st, a);
    PTX_LAUNCHED("k_sn_apply");
    return PTX_OK;
}

int ptx_op_slotnet_bwd(const float *center, const float *cluster, long nclus, int K, int C, const float *conv_w,
                       const float *conv_b, const float *bn_w, const float *bn_b, const float *mean_rstd, int maxpool,
                       const int32_t *arg, const float *dout, float *dconv_w, float *dconv_b, float *dbeta_dgamma,
                       float *dcenter, void *scratch, size_t scratch_bytes, void *stream)
{
    PTX_REQUIRE(center && cluster && conv_w && conv_b && bn_w && bn_b && mean_rstd && dout && dconv_w && dconv_b &&
                dbeta_dgamma && scratch && (!maxpool || arg), "ptx_op_slotnet_bwd: null argument");
    float *dbeta = dbeta_dgamma, *dgamma = dbeta_dgamma + C;        // (2,C): the two reductions are finished by one launch
    PTX_REQUIRE(nclus >= 1 && K >= 1 && K <= 63 && (C == 256 || C == 512), "ptx_op_slotnet_bwd: nclus=%ld K=%d C=%d", nclus, K, C);
    PTX_REQUIRE(scratch_bytes >= ptx_op_slotnet_scratch_bytes(C), "ptx_op_slotnet_bwd: scratch too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    SnArgs a{};
    a.center = center; a.cluster = cluster; a.nclus = nclus; a.K = K; a.C = C; a.conv_w = conv_w; a.conv_b = conv_b;
    a.bn_w = bn_w; a.bn_b = bn_b; a.mean_rstd = mean_rstd; a.dout = dout; a.arg = arg; a.part = static_cast<double *>(scratch);
    a.maxpool = maxpool; a.inv_total = 1.0f / (float)((double)nclus * K); a.dcenter = dcenter;
    const size_t lds2 = (size_t)4 * 2 * C * sizeof(double), lds7 = (size_t)4 * 7 * C * sizeof(double);
    if (C == 256) hipLaunchKernelGGL(k_sn_bwd_stats<4>, dim3(kSnBlocks), dim3(256), lds2, st, a);
    else          hipLaunchKernelGGL(k_sn_bwd_stats<8>, dim3(kSnBlocks), dim3(256), lds2, st, a);
    // partial layout [block][j * C + c]: j = 0 -> dbeta, j = 1 -> dgamma
    hipLaunchKernelGGL(k_sn_finish, dim3(cdiv(2 * C, 16)), dim3(256), 0, st, a.part, kSnBlocks, 2 * C, 1.0f, dbeta);
    PTX_LAUNCHED("k_sn_bwd_stats");
    a.dbeta = dbeta; a.dgamma = dgamma;
    if (lds7 > 64 * 1024) {
        if (C == 256) PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_bwd_dx<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds7));
        else          PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sn_bwd_dx<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds7));
    }
    if (C == 256) hipLaunchKernelGGL(k_sn_bwd_dx<4>, dim3(kSnBlocks), dim3(256), lds7, st, a);
    else          hipLaunchKernelGGL(k_sn_bwd_dx<8>, dim3(kSnBlocks), dim3(256), lds7, st, a);
    hipLaunchKernelGGL(k_sn_finish_w, dim3(cdiv(7 * C, 16)), dim3(256), 0, st, a.part, kSnBlocks, C, dconv_w, dconv_b);
    PTX_LAUNCHED("k_sn_bwd_dx");
    return PTX_OK;
}

}  // extern "C"
