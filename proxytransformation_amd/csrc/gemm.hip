// fp32 matrix-core kernels of the proxy blocks and the folded attention pool.
//
//   k_gemm_nt   C[r][n] = sum_k A[r][k] * W[n][k]  (+bias, GELU, residual, row-scaled addend)
//               grouped: blockIdx.z selects one of up to 8 problems (text / image branch,
//               or the 8 heads of the folded AttentionPool2d tables).
//               v_mfma_f32_32x32x2_f32: exact fp32 (parity config), 64x64 tile per
//               4-wave work-group, each wave one 32x32 accumulator.
//   k_ln_rows   LayerNorm over the channel dim, one wave per row (PRE:275 norm2, PRE:340 norm_img)
//   k_heads     trailing LayerNorm + Linear(C,3|9) + eval BatchNorm1d (PRE:443-446, 452-455)
#include "common.h"

namespace ptx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, LDT = BK + 4;   // 144-B LDS rows: 16-B aligned, b128 fragment reads conflict-free

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// TM x TN output tile per work-group, one 32x32 MFMA accumulator per wave ((TM/32)*(TN/32) waves).
// K loop: BK = 32 per step; the next A/W tiles are fetched into registers while the current
// ones are consumed from LDS, then stored into the other LDS buffer (one barrier per step).
template <int TM, int TN>
__global__ __launch_bounds__((TM / 32) * (TN / 32) * 64) void k_gemm_nt(GemmBatch gb)
{
    constexpr int NT = (TM / 32) * (TN / 32) * 64;     // threads
    constexpr int AV = TM * (BK / 4) / NT;             // float4 per thread per A tile
    constexpr int WV = TN * (BK / 4) / NT;
    const GemmProb pr = gb.p[blockIdx.z];          // by value: fields live in SGPRs, not re-read from kernarg
    const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TN;
    if (row0 >= pr.R || col0 >= pr.N) return;
    __shared__ __attribute__((aligned(16))) float As[2][TM][LDT];
    __shared__ __attribute__((aligned(16))) float Ws[2][TN][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / (TN / 32), wc = wid % (TN / 32);
    const int li = lane & 31, hh = lane >> 5;
    float4 av[AV], wv[WV];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int e = tid + i * NT, r = e >> 3, kq = (e & 7) * 4;
            av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < pr.R && k0 + kq < pr.K)         // K % 4 == 0 is validated by the host
                av[i] = *reinterpret_cast<const float4 *>(pr.A + (size_t)(row0 + r) * pr.lda + k0 + kq);
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int e = tid + i * NT, r = e >> 3, kq = (e & 7) * 4;
            wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col0 + r < pr.N && k0 + kq < pr.K)
                wv[i] = *reinterpret_cast<const float4 *>(pr.W + (size_t)(col0 + r) * pr.ldw + k0 + kq);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int e = tid + i * NT;
            *reinterpret_cast<float4 *>(&As[buf][e >> 3][(e & 7) * 4]) = av[i];
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int e = tid + i * NT;
            *reinterpret_cast<float4 *>(&Ws[buf][e >> 3][(e & 7) * 4]) = wv[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int nk = (pr.K + BK - 1) / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        if (it + 1 < nk) fetch((it + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            // lane half hh owns k = kk*8 + hh*4 .. +3 of this group for both operands, so the
            // two k-values an MFMA consumes (one per half) are consistent between A and B
            const float4 a4 = *reinterpret_cast<const float4 *>(&As[cur][wr * 32 + li][kk * 8 + hh * 4]);
            const float4 b4 = *reinterpret_cast<const float4 *>(&Ws[cur][wc * 32 + li][kk * 8 + hh * 4]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        if (it + 1 < nk) stash(cur ^ 1);
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int n = col0 + wc * 32 + li;
    if (n >= pr.N) return;
    const float bias = pr.bias ? pr.bias[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < pr.R) {
            float v = acc[r] + bias;
            if (pr.epi == EPI_GELU) v = gelu_erf(v);
            if (pr.rs) v = fmaf(pr.rs[(size_t)row * pr.rs_stride], pr.ad[(size_t)row * pr.ldad + n], v);
            if (pr.res) v += pr.res[(size_t)row * pr.ldres + n];
            pr.C[(size_t)row * pr.ldc + n] = v;
        }
    }
}

int launch_gemm(const GemmBatch &gb, hipStream_t st)
{
    PTX_REQUIRE(gb.n >= 1 && gb.n <= kMaxGroups, "gemm: %d groups", gb.n);
    int rmax = 0, nmax = 0;
    long big_tiles = 0;
    for (int g = 0; g < gb.n; ++g) {
        const GemmProb &p = gb.p[g];
        PTX_REQUIRE(p.A && p.W && p.C, "gemm: null operand in group %d", g);
        PTX_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldw % 4 == 0,
                    "gemm: K=%d lda=%d ldw=%d must be multiples of 4", p.K, p.lda, p.ldw);
        PTX_REQUIRE(((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W)) & 15) == 0,
                    "gemm: operands of group %d are not 16-byte aligned", g);
        PTX_REQUIRE(p.rs == nullptr || p.ad != nullptr, "gemm: row scale without addend");
        rmax = p.R > rmax ? p.R : rmax;
        nmax = p.N > nmax ? p.N : nmax;
        big_tiles += (long)cdiv(p.R, 64) * cdiv(p.N, 64);
    }
    if (rmax == 0 || nmax == 0) return PTX_OK;
    // 64x64 tiles re-use each staged operand twice as often; below ~2 work-groups per CU the
    // launch is latency-bound and the 4x finer 32x32 decomposition fills the chip instead
    if (big_tiles >= 512 && nmax > 32) {
        hipLaunchKernelGGL((k_gemm_nt<64, 64>), dim3(cdiv(nmax, 64), cdiv(rmax, 64), gb.n), dim3(256), 0, st, gb);
    } else {
        hipLaunchKernelGGL((k_gemm_nt<32, 32>), dim3(cdiv(nmax, 32), cdiv(rmax, 32), gb.n), dim3(64), 0, st, gb);
    }
    PTX_LAUNCHED("k_gemm_nt");
    return PTX_OK;
}

// ------------------------------------------------------------------------------ LayerNorm rows
// one wave per row; C is a multiple of 64 and at most 512 (8 values per lane)
constexpr int kMaxPerLane = 8;

__device__ __forceinline__ void ln_row(const float *x, int C, float eps, float (&v)[kMaxPerLane],
                                       float &mean, float &rstd)
{
    const int lane = lane_id();
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) {
        const int c = lane + 64 * q;
        v[q] = c < C ? x[c] : 0.0f;
        s += v[q];
    }
    mean = wave_sum(s) / (float)C;
    float var = 0.0f;
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) {
        const int c = lane + 64 * q;
        const float d = c < C ? v[q] - mean : 0.0f;
        var = fmaf(d, d, var);
    }
    rstd = 1.0f / sqrtf(wave_sum(var) / (float)C + eps);
}

__global__ __launch_bounds__(256) void k_ln_rows(LnBatch lb)
{
    const LnProb p = lb.p[blockIdx.y];
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= p.R) return;
    float v[kMaxPerLane], mean, rstd;
    ln_row(p.x + (size_t)row * lb.C, lb.C, lb.eps, v, mean, rstd);
    const int lane = lane_id();
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) {
        const int c = lane + 64 * q;
        if (c < lb.C) {
            float y = (v[q] - mean) * rstd * p.w[c] + p.b[c];
            if (p.add) y += p.add[(size_t)(row % p.add_rows) * lb.C + c];
            p.y[(size_t)row * lb.C + c] = y;
        }
    }
}

int launch_ln_rows(const LnBatch &lb, hipStream_t st)
{
    PTX_REQUIRE(lb.C % 64 == 0 && lb.C <= 64 * kMaxPerLane, "layer norm: C=%d unsupported", lb.C);
    int rmax = 0;
    for (int g = 0; g < lb.n; ++g) rmax = lb.p[g].R > rmax ? lb.p[g].R : rmax;
    if (rmax == 0) return PTX_OK;
    hipLaunchKernelGGL(k_ln_rows, dim3(cdiv(rmax, 4), lb.n), dim3(256), 0, st, lb);
    PTX_LAUNCHED("k_ln_rows");
    return PTX_OK;
}

// ------------------------------------------------------------------------------ heads
__global__ __launch_bounds__(256) void k_heads(HeadBatch hb)
{
    const HeadProb p = hb.p[blockIdx.y];
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= p.R) return;
    float v[kMaxPerLane], mean, rstd;
    ln_row(p.x + (size_t)row * hb.C, hb.C, hb.eps, v, mean, rstd);
    const int lane = lane_id();
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) {
        const int c = lane + 64 * q;
        if (c < hb.C) {
            v[q] = (v[q] - mean) * rstd * p.nw[c] + p.nb[c];
            if (p.guide) p.guide[(size_t)row * hb.C + c] = v[q];
        } else v[q] = 0.0f;
    }
    for (int o = 0; o < p.nout; ++o) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < kMaxPerLane; ++q) {
            const int c = lane + 64 * q;
            if (c < hb.C) s = fmaf(p.hw[(size_t)o * hb.C + c], v[q], s);
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float y = s + p.hb[o];
            p.out[(size_t)row * p.nout + o] = fmaf(y, p.ab[o], p.ab[p.nout + o]);   // eval BatchNorm1d
        }
    }
}

int launch_heads(const HeadBatch &hb, hipStream_t st)
{
    PTX_REQUIRE(hb.C % 64 == 0 && hb.C <= 64 * kMaxPerLane, "heads: C=%d unsupported", hb.C);
    int rmax = 0;
    for (int g = 0; g < hb.n; ++g) rmax = hb.p[g].R > rmax ? hb.p[g].R : rmax;
    if (rmax == 0) return PTX_OK;
    hipLaunchKernelGGL(k_heads, dim3(cdiv(rmax, 4), hb.n), dim3(256), 0, st, hb);
    PTX_LAUNCHED("k_heads");
    return PTX_OK;
}

}  // namespace ptx
