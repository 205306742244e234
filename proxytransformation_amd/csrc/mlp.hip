// The Mlp of a ProxyBlock in ONE launch for calls with few scenes (timm Mlp, PRE:275-276):
//     x2 = x1 + fc2(GELU(fc1(norm2(x1))))                              x1 (R, 256), hidden 1024
// As two GEMM launches (k_gemm64x with the LayerNorm fold + GELU epilogue, then k_gemm32 over K = 1024) this was 17 + 19 us of
// the 4-scene step for 2 GFLOP: each launch is a prologue of cold round trips, eight K steps and an epilogue, and the hidden
// activations (8 MB) make a round trip through memory in between (r03 stamps, gemm.hip).  Here a work-group owns 32 rows
// of one branch and a SLICE of 256 hidden units:
//   1. the x1 tile is split once into three bf16 planes in LDS (A operand of every MFMA of phase 1)
//   2. phase 1   H = GELU(rstd (x1 Wg^T - mu s) + c)  for its 256 hidden units: eight waves, one 32 x 32 tile each, K = 256
//   3. H goes back to LDS as split planes (A operand of phase 2) -- it never leaves the CU
//   4. phase 2   P = H W2[:, slice]^T  (32 x 256, K = the slice): a partial sum of fc2 over the hidden units
//   5. the four slices of a row tile leave their partials as agent-scope stores; the last one to arrive (ticket) adds them
//      in slice order with the bias and the residual, writes x2 (the pattern of the split attention, fattn.hip) and applies
//      the block's output head to the finished rows (k_heads is no launch of its own).
// 32 row tiles x 4 slices x 2 branches = 256 work-groups at the benchmark shape.  The WEIGHTS are parameter-only: ptx_prepare
// stores them already split into bf16 planes in the order the MFMA B fragments are read (k_prep_planes), so a lane's
// fragment is one coalesced 16-byte load straight from L2 -- no staging through LDS, no split arithmetic in the loop.
// Every product is the six-term split product of split3.h (fp32 in another summation order).
#include <cstdlib>

#include "common.h"
#include "split3.h"

namespace ptx {

constexpr int kMlpRows = 32, kMlpSlice = 256, kMlpWaves = 8;
constexpr int kMlpPlane = 16 * 2 * 32 * 16;                 // one bf16 plane of a 32 x 256 A operand: [step][hh][row][8]
constexpr int kMlpLds = 2 * 3 * kMlpPlane + 2 * 32 * 4;     // x1 planes, H planes, (mean, rstd) of the rows

__device__ __forceinline__ int mlp_acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// byte offset of the 16-byte piece (step, hh, row) inside plane p of an A operand in LDS
// (rows are swizzled by the (step, hh) index: the stash of a row -- one wave, its lanes walking k -- and the 2-byte stores of H
//  -- lanes walking the hidden unit -- would otherwise hit pieces 512 B apart, i.e. the same banks: PMC before the swizzle,
//  1.28 M bank-conflict cycles of 2.43 M LDS-active cycles per launch; a fragment read takes 32 consecutive pieces either way)
__device__ __forceinline__ int mlp_a_off(int p, int step, int hh, int row)
{
    const int g = step * 2 + hh;
    return p * kMlpPlane + (g * 32 + (row ^ (g & 31))) * 16;
}
// 16-byte pieces of a weight matrix (rows x K) in memory: [row tile][step][plane][hh][row in tile]
__host__ __device__ __forceinline__ size_t mlp_w_piece(int t, int steps, int step, int p, int hh, int li)
{
    return ((((size_t)t * steps + step) * 3 + p) * 2 + hh) * 32 + li;
}

// W (rows, K) fp32 -> three bf16 planes in fragment order (rows % 32 == 0, K % 16 == 0); a thread per pair of k
__global__ void k_prep_planes(const float *W, int rows, int K, unsigned short *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * (K / 2)) return;
    const int row = i / (K / 2), k = (i - row * (K / 2)) * 2;
    unsigned q[3];
    split3_pair(W[(size_t)row * K + k], W[(size_t)row * K + k + 1], q[0], q[1], q[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const size_t piece = mlp_w_piece(row >> 5, K / 16, k >> 4, p, (k >> 3) & 1, row & 31);
        *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(out) + piece * 16 + (k & 7) * 2) = q[p];
    }
}

int launch_prep_planes(const float *W, int rows, int K, void *out, hipStream_t st)
{
    PTX_REQUIRE(rows % 32 == 0 && K % 16 == 0, "weight planes: rows=%d K=%d", rows, K);
    const int n = rows * (K / 2);
    hipLaunchKernelGGL(k_prep_planes, dim3(cdiv(n, 256)), dim3(256), 0, st, W, rows, K, static_cast<unsigned short *>(out));
    PTX_LAUNCHED("k_prep_planes");
    return PTX_OK;
}

// agent-scope relaxed store / loads (see fattn.hip: visible across the XCDs' L2s without a fence)
__device__ __forceinline__ void mlp_st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
typedef float mlp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mlp_f4 mlp_ld4_agent(const float *p)
{
    mlp_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void mlp_wait4(mlp_f4 &a, mlp_f4 &b, mlp_f4 &c, mlp_f4 &d)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory");
}

// LITE (calls with more work-groups than CUs: the shipped configuration at its training batch is 1 040): the H planes take the
// place of the x1 planes (one more barrier) and the weight fragments are requested four steps ahead instead of eight, so that TWO
// work-groups fit a CU (49 KB of LDS, <= 128 registers) and hide each other's round trips; with one work-group per CU -- the
// benchmark shape -- the deeper prefetch is what hides them (r03 stamps) and LITE is off.
template <int NP, bool LITE>       // NP 3: split operands (fp32-equivalent); 1: plain bf16 operands = the first plane only (compute_dtype = 1)
__global__ __launch_bounds__(kMlpWaves * 64) __attribute__((amdgpu_waves_per_eu(LITE ? 4 : 1, LITE ? 4 : 4))) void k_mlp(MlpBatch mb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MlpProb p = mb.p[blockIdx.z];
    const int row0 = blockIdx.x * kMlpRows, sl = blockIdx.y, nsl = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = 256, R = p.R;
    char *Xp = smem, *Hp = LITE ? smem : smem + 3 * kMlpPlane;
    float *s_mu = reinterpret_cast<float *>(smem + (LITE ? 3 : 6) * kMlpPlane), *s_rs = s_mu + 32;

    // ---- requests first: the x1 tile, the LayerNorm partials of its rows, the column terms, the first weight fragments
    float4 xv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 512 * i, row = min(row0 + (e >> 6), R - 1);
        xv[i] = *reinterpret_cast<const float4 *>(p.x1 + (size_t)row * C + (e & 63) * 4);
    }
    float2 lp[8];
    if (tid < 32) {
        const float2 *pp = reinterpret_cast<const float2 *>(p.lnp) + (size_t)min(row0 + tid, R - 1) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) lp[t] = pp[t];
    }
    const int j = sl * kMlpSlice + 32 * wv + li;                    // this lane's hidden unit in phase 1
    const float s_j = p.fc1_s[j], c_j = p.fc1_c[j];
    const u32x4 *W1 = reinterpret_cast<const u32x4 *>(p.w1p), *W2 = reinterpret_cast<const u32x4 *>(p.w2p);
    const int t1 = sl * (kMlpSlice / 32) + wv;                      // row tile of fc1's weight: 32 hidden units
    auto load_b = [&](const u32x4 *W, int t, int steps, int step, u32x4 (&b)[3]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) b[q] = W[mlp_w_piece(t, steps, step, q, hh, li)];
    };
    constexpr int kAhead = LITE ? 4 : 8;                            // weight fragments requested this many k steps ahead (r03
                                                                    // stamps: with 4 a phase was 10 k cycles for 3 k of MFMA time)
    u32x4 bq[kAhead][3];
#pragma unroll
    for (int s = 0; s < kAhead; ++s) load_b(W1, t1, 16, s, bq[s]);

    // ---- the x1 tile -> three bf16 planes (A operand of phase 1); row statistics
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 512 * i, row = e >> 6, k = (e & 63) * 4;
        stash_parts<NP>(Xp + mlp_a_off(0, k >> 4, (k >> 3) & 1, row) + (k & 4) * 2, kMlpPlane, xv[i]);
    }
    if (tid < 32) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; ++t) { s1 += lp[t].x; s2 += lp[t].y; }
        const float mu = s1 * (1.0f / 256.0f);
        const float var = fmaxf(fmaf(-mu, mu, s2 * (1.0f / 256.0f)), 0.0f);
        s_mu[tid] = mu; s_rs[tid] = 1.0f / sqrtf(var + mb.ln_eps);
    }
    __syncthreads();

    // ---- phase 1: H tile (32 rows x this wave's 32 hidden units), K = 256
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // (the A fragments of step s + 1 are read from LDS before the MFMAs of step s)
    auto load_a = [&](const char *base, int s, u32x4 (&a)[3]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const u32x4 *>(base + mlp_a_off(q, s, hh, li));
    };
    u32x4 aq[2][3];
    load_a(Xp, 0, aq[0]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s + 1 < 16) load_a(Xp, s + 1, aq[(s + 1) & 1]);
        acc = mfma_parts<NP>(aq[s & 1], bq[s % kAhead], acc);
        if (s + kAhead < 16) load_b(W1, t1, 16, s + kAhead, bq[s % kAhead]);
    }
    // phase 2's first weight fragments: in flight under the GELU epilogue
    const int st2 = sl * 16;                                        // first k step of this slice in fc2's K = 1024
#pragma unroll
    for (int s = 0; s < kAhead; ++s) load_b(W2, wv, 64, st2 + s, bq[s]);
    {
        float h[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mlp_acc_row(r, hh);
            h[r] = gelu_erf(fmaf(s_rs[row], fmaf(-s_mu[row], s_j, acc[r]), c_j));
        }
        if (LITE) __syncthreads();                                  // every wave is done reading the x1 planes H overwrites
        // column j of H is k = 32 wv + li of phase 2: step 2 wv + (li >> 4), half (li >> 3) & 1, position li & 7
        const int step = 2 * wv + (li >> 4), h2 = (li >> 3) & 1, pos = (li & 7) * 2;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            unsigned q1, q2 = 0u, q3 = 0u;
            if (NP == 3) split3_pair(h[r], h[r + 1], q1, q2, q3);
            else {
                const f32x2 hv = {h[r], h[r + 1]};
                q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(hv, bf16x2));
            }
            char *d0 = Hp + mlp_a_off(0, step, h2, mlp_acc_row(r, hh)) + pos, *d1 = Hp + mlp_a_off(0, step, h2, mlp_acc_row(r + 1, hh)) + pos;
            *reinterpret_cast<unsigned short *>(d0) = (unsigned short)(q1 & 0xffffu);
            *reinterpret_cast<unsigned short *>(d1) = (unsigned short)(q1 >> 16);
            if (NP == 3) {
                *reinterpret_cast<unsigned short *>(d0 + kMlpPlane) = (unsigned short)(q2 & 0xffffu);
                *reinterpret_cast<unsigned short *>(d1 + kMlpPlane) = (unsigned short)(q2 >> 16);
                *reinterpret_cast<unsigned short *>(d0 + 2 * kMlpPlane) = (unsigned short)(q3 & 0xffffu);
                *reinterpret_cast<unsigned short *>(d1 + 2 * kMlpPlane) = (unsigned short)(q3 >> 16);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: partial fc2 output (32 rows x this wave's 32 output columns) over the slice's 256 hidden units
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    load_a(Hp, 0, aq[0]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s + 1 < 16) load_a(Hp, s + 1, aq[(s + 1) & 1]);
        acc = mfma_parts<NP>(aq[s & 1], bq[s % kAhead], acc);
        if (s + kAhead < 16) load_b(W2, wv, 64, st2 + s + kAhead, bq[s % kAhead]);
    }
    const int n = 32 * wv + li;                                     // this lane's output column
    const int tile = blockIdx.x, ntiles = gridDim.x;
    float *part = mb.part + (((size_t)blockIdx.z * ntiles + tile) * nsl + sl) * (kMlpRows * 256);
#pragma unroll
    for (int r = 0; r < 16; ++r) mlp_st_agent(part + mlp_acc_row(r, hh) * 256 + n, acc[r]);

    // ---- the last slice of this row tile to arrive adds the partials (slice order), the bias and the residual
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *tk = mb.tickets + blockIdx.z * ntiles + tile;
    int *flag = reinterpret_cast<int *>(s_mu);                      // (the statistics are dead)
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != nsl - 1) return;
    if (tid == 0) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left zero for the next launch
    const float *gp = mb.part + ((size_t)blockIdx.z * ntiles + tile) * nsl * (kMlpRows * 256);
    // the output head of the block on the finished rows (trailing LayerNorm + Linear(256, 3 | 9) + eval BatchNorm1d, PRE:443-446,
    // 452-455): a wave holds whole rows here -- lane l has columns 4 l .. 4 l + 3 -- so k_heads is no launch of its own
    const bool heads = p.head_out != nullptr;
    constexpr int kMaxOut = 9;
    float4 hwt[kMaxOut], nw4 = make_float4(0.f, 0.f, 0.f, 0.f), nb4 = nw4;
    float hbias = 0.0f, bn_a = 0.0f, bn_b = 0.0f;
    if (heads) {
        nw4 = *reinterpret_cast<const float4 *>(p.nw + 4 * lane); nb4 = *reinterpret_cast<const float4 *>(p.nb + 4 * lane);
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o)
            hwt[o] = *reinterpret_cast<const float4 *>(p.hw + (size_t)min(o, p.nout - 1) * C + 4 * lane);
        if (lane < p.nout) { hbias = p.hb[lane]; bn_a = p.ab[lane]; bn_b = p.ab[p.nout + lane]; }
    }
    // (every partial, residual and bias request of the four rows of this wave goes out before the first is used: as a loop
    //  over the rows the merge was four round trips to memory in sequence, 10 k of the last slice's 46 k cycles)
    constexpr int RB = LITE ? 2 : 4;                               // rows of this wave in flight at once (LITE: register budget)
    const float4 b = *reinterpret_cast<const float4 *>(p.b2 + 4 * lane);
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += RB) {
    mlp_f4 v[RB][4];
    float4 xr[RB];
#pragma unroll
    for (int ii = 0; ii < RB; ++ii) {
        const int row = wv + 8 * (i0 + ii), c4 = 4 * lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[ii][s] = mlp_ld4_agent(gp + (size_t)min(s, nsl - 1) * (kMlpRows * 256) + row * 256 + c4);
        xr[ii] = *reinterpret_cast<const float4 *>(p.x1 + (size_t)min(row0 + row, R - 1) * C + c4);
    }
#pragma unroll
    for (int ii = 0; ii < RB; ++ii) {
        const int i = ii, row = wv + 8 * (i0 + ii), c4 = 4 * lane;
        mlp_wait4(v[i][0], v[i][1], v[i][2], v[i][3]);
        const float4 x = xr[i];
        mlp_f4 sum = v[i][0];
#pragma unroll
        for (int s = 1; s < 4; ++s)
            if (s < nsl) sum += v[i][s];
        const float y[4] = {(sum[0] + b.x) + x.x, (sum[1] + b.y) + x.y, (sum[2] + b.z) + x.z, (sum[3] + b.w) + x.w};
        const bool live = row0 + row < R;                       // wave-uniform: a wave owns whole rows
        if (live) *reinterpret_cast<float4 *>(p.x2 + (size_t)(row0 + row) * C + c4) = make_float4(y[0], y[1], y[2], y[3]);
        if (!heads || !live) continue;
        const float mean = wave_sum((y[0] + y[1]) + (y[2] + y[3])) * (1.0f / 256.0f);
        float var = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = y[q] - mean; var = fmaf(d, d, var); }
        const float rstd = 1.0f / sqrtf(wave_sum(var) * (1.0f / 256.0f) + mb.ln_eps);
        const float g[4] = {(y[0] - mean) * rstd * nw4.x + nb4.x, (y[1] - mean) * rstd * nw4.y + nb4.y,
                            (y[2] - mean) * rstd * nw4.z + nb4.z, (y[3] - mean) * rstd * nw4.w + nb4.w};
        if (p.guide) *reinterpret_cast<float4 *>(p.guide + (size_t)(row0 + row) * C + c4) = make_float4(g[0], g[1], g[2], g[3]);
        float mine = 0.0f;
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < p.nout) {
                const float part = fmaf(hwt[o].w, g[3], fmaf(hwt[o].z, g[2], fmaf(hwt[o].y, g[1], hwt[o].x * g[0])));
                const float s = wave_sum(part);
                if (lane == o) mine = s;
            }
        }
        if (lane < p.nout) p.head_out[(size_t)(row0 + row) * p.nout + lane] = fmaf(mine + hbias, bn_a, bn_b);     // eval BatchNorm1d
    }
    }
}

bool mlp_fused_supported(int C, int hidden, int R, int compute_dtype)
{
    // r03 (one work-group per CU): 4146 rows (the shipped configuration at 6 scenes) +4.5 % over the two GEMM launches, 4096 rows
    // (cfg2 at 16 scenes) +0.7 %, 8192 rows (32 scenes) -3 %.  r04, with two work-groups per CU where there are more than 768 of
    // them (LITE): cfg2 at 16 scenes +3.8 % on top, at 32 scenes 26.2k vs 25.5k scenes/s for the two launches
    // (profiles/r04_mlp_lite_ab.txt) -- the fused form now covers every batch up to 36 scenes per call
    constexpr int rmax = 9216;
    return (compute_dtype == 0 || compute_dtype == 1) && C == 256 && hidden == 1024 && R >= 1 && R <= rmax;
}

size_t mlp_part_bytes(int R) { return (size_t)2 * cdiv(R, kMlpRows) * 4 * kMlpRows * 256 * sizeof(float); }
size_t mlp_ticket_bytes(int R) { return (size_t)2 * cdiv(R, kMlpRows) * sizeof(int); }

int launch_mlp(const MlpBatch &mb, hipStream_t st)
{
    PTX_REQUIRE(mb.n >= 1 && mb.n <= 2 && mb.part && mb.tickets, "fused mlp: bad batch");
    int rmax = 0;
    for (int g = 0; g < mb.n; ++g) {
        const MlpProb &p = mb.p[g];
        PTX_REQUIRE(p.x1 && p.lnp && p.w1p && p.w2p && p.fc1_s && p.fc1_c && p.b2 && p.x2 && p.R >= 1, "fused mlp: null operand in group %d", g);
        PTX_REQUIRE(p.R == mb.p[0].R, "fused mlp: the groups must have the same number of rows");
        rmax = p.R;
    }
    const dim3 grid(cdiv(rmax, kMlpRows), 4, mb.n), block(kMlpWaves * 64);
    // r04 A/B: 512 work-groups -0.4 %, 1024 +3.8 %.  r05: also where the one-per-CU form would run a second round that is less than
    // ~60 % full (257 .. 400 work-groups: cfg2 at 5 / 6 scenes +1.6 / +1.1 %, cfg4 at 2 scenes +2.5 %; 448 and 512 work-groups: 0 / -0.5 %,
    // profiles/r05_mlp_lite_mid_ab.txt)
    const long wgs = (long)grid.x * grid.y * grid.z;
    const bool lite = wgs > 768 || (wgs > 256 && wgs <= 400);
    constexpr int kLiteLds = 3 * kMlpPlane + 2 * 32 * 4;
#define PTX_MLP_LAUNCH(NP_, LITE_, LDS_)                                                                                           \
    do {                                                                                                                         \
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp<NP_, LITE_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_)); \
        hipLaunchKernelGGL((k_mlp<NP_, LITE_>), grid, block, LDS_, st, mb);                                                      \
    } while (0)
    if (mb.compute_dtype == 1) { if (lite) PTX_MLP_LAUNCH(1, true, kLiteLds); else PTX_MLP_LAUNCH(1, false, kMlpLds); }
    else { if (lite) PTX_MLP_LAUNCH(3, true, kLiteLds); else PTX_MLP_LAUNCH(3, false, kMlpLds); }
#undef PTX_MLP_LAUNCH
    PTX_LAUNCHED("k_mlp");
    return PTX_OK;
}

}  // namespace ptx
