// fp32 matrix-core kernels of the proxy blocks and the folded attention pool.
//
//   k_gemm_nt   C[r][n] = sum_k A[r][k] * W[n][k]  (+bias, GELU, residual, row-scaled addend)
//               grouped: blockIdx.z selects one of up to 8 problems (text / image branch,
//               or the 8 heads of the folded AttentionPool2d tables).
//               v_mfma_f32_32x32x2_f32: exact fp32 (parity config), 64x64 tile per
//               4-wave work-group, each wave one 32x32 accumulator.
//   k_ln_rows   LayerNorm over the channel dim, one wave per row (PRE:275 norm2, PRE:340 norm_img)
//   k_heads     trailing LayerNorm + Linear(C,3|9) + eval BatchNorm1d (PRE:443-446, 452-455)
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "split3.h"

namespace ptx {

constexpr int BK = 32, LDT = BK + 4;   // 144-B LDS rows: 16-B aligned, b128 fragment reads conflict-free

// ---- three-way bf16 operand split (the "x3" kernels): split3.h
// the value is "re-defined" here: arithmetic on a prefetched register cannot be hoisted above this point (hipcc moves
// pure VALU work across s_barrier, which turns a three-steps-ahead prefetch into a wait on the loads just issued)
__device__ __forceinline__ void pin4(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// ---- LayerNorm fold helpers (GemmProb::lnp_out / lnp_in) ----------------------------------------------------------
// consumer: mean / rstd of row `row` from the producer's per-tile partials
__device__ __forceinline__ void ln_row_stats(const GemmProb &pr, int row, float &mu, float &rstd)
{
    // (all partials are requested before the first is added: as a loop the compiler waited for each load before issuing the
    //  next -- eight or sixteen dependent L2 round trips in front of the K loop of the work-group's first wave)
    const float2 *pp = reinterpret_cast<const float2 *>(pr.lnp_in) + (size_t)row * pr.ln_parts;
    constexpr int kMaxParts = 16;                           // embed_dim <= 512 (validate_shape)
    float2 v[kMaxParts];
#pragma unroll
    for (int t = 0; t < kMaxParts; ++t) v[t] = pp[min(t, pr.ln_parts - 1)];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < kMaxParts; ++t) { s1 += t < pr.ln_parts ? v[t].x : 0.0f; s2 += t < pr.ln_parts ? v[t].y : 0.0f; }
    const float inv = 1.0f / (float)pr.ln_C;
    mu = s1 * inv;
    const float var = fmaxf(fmaf(-mu, mu, s2 * inv), 0.0f);
    rstd = 1.0f / sqrtf(var + pr.ln_eps);
}
// producer: one wave holds the final values v[r] of a 32x32 tile in the MFMA C layout (col = lane & 31,
// row = (r&3) + 8 (r>>2) + 4 (lane>>5)); scratch = 32 x 33 floats of wave-private LDS.  Writes the (sum, sum of
// squares) of each row's 32 columns (invalid columns / rows contribute 0) to lnp_out[(row * parts + part) * 2].
__device__ __forceinline__ void ln_tile_partials(const GemmProb &pr, const float (&v)[16], float *scratch, int row0,
                                                 int part, int parts)
{
    const int lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + li] = v[r];
    // same wave: LDS operations complete in order, no barrier needed
    const float *rowp = scratch + li * 33 + 16 * hh;       // lane (li, hh) sums columns 16 hh .. 16 hh + 15 of row li
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { const float x = rowp[c]; s1 += x; s2 = fmaf(x, x, s2); }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const int row = row0 + li;
    if (hh == 0 && row < pr.R)
        *reinterpret_cast<float2 *>(pr.lnp_out + ((size_t)row * parts + part) * 2) = make_float2(s1, s2);
}

// the join of the two chains folded into a launch (GemmBatch::tail_gate): ONE thread of the launch waits for the other stream's
// signal after its own work is done; the launch, and with it everything behind it on its stream, completes only then
#define PTX_TAIL_GATE(gb_, first_thread_)                                                                          \
    do {                                                                                                           \
        if ((gb_).tail_gate.flag != nullptr && (first_thread_) && (blockIdx.x | blockIdx.y | blockIdx.z) == 0)     \
            gate_wait((gb_).tail_gate);                                                                            \
    } while (0)

// Throughput-regime kernel: 64x64 output tile per 4-wave work-group, one 32x32 MFMA accumulator
// per wave.  K loop, BK = 32 per step, three stages: tile it is consumed from LDS, tile it+1
// sits in the other LDS buffer, tile it+2 is in flight in registers (two register sets, loop
// unrolled by two) -- a fetch has two full MFMA phases (>= 2000 cycles) to land.  Fetches are
// branch-free (clamped rows; the K tail is zeroed on the W side only).
#define PTX_G64_FETCH(S, it_)                                                              \
    do {                                                                                   \
        const int k_ = min((it_), nk - 1) * BK + kq, kc_ = min(k_, pr.K - 4);              \
        const bool ok_ = k_ < pr.K;                                                        \
        a##S##0 = *reinterpret_cast<const float4 *>(pr.A + ao0 + kc_);                     \
        a##S##1 = *reinterpret_cast<const float4 *>(pr.A + ao1 + kc_);                     \
        w##S##0 = *reinterpret_cast<const float4 *>(pr.W + wo0 + kc_);                     \
        w##S##1 = *reinterpret_cast<const float4 *>(pr.W + wo1 + kc_);                     \
        if (!ok_) { w##S##0 = z4; w##S##1 = z4; }                                          \
    } while (0)
#define PTX_G64_STASH(S, buf_)                                                             \
    do {                                                                                   \
        *reinterpret_cast<float4 *>(&As[buf_][sr][kq]) = a##S##0;                          \
        *reinterpret_cast<float4 *>(&As[buf_][sr + 32][kq]) = a##S##1;                     \
        *reinterpret_cast<float4 *>(&Ws[buf_][sr][kq]) = w##S##0;                          \
        *reinterpret_cast<float4 *>(&Ws[buf_][sr + 32][kq]) = w##S##1;                     \
    } while (0)
#define PTX_G64_COMPUTE(buf_)                                                              \
    do {                                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < BK / 8; ++kk) {                            \
            const float4 a4 = *reinterpret_cast<const float4 *>(&As[buf_][wr * 32 + li][kk * 8 + hh * 4]); \
            const float4 b4 = *reinterpret_cast<const float4 *>(&Ws[buf_][wc * 32 + li][kk * 8 + hh * 4]); \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);          \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);          \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);          \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);          \
        }                                                                                  \
    } while (0)

__global__ __launch_bounds__(256) void k_gemm64(GemmBatch gb)
{
    const GemmProb pr = gb.p[blockIdx.z];          // by value: fields live in SGPRs, not re-read from kernarg
    const int row0 = blockIdx.x * 64, col0 = blockIdx.y * 64;
    if (row0 >= pr.R || col0 >= pr.N) return;
    __shared__ __attribute__((aligned(16))) float As[2][64][LDT];
    __shared__ __attribute__((aligned(16))) float Ws[2][64][LDT];
    __shared__ float s_mu[64], s_rs[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (pr.lnp_in != nullptr && tid < 64) {
        float mu, rs;
        ln_row_stats(pr, min(row0 + tid, pr.R - 1), mu, rs);
        s_mu[tid] = mu; s_rs[tid] = rs;                    // read after the barriers of the K loop
    }
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, hh = lane >> 5;
    const int sr = tid >> 3, kq = (tid & 7) * 4;        // staging: rows sr and sr + 32
    const size_t ao0 = (size_t)min(row0 + sr, pr.R - 1) * pr.lda, ao1 = (size_t)min(row0 + sr + 32, pr.R - 1) * pr.lda;
    const size_t wo0 = (size_t)min(col0 + sr, pr.N - 1) * pr.ldw, wo1 = (size_t)min(col0 + sr + 32, pr.N - 1) * pr.ldw;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 aA0, aA1, wA0, wA1, aB0, aB1, wB0, wB1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int nk = (pr.K + BK - 1) / BK;
    // epilogue operands are requested up front: a dependent ~1 us round trip after the K loop otherwise
    const int n = col0 + wc * 32 + li;
    const float bias = (pr.bias && n < pr.N) ? pr.bias[n] : 0.0f;
    float resv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        resv[r] = (pr.res && row < pr.R && n < pr.N) ? pr.res[(size_t)row * pr.ldres + n] : 0.0f;
    }
    PTX_G64_FETCH(A, 0);
    PTX_G64_FETCH(B, 1);
    PTX_G64_STASH(A, 0);
    __syncthreads();
    for (int it = 0; it < nk; it += 2) {
        PTX_G64_FETCH(A, it + 2);
        PTX_G64_COMPUTE(0);
        PTX_G64_STASH(B, 1);
        __syncthreads();
        if (it + 1 >= nk) break;
        PTX_G64_FETCH(B, it + 3);
        PTX_G64_COMPUTE(1);
        PTX_G64_STASH(A, 0);
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool ncol = n < pr.N;
    float lns = 0.0f, lnc = 0.0f;
    if (pr.lnp_in != nullptr && ncol) { lns = pr.ln_s[n]; lnc = pr.ln_c[n]; }
    float fin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, row = row0 + rl;
        float v = acc[r] + bias;
        if (pr.lnp_in != nullptr) v = fmaf(s_rs[rl], fmaf(-s_mu[rl], lns, acc[r]), lnc);
        if (pr.epi == EPI_GELU) v = gelu_erf(v);
        const bool ok = ncol && row < pr.R;
        if (ok) {
            if (pr.rs) v = fmaf(pr.rs[(size_t)row * pr.rs_stride], pr.ad[(size_t)row * pr.ldad + n], v);
            if (pr.res) v += resv[r];
            pr.C[(size_t)row * pr.ldc + n] = v;
        }
        fin[r] = ok ? v : 0.0f;
    }
    if (pr.lnp_out != nullptr)      // every wave is past its last LDS read (barrier after the last stash)
        ln_tile_partials(pr, fin, &As[0][0][0] + wid * (32 * 33), row0 + wr * 32, (col0 >> 5) + wc, (pr.N + 31) >> 5);
    PTX_TAIL_GATE(gb, tid == 0);
}
#undef PTX_G64_FETCH
#undef PTX_G64_STASH
#undef PTX_G64_COMPUTE

// (Measured on the way, r02: 2048x1024x1024 takes 49.4 us with the fp32 instruction -- 55 % of the matrix pipe, and neither
// four register stages, nor removing the stash or the barrier moves it much: the phases of a step do not overlap
// across the two resident work-groups of a CU.  The split kernel: 37.8 us; its matrix part is 3/8 but the split is
// ~110 VALU instructions per step and VALU and MFMA time mostly ADD (SQ_VALU_MFMA_COEXEC 24 % of MFMA busy).  A
// register-direct variant -- every lane loads its own k-contiguous fragments from global memory, no LDS at all -- is
// correct and slower (68 us): 32 rows x 16 B per load instruction touches 32 cache lines.)
// The same tiling with the operands split three ways into bf16 on the way into LDS (see split3_pair) and the K loop
// written out for K = 128 NKG (every problem of the batch): 3/8 of the matrix-pipe cycles of the fp32 instruction.
// Step t is consumed from LDS[t & 1] while step t + 1 is split (VALU) and stored into the other buffer IN THE SHADOW
// of step t's matrix instructions -- a wave's next MFMA cannot issue before the previous one leaves the pipe, the
// stash fills those slots (one scheduling region per step, the interleave spelled out with sched_group_barrier);
// steps t + 2 .. t + 4 are in flight in registers.  Written out, not a loop: at a back edge hipcc copies the
// loop-carried prefetch registers and waits for the loads just issued to do it; and arithmetic on a prefetched
// register is pinned (pin4) below its step's barrier, or hipcc hoists it and waits there.
#define PTX_X64_FETCH(S, it_)                                                              \
    do {                                                                                   \
        const int kc_ = (it_) * BK + kq;                                                   \
        a##S##0 = *reinterpret_cast<const float4 *>(pr.A + ao0 + kc_);                     \
        a##S##1 = *reinterpret_cast<const float4 *>(pr.A + ao1 + kc_);                     \
        w##S##0 = *reinterpret_cast<const float4 *>(pr.W + wo0 + kc_);                     \
        w##S##1 = *reinterpret_cast<const float4 *>(pr.W + wo1 + kc_);                     \
    } while (0)
#define PTX_X64_STASH(S, buf_)                                                             \
    do {                                                                                   \
        pin4(a##S##0); pin4(a##S##1); pin4(w##S##0); pin4(w##S##1);                        \
        char *d_ = smem + (buf_) * kXBuf + sr * XROW + xswz(sr, kq >> 3) + (kq & 4) * 2;   /* (sr + 32) swizzles alike */ \
        stash_parts<NP>(d_, kXPlane, a##S##0);                                             \
        stash_parts<NP>(d_ + 32 * XROW, kXPlane, a##S##1);                                 \
        stash_parts<NP>(d_ + 3 * kXPlane, kXPlane, w##S##0);                               \
        stash_parts<NP>(d_ + 3 * kXPlane + 32 * XROW, kXPlane, w##S##1);                   \
    } while (0)
#define PTX_X64_COMPUTE(buf_)                                                              \
    do {                                                                                   \
        const char *A_ = smem + (buf_) * kXBuf + (wr * 32 + li) * XROW;                    \
        const char *W_ = smem + (buf_) * kXBuf + 3 * kXPlane + (wc * 32 + li) * XROW;      \
        _Pragma("unroll") for (int kg = 0; kg < BK / 16; ++kg) {                           \
            const int o_ = xswz(li, kg * 2 + hh);           /* wr * 32, wc * 32 do not change the swizzle */ \
            const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(A_ + o_);                  \
            const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(W_ + o_);                  \
            if (NP == 1) {          /* plain bf16 operands */                              \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);       \
                continue;                                                                  \
            }                                                                              \
            const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(A_ + o_ + kXPlane);        \
            const bf16x8 a3 = *reinterpret_cast<const bf16x8 *>(A_ + o_ + 2 * kXPlane);    \
            const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(W_ + o_ + kXPlane);        \
            const bf16x8 b3 = *reinterpret_cast<const bf16x8 *>(W_ + o_ + 2 * kXPlane);    \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);   /* small terms first */ \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);           \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);           \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);           \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);           \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);           \
        }                                                                                  \
    } while (0)

template <int NKG, int REP, int NP = 3>   // K = 128 NKG REP: the written-out body runs REP times (one drain of the prefetch per
                                         // repetition); NP = 3: split operands (fp32-equivalent), 1: plain bf16 operands
__global__ __launch_bounds__(256) void k_gemm64x(GemmBatch gb)
{
    constexpr int kXPlane = 64 * XROW, kXBuf = 6 * kXPlane;     // one plane of a 64 x 32 tile; [A1 A2 A3 W1 W2 W3] per buffer
    static_assert(2 * kXBuf >= 4 * 32 * 33 * 4, "the LayerNorm-partials scratch re-uses the staging area");
    const GemmProb pr = gb.p[blockIdx.z];          // by value: fields live in SGPRs, not re-read from kernarg
    const int row0 = blockIdx.x * 64, col0 = blockIdx.y * 64;
    if (row0 >= pr.R || col0 >= pr.N) return;
    __shared__ __attribute__((aligned(16))) char smem[2 * kXBuf];
    __shared__ float s_mu[64], s_rs[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, hh = lane >> 5;
    const int sr = tid >> 3, kq = (tid & 7) * 4;        // staging: rows sr and sr + 32
    const size_t ao0 = (size_t)min(row0 + sr, pr.R - 1) * pr.lda, ao1 = (size_t)min(row0 + sr + 32, pr.R - 1) * pr.lda;
    const size_t wo0 = (size_t)min(col0 + sr, pr.N - 1) * pr.ldw, wo1 = (size_t)min(col0 + sr + 32, pr.N - 1) * pr.ldw;
    float4 aA0, aA1, wA0, wA1, aB0, aB1, wB0, wB1, aC0, aC1, wC0, wC1, aD0, aD1, wD0, wD1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // epilogue operands are requested up front: a dependent ~1 us round trip after the K loop otherwise
    const int n = col0 + wc * 32 + li;
    const float bias = (pr.bias && n < pr.N) ? pr.bias[n] : 0.0f;
    float lns = 0.0f, lnc = 0.0f;                           // LayerNorm-consumer column terms: requested here, not after the K loop
    if (pr.lnp_in != nullptr) { lns = pr.ln_s[min(n, pr.N - 1)]; lnc = pr.ln_c[min(n, pr.N - 1)]; }
    float resv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        resv[r] = (pr.res && row < pr.R && n < pr.N) ? pr.res[(size_t)row * pr.ldres + n] : 0.0f;
    }
#define PTX_X64_PIPE()                                                                     \
    do {                                                                                   \
        if (NP != 3) break;         /* the interleave below is written for the 12 MFMAs of the split product */ \
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);             /* fragment reads */ \
        _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);                   \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                             \
        }                                                                                  \
    } while (0)
#define PTX_X64_STEP(S, SN, par_)                                                          \
    if (it + 4 < 4 * NKG || rep + 1 < REP) PTX_X64_FETCH(S, kbase + it + 4);               \
    __builtin_amdgcn_sched_barrier(0);     /* the loads are issued first: live across the step, their own registers */ \
    PTX_X64_COMPUTE(par_);                                                                 \
    PTX_X64_STASH(SN, 1 - (par_));                                                         \
    PTX_X64_PIPE();                                                                        \
    __syncthreads();                                                                       \
    ++it;
    PTX_X64_FETCH(A, 0); PTX_X64_FETCH(B, 1); PTX_X64_FETCH(C, 2); PTX_X64_FETCH(D, 3);
    // (behind the first four tiles' requests: in front of them the statistics were a round trip of their own before the
    //  work-group's first wave had requested anything -- r03 stamps: 3 k of the 6.5 k cycles in front of the K loop)
    if (pr.lnp_in != nullptr && tid < 64) {
        float mu, rs;
        ln_row_stats(pr, min(row0 + tid, pr.R - 1), mu, rs);
        s_mu[tid] = mu; s_rs[tid] = rs;                    // read after the barriers of the K loop
    }
    PTX_X64_STASH(A, 0);
    __syncthreads();
    int kbase = 0;
    for (int rep = 0; rep < REP; ++rep) {
        int it = 0;
#pragma unroll
        for (int g = 0; g < NKG; ++g) {
            PTX_X64_STEP(A, B, 0) PTX_X64_STEP(B, C, 1) PTX_X64_STEP(C, D, 0) PTX_X64_STEP(D, A, 1)
        }
        kbase += 4 * NKG;
    }
#undef PTX_X64_STEP
#undef PTX_X64_PIPE
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool ncol = n < pr.N;
    float fin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, row = row0 + rl;
        float v = acc[r] + bias;
        if (pr.lnp_in != nullptr) v = fmaf(s_rs[rl], fmaf(-s_mu[rl], lns, acc[r]), lnc);
        if (pr.epi == EPI_GELU) v = gelu_erf(v);
        const bool ok = ncol && row < pr.R;
        if (ok) {
            if (pr.rs) v = fmaf(pr.rs[(size_t)row * pr.rs_stride], pr.ad[(size_t)row * pr.ldad + n], v);
            if (pr.res) v += resv[r];
            pr.C[(size_t)row * pr.ldc + n] = v;
        }
        fin[r] = ok ? v : 0.0f;
    }
    if (pr.lnp_out != nullptr)      // every wave is past its last LDS read (barrier after the last stash)
        ln_tile_partials(pr, fin, reinterpret_cast<float *>(smem) + wid * (32 * 33), row0 + wr * 32, (col0 >> 5) + wc,
                         (pr.N + 31) >> 5);
    PTX_TAIL_GATE(gb, tid == 0);
}
#undef PTX_X64_FETCH
#undef PTX_X64_STASH
#undef PTX_X64_COMPUTE

// ---- 128 x 128 tiles (r06) ------------------------------------------------------------------------------------------------
// The same split product on a tile four times as large: a work-group of four waves owns 128 x 128 outputs, a wave a 64 x 64
// quadrant = FOUR 32 x 32 accumulators (i, j) that share their fragments -- per 16 k a wave reads 2 x 3 A fragments and 2 x 3 W
// fragments (12 ds_read_b128) for 24 matrix instructions, where the 64 x 64 kernel reads 12 for 12; a thread splits four float4
// per 16 k for 24 MFMAs of its wave, where the 64 x 64 kernel splits four per 32 k for 12.  VERDICT r05 #3: the 64 x 64 kernel's
// K step takes ~1 us for 0.16 us of matrix time (0.21 of the bf16 pipe it runs on) because the VALU stream of the split and the
// LDS traffic per matrix instruction bound it, not the matrix pipe.
// Steps of 16 k (a 24 KB LDS buffer: two of them = 50 KB, so that TWO work-groups share a CU and cover each other's barriers and
// round trips), fetched from memory in "macro steps" of 32 k so that a row's 128-byte line is requested whole: thread (r, q) holds
// rows r and r + 64 at k = 4 q and 4 q + 16 of both operands -- eight float4 per macro step in one of two register sets (P / Q);
// sub-step (m, 0) consumes LDS[0] while the second half of macro step m is split into LDS[1], sub-step (m, 1) consumes LDS[1]
// while the first half of macro step m + 1 goes to LDS[0] and the loads of macro step m + 2 are issued into the set m just freed.
// (Measured and not kept, r06: half register sets -- each k half fetched ONE 16-k step ahead, 32 prefetch registers instead of 64, 168
// VGPRs, THREE work-groups per CU -- 2 - 8 % slower at every shape: profiles/r06_gemm128_lab.txt.)
// LDS layout per plane: [k half hh][row][8 bf16] with 64 B of padding between the halves -- a fragment read (16 lanes of one
// half, 16 B each, rows consecutive) covers 256 consecutive bytes, and the 8-byte stash writes of 16 consecutive lanes (four rows
// x both halves x two quads) hit sixteen distinct even banks: both conflict-free by construction (cf. xswz for the 64-B rows).
constexpr int kYHalf = 128 * 16 + 64, kYPlane = 2 * kYHalf, kYBuf = 6 * kYPlane;     // 2112 / 4224 / 25344 bytes
// lab knobs (scratch/g128_variants.py builds the library with other values; the product uses the defaults)
#ifndef PTX_G128_NM
#define PTX_G128_NM 8          // macro steps in the written-out body
#endif
#ifndef PTX_G128_ORDER
#define PTX_G128_ORDER 0       // 0: the six terms go term by term over the four accumulators; 1: accumulator by accumulator
#endif
#ifndef PTX_G128_SWZ
#define PTX_G128_SWZ 0         // 1: XCD-aware tile order (k_gemm128x)
#endif
#ifndef PTX_G128_PIPE
#define PTX_G128_PIPE 1        // 1: sched_group_barrier interleave of a step (reads, then MFMA + VALU + DS-write groups): +3-5 %
#endif

#define PTX_Y_FETCH(S, m_)                                                                 \
    do {                                                                                   \
        const int kc_ = (m_) * 32 + kq;                                                    \
        a##S##0 = *reinterpret_cast<const float4 *>(pr.A + ao0 + kc_);                     \
        a##S##1 = *reinterpret_cast<const float4 *>(pr.A + ao0 + kc_ + 16);                \
        a##S##2 = *reinterpret_cast<const float4 *>(pr.A + ao1 + kc_);                     \
        a##S##3 = *reinterpret_cast<const float4 *>(pr.A + ao1 + kc_ + 16);                \
        w##S##0 = *reinterpret_cast<const float4 *>(pr.W + wo0 + kc_);                     \
        w##S##1 = *reinterpret_cast<const float4 *>(pr.W + wo0 + kc_ + 16);                \
        w##S##2 = *reinterpret_cast<const float4 *>(pr.W + wo1 + kc_);                     \
        w##S##3 = *reinterpret_cast<const float4 *>(pr.W + wo1 + kc_ + 16);                \
    } while (0)
// half h_ (0 / 1) of register set S -> LDS buffer buf_ (rows sr and sr + 64 of both operands)
#define PTX_Y_STASH(S, h_, buf_)                                                           \
    do {                                                                                   \
        float4 &pa_ = (h_) ? a##S##1 : a##S##0, &pb_ = (h_) ? a##S##3 : a##S##2;           \
        float4 &pc_ = (h_) ? w##S##1 : w##S##0, &pd_ = (h_) ? w##S##3 : w##S##2;           \
        pin4(pa_); pin4(pb_); pin4(pc_); pin4(pd_);                                        \
        char *d_ = smem + (buf_) * kYBuf + stash_off;                                      \
        stash_parts<NP>(d_, kYPlane, pa_);                                                 \
        stash_parts<NP>(d_ + 64 * 16, kYPlane, pb_);                                       \
        stash_parts<NP>(d_ + 3 * kYPlane, kYPlane, pc_);                                   \
        stash_parts<NP>(d_ + 3 * kYPlane + 64 * 16, kYPlane, pd_);                         \
    } while (0)
// one 16-k step of the wave's quadrant from LDS buffer buf_: 12 fragment reads, 24 (NP = 3) or 4 (NP = 1) matrix instructions;
// the six products of a split pair go term by term over the four accumulators (small terms first within each accumulator, four
// independent accumulator chains in flight)
#define PTX_Y_COMPUTE(buf_)                                                                \
    do {                                                                                   \
        const char *A_ = smem + (buf_) * kYBuf + frag_a;                                   \
        const char *W_ = smem + (buf_) * kYBuf + frag_w;                                   \
        bf16x8 fa[2][3], fw[2][3];                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                   \
            _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) {                            \
                fa[i_][p_] = *reinterpret_cast<const bf16x8 *>(A_ + i_ * 512 + p_ * kYPlane); \
                fw[i_][p_] = *reinterpret_cast<const bf16x8 *>(W_ + i_ * 512 + p_ * kYPlane); \
            }                                                                              \
        if (NP == 3 && PTX_G128_ORDER == 0) {                                              \
            constexpr int ta_[6] = {2, 1, 0, 1, 0, 0}, tw_[6] = {0, 1, 2, 0, 1, 0};        \
            _Pragma("unroll") for (int t_ = 0; t_ < 6; ++t_)                               \
                _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                           \
                    acc[q_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q_ >> 1][ta_[t_]], fw[q_ & 1][tw_[t_]], acc[q_], 0, 0, 0); \
        } else if (NP == 3) {                                                              \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) acc[q_] = mfma_split6(fa[q_ >> 1], fw[q_ & 1], acc[q_]); \
        } else {                                                                           \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                               \
                acc[q_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q_ >> 1][0], fw[q_ & 1][0], acc[q_], 0, 0, 0); \
        }                                                                                  \
    } while (0)

template <int NM, int NP = 3>     // K = 32 NM nrep: the written-out body of NM macro steps runs nrep times (one drain of the prefetch
                                  // per repetition, like k_gemm64x's REP)
__global__ __launch_bounds__(256, 2) void k_gemm128x(GemmBatch gb, int nrep)
{
    static_assert(NM >= 2 && NM % 2 == 0, "macro steps come in P / Q pairs");
    const GemmProb pr = gb.p[blockIdx.z];
    int bx = blockIdx.x, by = blockIdx.y;
    if (PTX_G128_SWZ) {
        // XCD-aware tile order: work-groups are dealt round-robin over the 8 XCDs (linear id mod 8), each with its own L2.  The
        // launch order (row tile fastest) gives an XCD every 8th row tile of ONE column tile at a time, so every column tile re-reads
        // the whole A operand from beyond the L2; here an XCD owns a contiguous range of row tiles and walks their column tiles
        // first: an A row panel is fetched once per XCD and the (small) W operand stays L2-resident
        const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        if ((total & 7u) == 0u) {
            const unsigned l2 = (lin & 7u) * (total >> 3) + (lin >> 3);
            bx = (int)(l2 / gridDim.y); by = (int)(l2 % gridDim.y);
        }
    }
    const int row0 = bx * 128, col0 = by * 128;
    if (row0 >= pr.R || col0 >= pr.N) return;
    __shared__ __attribute__((aligned(16))) char smem[2 * kYBuf];
    __shared__ float s_mu[128], s_rs[128];
    static_assert(2 * kYBuf >= 4 * 32 * 33 * 4, "the LayerNorm-partials scratch re-uses the staging area");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, hh = lane >> 5;
    const int sr = tid >> 2, kq = (tid & 3) * 4;        // staging: rows sr and sr + 64, k = kq and kq + 16 of a macro step
    const size_t ao0 = (size_t)min(row0 + sr, pr.R - 1) * pr.lda, ao1 = (size_t)min(row0 + sr + 64, pr.R - 1) * pr.lda;
    const size_t wo0 = (size_t)min(col0 + sr, pr.N - 1) * pr.ldw, wo1 = (size_t)min(col0 + sr + 64, pr.N - 1) * pr.ldw;
    const int stash_off = (kq >> 3) * kYHalf + sr * 16 + (kq & 4) * 2;
    const int frag_a = hh * kYHalf + (wr * 64 + li) * 16, frag_w = 3 * kYPlane + hh * kYHalf + (wc * 64 + li) * 16;
    float4 aP0, aP1, aP2, aP3, wP0, wP1, wP2, wP3, aQ0, aQ1, aQ2, aQ3, wQ0, wQ1, wQ2, wQ3;
    f32x16 acc[4];                                       // (i, j) = (q >> 1, q & 1): rows wr * 64 + 32 i, columns wc * 64 + 32 j
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    const int total = NM * nrep;
    PTX_Y_FETCH(P, 0);
    PTX_Y_FETCH(Q, 1);
    if (pr.lnp_in != nullptr && tid < 128) {               // behind the first tiles' requests (k_gemm64x)
        float mu, rs;
        ln_row_stats(pr, min(row0 + tid, pr.R - 1), mu, rs);
        s_mu[tid] = mu; s_rs[tid] = rs;
    }
    PTX_Y_STASH(P, 0, 0);
    __syncthreads();
    // sub-step (m, 0): consume LDS[0], second half of set S (macro step m) -> LDS[1]
    // sub-step (m, 1): loads of macro step m + 2 -> set S (free now); consume LDS[1]; first half of set T (macro step m + 1) -> LDS[0]
#define PTX_Y_PIPE()                                                                       \
    do {                                                                                   \
        if (NP != 3 || !PTX_G128_PIPE) break;                                              \
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);             /* fragment reads */ \
        _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                             \
        }                                                                                  \
    } while (0)
#define PTX_Y_MACRO(S, T, u_)                                                              \
    PTX_Y_COMPUTE(0);                                                                      \
    PTX_Y_STASH(S, 1, 1);                                                                  \
    PTX_Y_PIPE();                                                                          \
    __syncthreads();                                                                       \
    if ((u_) + 2 < NM || rep + 1 < nrep) PTX_Y_FETCH(S, mbase + (u_) + 2);                 \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    PTX_Y_COMPUTE(1);                                                                      \
    if ((u_) + 1 < NM || rep + 1 < nrep) PTX_Y_STASH(T, 0, 0);                             \
    PTX_Y_PIPE();                                                                          \
    __syncthreads();
    int mbase = 0;
    for (int rep = 0; rep < nrep; ++rep) {
#pragma unroll
        for (int u = 0; u < NM; u += 2) {
            PTX_Y_MACRO(P, Q, u)
            PTX_Y_MACRO(Q, P, u + 1)
        }
        mbase += NM;
    }
#undef PTX_Y_MACRO
#undef PTX_Y_PIPE
    (void)total;
    // epilogue, one accumulator at a time (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5));
    // its operands are requested per accumulator (64 residual registers up front would not fit beside the prefetch sets)
    float *scratch = reinterpret_cast<float *>(smem) + wid * (32 * 33);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = q >> 1, j = q & 1;
        const int n = col0 + wc * 64 + j * 32 + li, nc = min(n, pr.N - 1);
        const bool ncol = n < pr.N;
        const float bias = pr.bias ? pr.bias[nc] : 0.0f;
        float lns = 0.0f, lnc = 0.0f;
        if (pr.lnp_in != nullptr) { lns = pr.ln_s[nc]; lnc = pr.ln_c[nc]; }
        float resv[16], adv[16], rsv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(row0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, pr.R - 1);
            resv[r] = pr.res ? pr.res[(size_t)row * pr.ldres + nc] : 0.0f;
            adv[r] = pr.rs ? pr.ad[(size_t)row * pr.ldad + nc] : 0.0f;
            rsv[r] = pr.rs ? pr.rs[(size_t)row * pr.rs_stride] : 0.0f;
        }
        float fin[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, row = row0 + rl;
            float v = acc[q][r] + bias;
            if (pr.lnp_in != nullptr) v = fmaf(s_rs[rl], fmaf(-s_mu[rl], lns, acc[q][r]), lnc);
            if (pr.epi == EPI_GELU) v = gelu_erf(v);
            const bool ok = ncol && row < pr.R;
            if (pr.rs) v = fmaf(rsv[r], adv[r], v);
            if (pr.res) v += resv[r];
            if (ok) pr.C[(size_t)row * pr.ldc + n] = v;
            fin[r] = ok ? v : 0.0f;
        }
        if (pr.lnp_out != nullptr)      // every wave is past its last LDS read (barrier after the last step); the scratch is wave-private
            ln_tile_partials(pr, fin, scratch, row0 + wr * 64 + i * 32, (col0 >> 5) + wc * 2 + j, (pr.N + 31) >> 5);
    }
    PTX_TAIL_GATE(gb, tid == 0);
}
#undef PTX_Y_FETCH
#undef PTX_Y_STASH
#undef PTX_Y_COMPUTE

// Latency-regime variant for the small GEMMs of this path (a few hundred 32x32 tiles): one wave
// per (tile, K-slice).  SK waves of a work-group split the K range of ONE 32x32 tile, each with
// private LDS staging (no barrier in the K loop: a wave's LDS traffic is ordered), partial
// accumulators are summed in fixed wave order through LDS -> bit-reproducible.  Operand
// fetches are branch-free (clamped addresses; the K tail is zeroed on the W side only:
// out-of-range rows / columns are computed on valid memory and never stored), so the loop is one
// basic block and hipcc interleaves the next tile's global loads with the MFMAs.
// blockIdx.x = row tile: work-groups that share an A row-panel land on the same XCD (b % 8)
// whenever the row-tile count is a multiple of 8.
template <int SK, int AMODE, bool CHAIN = false>   // AMODE 1: A merged on the fly from the k_img_pool tiles; CHAIN: see GemmProb::w2
__global__ __launch_bounds__(SK * 64) void k_gemm32(GemmBatch gb)
{
    const GemmProb pr = gb.p[blockIdx.z];
    const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
    if (row0 >= pr.R || col0 >= pr.N) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *base = lds + (size_t)wv * (4 * 32 * LDT);        // [A0 | W0 | A1 | W1], each 32 x LDT
    const int li = lane & 31, hh = lane >> 5;
    const int nk = (pr.K + BK - 1) / BK, per = (nk + SK - 1) / SK;
    const int it0 = wv * per, it1 = min(nk, it0 + per);
    // staging role: e = lane + 64 i -> row e >> 3 = (lane >> 3) + 8 i, k offset (e & 7) * 4
    const int r0 = lane >> 3, kq = (lane & 7) * 4;
    const size_t a0 = (size_t)min(row0 + r0, pr.R - 1) * pr.lda, a1 = (size_t)min(row0 + r0 + 8, pr.R - 1) * pr.lda;
    const size_t a2 = (size_t)min(row0 + r0 + 16, pr.R - 1) * pr.lda, a3 = (size_t)min(row0 + r0 + 24, pr.R - 1) * pr.lda;
    const size_t w0 = (size_t)min(col0 + r0, pr.N - 1) * pr.ldw, w1 = (size_t)min(col0 + r0 + 8, pr.N - 1) * pr.ldw;
    const size_t w2 = (size_t)min(col0 + r0 + 16, pr.N - 1) * pr.ldw, w3 = (size_t)min(col0 + r0 + 24, pr.N - 1) * pr.ldw;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 aA0, aA1, aA2, aA3, wA0, wA1, wA2, wA3, aB0, aB1, aB2, aB3, wB0, wB1, wB2, wB3;
    float4 gA0 = z4, gA1 = z4, gA2 = z4, gA3 = z4, gB0 = z4, gB1 = z4, gB2 = z4, gB3 = z4;      // AMODE 1: second tile's sums
    int kA = 0, kB = 0;
    (void)kA; (void)kB;
    // AMODE 1: rows of the pooled partials and the split-softmax factors of this lane's four staging rows
    size_t g0 = 0, g1 = 0, g2 = 0, g3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    float fc0[4] = {0.f, 0.f, 0.f, 0.f}, fc1[4] = {0.f, 0.f, 0.f, 0.f}, fct[4] = {0.f, 0.f, 0.f, 0.f};
    float *cts = lds + (size_t)SK * (4 * 32 * LDT);         // [32] a_h(0) of the tile's rows (AMODE 1)
    float *lnst = cts + 32;                                 // [32][2] mean, rstd of the tile's rows (LayerNorm consumer)
    if (AMODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(row0 + r0 + 8 * i, pr.R - 1);
            const float *ml = pr.pml + (size_t)row * pr.ldml;
            const float m0 = ml[0], l0 = ml[1], m1 = ml[2], l1 = ml[3], s0 = ml[4];
            const float m = fmaxf(fmaxf(m0, m1), s0);
            const float x0 = __expf(m0 - m), x1 = __expf(m1 - m), xt = __expf(s0 - m);
            const float inv = 1.0f / ((l0 * x0 + l1 * x1) + xt);
            fc0[i] = x0 * inv; fc1[i] = x1 * inv; fct[i] = xt * inv;
            if (wv == 0 && (lane & 7) == 0) cts[r0 + 8 * i] = fct[i];
        }
        g0 = (size_t)min(row0 + r0, pr.R - 1) * pr.ldg; g1 = (size_t)min(row0 + r0 + 8, pr.R - 1) * pr.ldg;
        g2 = (size_t)min(row0 + r0 + 16, pr.R - 1) * pr.ldg; g3 = (size_t)min(row0 + r0 + 24, pr.R - 1) * pr.ldg;
        e0 = (size_t)min(row0 + r0, pr.R - 1) * pr.lde; e1 = (size_t)min(row0 + r0 + 8, pr.R - 1) * pr.lde;
        e2 = (size_t)min(row0 + r0 + 16, pr.R - 1) * pr.lde; e3 = (size_t)min(row0 + r0 + 24, pr.R - 1) * pr.lde;
    }
#define PTX_FETCH(S, j_)                                                                  \
    do {                                                                                  \
        int jj_ = min((j_), cnt - 1) + rot;                                               \
        jj_ -= jj_ >= cnt ? cnt : 0;                                                      \
        const int k_ = (it0 + jj_) * BK + kq, kc_ = min(k_, pr.K - 4);                    \
        const bool ok_ = k_ < pr.K;                                                       \
        if (AMODE == 0) {                                                                 \
            a##S##0 = *reinterpret_cast<const float4 *>(pr.A + a0 + kc_);                 \
            a##S##1 = *reinterpret_cast<const float4 *>(pr.A + a1 + kc_);                 \
            a##S##2 = *reinterpret_cast<const float4 *>(pr.A + a2 + kc_);                 \
            a##S##3 = *reinterpret_cast<const float4 *>(pr.A + a3 + kc_);                 \
        } else if (kc_ < pr.kg) {     /* wave-uniform: kg is a multiple of BK */          \
            a##S##0 = *reinterpret_cast<const float4 *>(pr.pg + g0 + kc_);                \
            a##S##1 = *reinterpret_cast<const float4 *>(pr.pg + g1 + kc_);                \
            a##S##2 = *reinterpret_cast<const float4 *>(pr.pg + g2 + kc_);                \
            a##S##3 = *reinterpret_cast<const float4 *>(pr.pg + g3 + kc_);                \
            g##S##0 = *reinterpret_cast<const float4 *>(pr.pg + g0 + pr.gslab + kc_);     \
            g##S##1 = *reinterpret_cast<const float4 *>(pr.pg + g1 + pr.gslab + kc_);     \
            g##S##2 = *reinterpret_cast<const float4 *>(pr.pg + g2 + pr.gslab + kc_);     \
            g##S##3 = *reinterpret_cast<const float4 *>(pr.pg + g3 + pr.gslab + kc_);     \
        } else {                                                                          \
            a##S##0 = *reinterpret_cast<const float4 *>(pr.pe + e0 + (kc_ - pr.kg));      \
            a##S##1 = *reinterpret_cast<const float4 *>(pr.pe + e1 + (kc_ - pr.kg));      \
            a##S##2 = *reinterpret_cast<const float4 *>(pr.pe + e2 + (kc_ - pr.kg));      \
            a##S##3 = *reinterpret_cast<const float4 *>(pr.pe + e3 + (kc_ - pr.kg));      \
        }                                                                                 \
        k##S = kc_;                                                                       \
        w##S##0 = *reinterpret_cast<const float4 *>(pr.W + w0 + kc_);                     \
        w##S##1 = *reinterpret_cast<const float4 *>(pr.W + w1 + kc_);                     \
        w##S##2 = *reinterpret_cast<const float4 *>(pr.W + w2 + kc_);                     \
        w##S##3 = *reinterpret_cast<const float4 *>(pr.W + w3 + kc_);                     \
        if (!ok_) { w##S##0 = z4; w##S##1 = z4; w##S##2 = z4; w##S##3 = z4; }             \
    } while (0)
#define PTX_MERGE(S, i_)                                                                  \
    do {                                                                                  \
        if (k##S < pr.kg) {                                                               \
            a##S##i_ = make_float4(a##S##i_.x * fc0[i_] + g##S##i_.x * fc1[i_], a##S##i_.y * fc0[i_] + g##S##i_.y * fc1[i_], \
                                   a##S##i_.z * fc0[i_] + g##S##i_.z * fc1[i_], a##S##i_.w * fc0[i_] + g##S##i_.w * fc1[i_]); \
        } else {                                                                          \
            const int t_ = k##S - pr.kg;      /* token: 0 = mean token, 1..128 tile 0, 129.. tile 1 */ \
            a##S##i_.x *= t_ == 0 ? fct[i_] : (t_ <= 128 ? fc0[i_] : fc1[i_]);           \
            a##S##i_.y *= t_ + 1 <= 128 ? fc0[i_] : fc1[i_];                              \
            a##S##i_.z *= t_ + 2 <= 128 ? fc0[i_] : fc1[i_];                              \
            a##S##i_.w *= t_ + 3 <= 128 ? fc0[i_] : fc1[i_];                              \
        }                                                                                 \
    } while (0)
#define PTX_STASH(S, buf_)                                                                \
    do {                                                                                  \
        float *A_s = base + (buf_) * (2 * 32 * LDT) + r0 * LDT + kq, *W_s = A_s + 32 * LDT; \
        if (AMODE == 1) { PTX_MERGE(S, 0); PTX_MERGE(S, 1); PTX_MERGE(S, 2); PTX_MERGE(S, 3); } \
        *reinterpret_cast<float4 *>(A_s) = a##S##0;                                       \
        *reinterpret_cast<float4 *>(A_s + 8 * LDT) = a##S##1;                             \
        *reinterpret_cast<float4 *>(A_s + 16 * LDT) = a##S##2;                            \
        *reinterpret_cast<float4 *>(A_s + 24 * LDT) = a##S##3;                            \
        *reinterpret_cast<float4 *>(W_s) = w##S##0;                                       \
        *reinterpret_cast<float4 *>(W_s + 8 * LDT) = w##S##1;                             \
        *reinterpret_cast<float4 *>(W_s + 16 * LDT) = w##S##2;                            \
        *reinterpret_cast<float4 *>(W_s + 24 * LDT) = w##S##3;                            \
    } while (0)
#define PTX_COMPUTE(buf_)                                                                 \
    do {                                                                                  \
        const float *A_ = base + (buf_) * (2 * 32 * LDT) + li * LDT + hh * 4, *W_ = A_ + 32 * LDT; \
        _Pragma("unroll") for (int kk = 0; kk < BK / 8; ++kk) {                           \
            const float4 a4 = *reinterpret_cast<const float4 *>(A_ + kk * 8);             \
            const float4 b4 = *reinterpret_cast<const float4 *>(W_ + kk * 8);             \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);         \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);         \
        }                                                                                 \
    } while (0)
    // epilogue operands (used by wave 0 only) are requested up front, branch-free (clamped row / column): after the K loop
    // each would be a dependent round trip -- and the row-scaled addend was one PER OUTPUT ROW, sixteen in sequence, because a
    // load behind the previous row's store cannot be hoisted above it (r03)
    const int n = col0 + li;
    float bias = 0.0f, lns = 0.0f, lnc = 0.0f, resv[16], adv[16], rsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { resv[r] = 0.0f; adv[r] = 0.0f; rsv[r] = 0.0f; }
    const bool has_ad = AMODE == 1 || pr.rs != nullptr;      // work-group uniform
    if (wv == 0) {
        const int nc = min(n, pr.N - 1);
        if (pr.bias) bias = pr.bias[nc];
        if (pr.lnp_in != nullptr) { lns = pr.ln_s[nc]; lnc = pr.ln_c[nc]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * hh, pr.R - 1);
            if (pr.res) resv[r] = pr.res[(size_t)row * pr.ldres + nc];
            if (has_ad) adv[r] = pr.ad[(size_t)row * pr.ldad + nc];
            if (AMODE != 1 && has_ad) rsv[r] = pr.rs[(size_t)row * pr.rs_stride];
        }
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int cnt = it1 - it0;
    // K tiles are walked from a per-work-group rotated start: work-groups that share an operand
    // panel (same row tile or same column tile) then miss on DIFFERENT cold lines instead of all
    // queueing on the same one (the sum order changes per tile, deterministically)
    const int rot = (cnt > 1 && gb.rotate) ? (int)((blockIdx.x + 3u * blockIdx.y) % (unsigned)cnt) : 0;
    if (cnt > 0) {
        // three stages: step j in LDS[cur], step j+1 in LDS[cur^1], step j+2 in flight (registers)
        PTX_FETCH(A, 0);
        PTX_FETCH(B, 1);
        // LayerNorm-consumer row statistics (wave 0 only, which also reads them): requested behind the first two tiles
        if (pr.lnp_in != nullptr && wv == 0 && lane < 32) {
            float mu, rs;
            ln_row_stats(pr, min(row0 + lane, pr.R - 1), mu, rs);
            lnst[2 * lane] = mu; lnst[2 * lane + 1] = rs;
        }
        // (a fourth stage -- a third register set, fetches three MFMA phases ahead -- for the pooled-A o-projection, whose K
        //  steps take 3.1 k cycles for 1 k of MFMA time next to the clustering stream's tail: 20.6 vs 21.8 us in the trace, no
        //  change of the step (r03).  That launch reads 31 MB of pooling partials in ~8 us: it waits on bandwidth, not latency.
        //  Reading those partials with streaming (nt) loads: -2 % on the step at 32 scenes, -1.3 % at 4, three interleaved pairs.)
        PTX_STASH(A, 0);
        for (int j = 0; j < cnt; j += 2) {
            PTX_FETCH(A, j + 2);
            PTX_COMPUTE(0);
            PTX_STASH(B, 1);
            if (j + 1 >= cnt) break;
            PTX_FETCH(B, j + 3);
            PTX_COMPUTE(1);
            PTX_STASH(A, 0);
        }
    }
#undef PTX_FETCH
#undef PTX_STASH
#undef PTX_COMPUTE
    const bool chain = CHAIN && pr.w2 != nullptr && (int)blockIdx.y < pr.chain_tiles;      // work-group uniform
    if (SK > 1) {
        // fixed-order reduction of the K slices: slice w parks its accumulator in its own LDS
        __syncthreads();
        if (wv > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) base[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wv > 0 && !chain) return;
        if (wv == 0) {
#pragma unroll
            for (int w = 1; w < SK; ++w) {
                const float *o = lds + (size_t)w * (4 * 32 * LDT);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += o[r * 64 + lane];
            }
        }
    }
    if (wv == 0) {
        const bool ncol = n < pr.N;
        float fin[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh, row = row0 + rl;
            float v = acc[r] + bias;
            if (pr.lnp_in != nullptr) v = fmaf(lnst[2 * rl + 1], fmaf(-lnst[2 * rl], lns, acc[r]), lnc);
            if (pr.epi == EPI_GELU) v = gelu_erf(v);
            const bool ok = ncol && row < pr.R;
            if (ok) {
                if (AMODE == 1) v = fmaf(cts[rl], adv[r], v);
                else if (pr.rs) v = fmaf(rsv[r], adv[r], v);
                if (pr.res) v += resv[r];
                pr.C[(size_t)row * pr.ldc + n] = v;
            }
            fin[r] = ok ? v : 0.0f;
        }
        if (pr.lnp_out != nullptr)      // wave 0's staging area is free: its K loop is over, the other slices parked elsewhere
            ln_tile_partials(pr, fin, lds, row0, col0 >> 5, (pr.N + 31) >> 5);
        if (chain) {                    // the finished tile, [row][column], for the chained product (wave 0's staging area)
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[((r & 3) + 8 * (r >> 2) + 4 * hh) * LDT + li] = fin[r];
        }
    }
    PTX_TAIL_GATE(gb, threadIdx.x == 0);
    if (!chain) return;
    if (CHAIN) {
        __syncthreads();
        // C2[r][m] = sum_k Y[r][k] W2[m][k]: the SK waves take the 32-column tiles of C2 round-robin; A fragments from LDS,
        // W2 rows (32 floats = one 128-byte line each) straight from memory, 16 fp32 MFMAs per tile
        const float *W2 = pr.w2 + (size_t)blockIdx.y * pr.w2_stride;
        float *C2 = pr.c2 + (size_t)blockIdx.y * pr.c2_stride;
        float4 ya[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ya[kk] = *reinterpret_cast<const float4 *>(lds + li * LDT + kk * 8 + hh * 4);
        const int nt2 = (pr.n2 + 31) >> 5;
        // (the next tile's W2 rows are requested before the current tile's MFMAs and stores: one exposed round trip per wave
        //  instead of one per tile.  r03 stamps of the qkv0 launch, 600 work-groups on 512 slots: a chain work-group lives
        //  41 k cycles -- K loop 12 k, slice sum 2 k, epilogue 4 k, this product 21 k = 3.5 k per tile for 1 k of MFMA time --
        //  a plain one 23 k, and the launch is bounded by two rounds of plain ones.  Tried without effect on the 25 us launch:
        //  K sliced two ways so that all 600 are resident (29 us); the stores issued behind the wait for the next rows; the
        //  tile order rotated per row tile so that the 25 work-groups of a head do not read the same W2 lines at once; ONE staging
        //  buffer per wave instead of two (a wave's LDS operations execute in order, the area is wave-private) and the chained
        //  variant under 170 VGPRs, so that all 600 work-groups are resident at once: correct, 24.2 us.  What bounds the launch
        //  is neither latency nor residency: the 200 chain work-groups write the 19 MB of `we` in ~9 us -- 2 TB/s of stores.)
        auto load_w2 = [&](int t, float4 (&wb)[4]) {
            const float *wr = W2 + (size_t)min(t * 32 + li, pr.n2 - 1) * 32 + hh * 4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wb[kk] = *reinterpret_cast<const float4 *>(wr + kk * 8);
        };
        float4 wb[4], wn[4];
        load_w2(min(wv, nt2 - 1), wb);
        for (int t = wv; t < nt2; t += SK) {
            const int m = t * 32 + li;
            load_w2(min(t + SK, nt2 - 1), wn);
            f32x16 c2;
#pragma unroll
            for (int i = 0; i < 16; ++i) c2[i] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[kk].x, wb[kk].x, c2, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[kk].y, wb[kk].y, c2, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[kk].z, wb[kk].z, c2, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[kk].w, wb[kk].w, c2, 0, 0, 0);
            }
            if (m < pr.n2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (row < pr.R) C2[(size_t)row * pr.ldc2 + m] = c2[r];
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wb[kk] = wn[kk];
        }
    }
}

template <int SK, int AMODE, bool CHAIN = false>
static int launch_gemm32(const GemmBatch &gb, int rmax, int nmax, hipStream_t st)
{
    const size_t lds = sizeof(float) * (SK * 4 * 32 * LDT + 32 + 64);
    if (lds > 64 * 1024)
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm32<SK, AMODE, CHAIN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_gemm32<SK, AMODE, CHAIN>), dim3(cdiv(rmax, 32), cdiv(nmax, 32), gb.n), dim3(SK * 64), lds, st, gb);
    return PTX_OK;
}

// 128 x 128 tiles in a launch from which k_gemm128x is used (ptx_gemm_policy: 0 = never, 1 = whenever the shape allows it)
static std::atomic<int> g_gemm128_min_tiles{256};
int gemm_policy(int min_tiles_128)
{
    return min_tiles_128 < 0 ? g_gemm128_min_tiles.load() : g_gemm128_min_tiles.exchange(min_tiles_128);
}

int launch_gemm(const GemmBatch &gb_in, hipStream_t st, int compute_dtype)
{
    GemmBatch gb = gb_in;
    gb.rotate = 0;                      // (rotating the tile order per group: measured neutral, r01)
    PTX_REQUIRE(gb.n >= 1 && gb.n <= kMaxGroups, "gemm: %d groups", gb.n);
    int rmax = 0, nmax = 0, kmin = 1 << 30, kmax = 0;
    long tiles32 = 0;
    for (int g = 0; g < gb.n; ++g) {
        const GemmProb &p = gb.p[g];
        PTX_REQUIRE((p.A || p.pg) && p.W && p.C, "gemm: null operand in group %d", g);
        PTX_REQUIRE((p.pg != nullptr) == (gb.p[0].pg != nullptr), "gemm: mixed A modes in one batch");
        PTX_REQUIRE(p.pg == nullptr || (p.pe && p.pml && p.ad && p.kg % BK == 0 && p.kg <= p.K && p.ldg % 4 == 0 &&
                                        p.gslab % 4 == 0 && p.lde % 4 == 0),
                    "gemm: bad pooled-A description in group %d", g);
        PTX_REQUIRE(p.R >= 1 && p.N >= 1 && p.K >= 4, "gemm: empty problem in group %d", g);
        PTX_REQUIRE(p.K % 4 == 0 && p.lda % 4 == 0 && p.ldw % 4 == 0,
                    "gemm: K=%d lda=%d ldw=%d must be multiples of 4", p.K, p.lda, p.ldw);
        PTX_REQUIRE(((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W) | reinterpret_cast<uintptr_t>(p.pg) |
                      reinterpret_cast<uintptr_t>(p.pe)) & 15) == 0,
                    "gemm: operands of group %d are not 16-byte aligned", g);
        PTX_REQUIRE(p.rs == nullptr || p.ad != nullptr, "gemm: row scale without addend");
        PTX_REQUIRE(p.lnp_in == nullptr || (p.ln_s && p.ln_c && p.ln_parts >= 1 && p.ln_C >= 1 && p.bias == nullptr),
                    "gemm: bad LayerNorm-consumer description in group %d", g);
        rmax = p.R > rmax ? p.R : rmax;
        nmax = p.N > nmax ? p.N : nmax;
        kmin = p.K < kmin ? p.K : kmin;
        kmax = p.K > kmax ? p.K : kmax;
        tiles32 += (long)cdiv(p.R, 32) * cdiv(p.N, 32);
    }
    if (gb.p[0].w2 != nullptr) {
        PTX_REQUIRE(gb.n == 1 && gb.p[0].pg == nullptr && gb.p[0].c2 && gb.p[0].n2 >= 1 && gb.p[0].chain_tiles >= 1 &&
                    gb.p[0].chain_tiles * 32 <= gb.p[0].N && gb.p[0].K >= 4 * BK && gb.p[0].lnp_out == nullptr,
                    "gemm: bad chained-product description");
        PTX_TRY((launch_gemm32<4, 0, true>(gb, rmax, nmax, st)));
        PTX_LAUNCHED("k_gemm");
        return PTX_OK;
    }
    constexpr int g64_min = 1024;
    // 128 x 128 tiles (k_gemm128x): K a multiple of 256 (256 .. 4096) and at least g_gemm128_min_tiles tiles in the launch (default
    // 256 = one per CU, ptx_gemm_policy).  Alone on the chip (scratch/gemm128_lab.py, profiles/r06_gemm128_lab.txt): 1.24 - 1.39 x
    // the 64 x 64 kernel from ~380 tiles on (8192 x 768 x 256 ... 16384 x 2048 x 512: 100 -> 160 TF fp32-equivalent), even at
    // 200 - 260 tiles, 0.65 - 0.8 x at 128 tiles (half of the CUs idle).  Groups of one launch may have any row count.
    long tiles128 = 0;
    for (int g = 0; g < gb.n; ++g) tiles128 += (long)cdiv(gb.p[g].R, 128) * cdiv(gb.p[g].N, 128);
    const int min128 = g_gemm128_min_tiles.load(std::memory_order_relaxed);
    // (at exactly one tile per CU the large tile only pays when a tile carries enough work: 2 x 8192 x 256 x 256 -- the blocks' proj at
    //  32 scenes, 256 tiles of K = 256 -- takes 37 us on 128 x 128 tiles and 31 on 64 x 64; from 1.5 tiles per CU on, or with K >= 512,
    //  it is ahead: profiles/r06_gemm128_lab.txt)
    const bool enough = tiles128 >= min128 && (min128 == 1 || 2 * tiles128 >= 3 * (long)min128 || kmin >= 512);
    const bool use128 = gb.p[0].pg == nullptr && kmin == kmax && kmin % 256 == 0 && kmin <= 4096 && min128 > 0 && enough;
    if (use128) {
        const dim3 grid(cdiv(rmax, 128), cdiv(nmax, 128), gb.n);
        const int nrep = kmin / (32 * PTX_G128_NM);
        if (compute_dtype == 1) hipLaunchKernelGGL((k_gemm128x<PTX_G128_NM, 1>), grid, dim3(256), 0, st, gb, nrep);
        else                    hipLaunchKernelGGL((k_gemm128x<PTX_G128_NM, 3>), grid, dim3(256), 0, st, gb, nrep);
        PTX_LAUNCHED("k_gemm128x");
        return PTX_OK;
    }
    if (compute_dtype == 1 && gb.p[0].pg == nullptr && kmin == kmax && kmin % 128 == 0 && kmin / 128 <= 8) {
        // reduced-precision mode: plain bf16 operands, fp32 accumulation, the 64 x 64-tile kernel at every size
        const dim3 grid(cdiv(rmax, 64), cdiv(nmax, 64), gb.n);
        const int nkg = kmin / 128;
        if (nkg == 1) hipLaunchKernelGGL((k_gemm64x<1, 1, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 2) hipLaunchKernelGGL((k_gemm64x<2, 1, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 4) hipLaunchKernelGGL((k_gemm64x<4, 1, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 8) hipLaunchKernelGGL((k_gemm64x<8, 1, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 6) hipLaunchKernelGGL((k_gemm64x<2, 3, 1>), grid, dim3(256), 0, st, gb);     // K = 3 C (train: qkv input gradient)
        else { set_error("gemm: K=%d in bf16 mode", kmin); return PTX_EINVAL; }
        PTX_LAUNCHED("k_gemm64x<bf16>");
        return PTX_OK;
    }
    if (gb.p[0].pg != nullptr) {
        // A merged on the fly from the pooling tiles: always the latency-regime kernel, K split four ways
        PTX_TRY((launch_gemm32<4, 1>(gb, rmax, nmax, st)));
    } else if (tiles32 >= g64_min) {
        // enough tiles to fill the chip: 64x64 tiles re-use each staged operand twice as often
        const dim3 grid(cdiv(rmax, 64), cdiv(nmax, 64), gb.n);
        const int nkg = (kmin == kmax && kmin % 128 == 0) ? kmin / 128 : 0;
        if (nkg == 1) hipLaunchKernelGGL((k_gemm64x<1, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 2) hipLaunchKernelGGL((k_gemm64x<2, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 4) hipLaunchKernelGGL((k_gemm64x<4, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 8) hipLaunchKernelGGL((k_gemm64x<8, 1>), grid, dim3(256), 0, st, gb);
        else if (nkg == 6) hipLaunchKernelGGL((k_gemm64x<2, 3>), grid, dim3(256), 0, st, gb);        // K = 3 C: the qkv input gradient (train)
        else if (nkg == 12) hipLaunchKernelGGL((k_gemm64x<4, 3>), grid, dim3(256), 0, st, gb);
        else if (nkg == 16) hipLaunchKernelGGL((k_gemm64x<8, 2>), grid, dim3(256), 0, st, gb);       // embed_dim 512: hidden = 2048
        else if (nkg == 32) hipLaunchKernelGGL((k_gemm64x<8, 4>), grid, dim3(256), 0, st, gb);
        else hipLaunchKernelGGL(k_gemm64, grid, dim3(256), 0, st, gb);      // fp32 matrix instruction, any K
    } else {
        // latency regime: aim for >= 4 waves per SIMD (4096 waves) by slicing K inside the work-group
        const int nk = cdiv(kmin, BK);
        int sk = 1;
        while (sk < 4 && tiles32 * sk < 4096 && nk >= 4 * sk) sk *= 2;
        if (sk == 1) PTX_TRY((launch_gemm32<1, 0>(gb, rmax, nmax, st)));
        else if (sk == 2) PTX_TRY((launch_gemm32<2, 0>(gb, rmax, nmax, st)));
        else PTX_TRY((launch_gemm32<4, 0>(gb, rmax, nmax, st)));
    }
    PTX_LAUNCHED("k_gemm");
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
// One wave handles kHeadRows consecutive rows: the LayerNorm / head weights are loaded once per
// wave (36 + 8 registers) instead of once per row -- cold, chip-wide shared lines are the cost
// of this kernel (measured: 12 of 20 us with one row per wave), not its arithmetic.
constexpr int kHeadRows = 4;

template <int Q>       // C / 64 values per lane
__global__ __launch_bounds__(256) void k_heads(HeadBatch hb)
{
    const HeadProb p = hb.p[blockIdx.y];
    const int row0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * kHeadRows);
    if (row0 >= p.R) return;
    const int lane = lane_id();
    constexpr int kMaxOut = 9;
    float hbias = 0.0f, bn_a = 0.0f, bn_b = 0.0f;
    if (lane < p.nout) { hbias = p.hb[lane]; bn_a = p.ab[lane]; bn_b = p.ab[p.nout + lane]; }
    float nw[Q], nb[Q], hwt[kMaxOut][Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { nw[q] = p.nw[lane + 64 * q]; nb[q] = p.nb[lane + 64 * q]; }
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o)
#pragma unroll
        for (int q = 0; q < Q; ++q) hwt[o][q] = o < p.nout ? p.hw[(size_t)o * hb.C + lane + 64 * q] : 0.0f;
    float x[kHeadRows][Q];
#pragma unroll
    for (int r = 0; r < kHeadRows; ++r)
#pragma unroll
        for (int q = 0; q < Q; ++q)
            x[r][q] = row0 + r < p.R ? p.x[(size_t)(row0 + r) * hb.C + lane + 64 * q] : 0.0f;
#pragma unroll
    for (int r = 0; r < kHeadRows; ++r) {
        const int row = row0 + r;
        if (row >= p.R) break;                               // wave-uniform
        float rsum = (x[r][0] + x[r][1]) + (x[r][2] + x[r][3]);
#pragma unroll
        for (int q = 4; q < Q; ++q) rsum += x[r][q];
        const float mean = wave_sum(rsum) / (float)hb.C;
        float var = 0.0f;
#pragma unroll
        for (int q = 0; q < Q; ++q) { const float d = x[r][q] - mean; var = fmaf(d, d, var); }
        const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)hb.C + hb.eps);
        float v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            v[q] = (x[r][q] - mean) * rstd * nw[q] + nb[q];
            if (p.guide) p.guide[(size_t)row * hb.C + lane + 64 * q] = v[q];
        }
        float mine = 0.0f;
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < p.nout) {
                float part = 0.0f;
#pragma unroll
                for (int q = 0; q < Q; ++q) part = fmaf(hwt[o][q], v[q], part);
                const float s = wave_sum(part);
                if (lane == o) mine = s;
            }
        }
        if (lane < p.nout) p.out[(size_t)row * p.nout + lane] = fmaf(mine + hbias, bn_a, bn_b);   // eval BatchNorm1d
    }
}

int launch_heads(const HeadBatch &hb, hipStream_t st)
{
    PTX_REQUIRE(hb.C == 256 || hb.C == 512, "heads: C=%d unsupported (256, 512)", hb.C);
    int rmax = 0;
    for (int g = 0; g < hb.n; ++g) {
        PTX_REQUIRE(hb.p[g].nout >= 1 && hb.p[g].nout <= 9, "heads: nout=%d", hb.p[g].nout);
        rmax = hb.p[g].R > rmax ? hb.p[g].R : rmax;
    }
    if (rmax == 0) return PTX_OK;
    const dim3 grid(cdiv(rmax, 4 * kHeadRows), hb.n);
    if (hb.C == 256) hipLaunchKernelGGL(k_heads<4>, grid, dim3(256), 0, st, hb);
    else             hipLaunchKernelGGL(k_heads<8>, grid, dim3(256), 0, st, hb);
    PTX_LAUNCHED("k_heads");
    return PTX_OK;
}

}  // namespace ptx
