// Attention pooling of bf16- / fp16-stored image features in ONE pass over the features after the mean
// (AttentionPool2d query 0, PRE:158-176; algebra in imgproxy.hip).  It replaced three launches -- scores on the
// matrix pipe, a softmax launch, weighted sums on the matrix pipe -- whose measurements are kept in profiles/LAB_NOTES_imgpool.md.
//
// A work unit is (image, tile of 128 pixels): 512 channel rows x 256 B = 128 KB, which is held in the
// REGISTERS of one 8-wave work-group (16 loads of 16 B per lane) from the moment it is loaded until it has
// been used twice:
//   1. scores   s_h(p) = sum_c w_h(c) f(c,p)  on the bf16 matrix pipe, K = channels: the lane's 8 x 8 block
//      of (channel, pixel) values is transposed in registers (v_perm_b32) into B fragments, one per pixel
//      column; the per-image head weights are split exactly into three 16-bit parts (A operand);
//      the eight waves' channel slices are summed through LDS in a fixed order by the wave that owns the head
//   2. tile-local softmax numerators  e = exp(s - m_tile), l_tile = sum e  (wave = head), split exactly into
//      three bf16 parts
//   3. weighted sums  G_h(c) = sum_p e_h(p) f(c,p), K = pixels: the SAME registers, moved to the lane map
//      that the MFMA needs with ds_bpermute (a fixed lane permutation that swaps the roles of "pixel
//      window" and "channel sub-block"), 4 channel rows x 128 pixels per MFMA
// The unit writes (m_tile, l_tile, e, G); the o-projection GEMM merges the two tiles of an image with the mean
// token while it loads its A operand (gemm.hip, k_gemm32<4, 1>; as a separate launch the merge cost 13 us): the
// usual split softmax,
//   m = max(m_0, m_1, s(0)),  l = l_0 e^(m_0-m) + l_1 e^(m_1-m) + e^(s(0)-m),  g = (G_0 e^(m_0-m) + G_1 e^(m_1-m)) / l.
// Every product is of two bf16 values (exact in fp32) accumulated in fp32: the result is the fp32 value in a
// different summation order.
//
// The measurements behind this design -- every decomposition, load map, cache policy and scheduling variant that was built and timed
// in rounds 1-3 (three-pass predecessors, whole-image and persistent units, whole rows with a partner, ...) -- are in
// profiles/LAB_NOTES_imgpool.md; round 4's in DESIGN.md 5.2.
#include <cstdlib>
#include <hip/hip_ext.h>

#include "common.h"

namespace ptx {

typedef unsigned int u4u2 __attribute__((ext_vector_type(4), aligned(2)));    // 16-B load at 2-B alignment
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kPoolHeads = 8;
constexpr int kPoolNtImages = 4096;      // images per launch (~21 scenes of 196 views) from which the partials are stored as streaming lines (see the write-out)
constexpr int kPoolWPad = 544;          // LDS row stride (elements) of the split weights: rows 16 words apart mod 64
constexpr int kPoolPPad = 136;          // LDS row stride (elements) of the split numerators (128 pixels + 8)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// The exact three-way split x = x1 + x2 + x3 of an fp32 operand into the storage type of the features
// (DT 1 = bf16: truncation, 3 x 8 significant bits; DT 2 = fp16: round to nearest, 3 x 11 bits).  fp16 has
// fp32's precision to spare but not its range: fp16 operands are first scaled by a power of two (exact) so
// that the largest one sits at 2^13..2^14 -- the third part of anything that matters then stays above the
// fp16 subnormal resolution -- and the MFMA result is scaled back.
template <int DT>
__device__ __forceinline__ void split3(float x, unsigned short &q1, unsigned short &q2, unsigned short &q3)
{
    if (DT == 1) {
        const unsigned int u1 = __float_as_uint(x) & 0xffff0000u;
        const float r1 = x - __uint_as_float(u1);
        const unsigned int u2 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(u2);
        q1 = (unsigned short)(u1 >> 16); q2 = (unsigned short)(u2 >> 16); q3 = (unsigned short)(__float_as_uint(r2) >> 16);
    } else {
        const _Float16 h1 = (_Float16)x;
        const float r1 = x - (float)h1;
        const _Float16 h2 = (_Float16)r1;
        const float r2 = r1 - (float)h2;
        const _Float16 h3 = (_Float16)r2;
        q1 = __builtin_bit_cast(unsigned short, h1); q2 = __builtin_bit_cast(unsigned short, h2);
        q3 = __builtin_bit_cast(unsigned short, h3);
    }
}

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const u32x4 &a, const u32x4 &b, const f32x4 &c)
{
    if (DT == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// exponent k of the power-of-two scale 2^k that puts |x| <= amax at 2^13 .. 2^14 (0 for amax = 0 / non-finite)
__device__ __forceinline__ int pow2_scale_exp(float amax)
{
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;           // floor(log2 amax) for normal numbers
    return (amax > 0.0f && e > -100 && e < 100) ? 13 - e : 0;
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned int)(127 + k) << 23); }
constexpr float kPoolEScale = 16384.0f;       // fp16: softmax numerators (<= 1) are split at 2^14

// 128-bit logical right shift by k elements of 16 bits (k = 0..7)
__device__ __forceinline__ u32x4 shr_elems(const u32x4 &v, int k)
{
    const int ws = k >> 1;
    unsigned int w[8] = {v[0], v[1], v[2], v[3], 0u, 0u, 0u, 0u};
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned int lo = 0u, hi = 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (ws == s) { lo = w[q + s]; hi = w[q + s + 1]; }
        }
        o[q] = (k & 1) ? ((lo >> 16) | (hi << 16)) : lo;
    }
    return o;
}

struct PoolArgs {
    const unsigned short *img; const float *we, *qkv0;
    int nimg, in_dim, hw, C, KT1, EW; float scale;
    float *Gs;          // [nimg][2][heads][in_dim]  sum_p e_h(p) f(c,p) of the tile
    float *E;           // [nimg][heads][EW]         token 0: 1, token 1 + p: e_h(p) = exp(s_h(p) - m_tile), 0 beyond hw
    float *ML;          // [nimg][heads][5]          m_0, l_0, m_1, l_1, s_h(0) = scale * q_h . k0_h (mean token)
};
// These three are the A operand of the o-projection GEMM, which merges the two tiles and the mean token on the
// fly (split softmax, gemm.hip k_gemm32<4, 1>): m = max(m_0, m_1, s(0)), l = l_0 e^(m_0-m) + l_1 e^(m_1-m) + e^(s(0)-m),
// g = (G_0 e^(m_0-m) + G_1 e^(m_1-m)) / l, a(p) = e(p) e^(m_T-m) / l, a(0) = e^(s(0)-m) / l.

template <int DT, bool NT>   // DT: storage type of the features, 1 = bf16, 2 = fp16; NT: streaming stores of the partials
__global__ __launch_bounds__(512) void k_img_pool(PoolArgs a)
{
    constexpr int heads = kPoolHeads;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *partial = reinterpret_cast<float *>(smem);                               // [8 waves][heads][128]; later G [heads][in_dim]
    unsigned short *wpart = reinterpret_cast<unsigned short *>(partial + 8 * heads * 128);   // [24][kPoolWPad]
    unsigned short *parts = wpart + 24 * kPoolWPad;                                  // [24][kPoolPPad]
    const int in_dim = a.in_dim, hw = a.hw;
    // The two tiles of an image run back to back on the SAME XCD (work-groups go round-robin over the 8 XCDs
    // by id): every row has a cache line at the tile boundary, and the run-on lanes of tile 1 read the head of
    // the next row, which both must come out of one L2 -- with the tiles on different XCDs the same loads took
    // 50 us instead of 33 (scratch/pattern_bench.hip P0/P2).  Images in reverse order: the mean pass left
    // the last ones in cache.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int T = slot & 1, imr = (slot >> 1) * 8 + xcd;
    if (imr >= a.nimg) return;
    const int im = a.nimg - 1 - imr;
    const int slab = im * 2 + T;                                // Gs slab of this unit
    const int tid = threadIdx.x, lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const float *wim = a.we + (size_t)im * heads * a.KT1;
    const unsigned short *f = a.img + (size_t)im * in_dim * hw;
    // This lane's window of 8 pixels.  A window that sticks out of the row simply runs on into the next row
    // (rows are contiguous; those lines are wanted anyway) and the surplus columns are ignored -- except in
    // the very last row of the tensor, where the load is moved back inside and the data shifted into place.
    const int px = 128 * T + 8 * n;
    const bool full = px + 8 <= hw;
    const int cw = 64 * wid;                                    // this wave's 64 channels = two blocks of 32
    // the head weights of this image are requested BEFORE the tile: loads return in order, so they are there (and
    // split into LDS) while the 16 tile loads are still in flight instead of after the last of them
    float wv[heads];
#pragma unroll
    for (int h = 0; h < heads; ++h) wv[h] = wim[(size_t)h * a.KT1 + tid];
    u32x4 L[2][8];
    const bool tensor_end = im == a.nimg - 1 && cw + 64 == in_dim;    // wave-uniform
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const unsigned short *r = f + (size_t)(cw + 32 * kb + 8 * kq) * hw + px;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (kb == 1 && i == 7 && tensor_end) {
                const bool lastrow = kq == 3 && !full;
                const int back = lastrow ? px + 8 - hw : 0;     // elements moved back (>= 8: wholly beyond the row)
                const int bk = back > 8 ? px - (hw - 8) : back;
                u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(r + (size_t)i * hw - bk));
                L[kb][i] = lastrow ? (back >= 8 ? u32x4{0u, 0u, 0u, 0u} : shr_elems(v, back)) : v;
            } else {
                L[kb][i] = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(r + (size_t)i * hw));
            }
        }
    }
    // positional score terms of head `wid` for this tile's pixels (lane, lane + 64)
    float ev[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int p = 128 * T + lane + 64 * u;
        ev[u] = p < hw ? wim[(size_t)wid * a.KT1 + in_dim + 1 + p] : 0.0f;
    }
    // score of the mean token, s_h(0) = scale * q_h . k0_h (wave = head; published by the tile-0 unit)
    // (only the loads here: reducing right away would wait for them -- and, in order, for all the feature loads
    // above -- before the weight loads below are even issued)
    float sq = 0.0f, sk = 0.0f;
    if (T == 0) {
        const int hd = a.C / heads;
        const float *qv = a.qkv0 + (size_t)im * 3 * a.C + wid * hd;
        if (lane < hd) { sq = qv[lane]; sk = qv[a.C + lane]; }
    }
    // head weights of this image -> three 16-bit parts in LDS.  Thread = channel, so wave w splits exactly the 64
    // channels it contracts in stage 1 (in_dim = 512 = one channel per thread) and, for fp16, scales them by a
    // power of two of its own: no other wave needs to know it (all loads first).
    // (the inverse scales live in the padding of the weight rows in LDS, 4 B per (head, wave): eight more live
    // registers through stage 1 would cost the second work-group per CU)
    {
        const int c = tid;                                      // in_dim == 512 == threads (img_pool_supported)
#pragma unroll
        for (int h = 0; h < heads; ++h) {
            if (DT == 2) {
                const int k = pow2_scale_exp(wave_max(fabsf(wv[h])));
                wv[h] *= pow2f(k);
                if (lane == 0) *reinterpret_cast<float *>(wpart + (size_t)h * kPoolWPad + in_dim + 2 * wid) = pow2f(-k);
            }
            unsigned short q1, q2, q3;
            split3<DT>(wv[h], q1, q2, q3);
            wpart[(size_t)h * kPoolWPad + c] = q1;
            wpart[(size_t)(8 + h) * kPoolWPad + c] = q2;
            wpart[(size_t)(16 + h) * kPoolWPad + c] = q3;
        }
    }
    __syncthreads();

    // ---- 1. scores of this wave's 64 channels: D rows 0-7 = heads, column n of MFMA j = pixel 8n + j
    {
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool lo = n < heads;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            u32x4 af[3];
#pragma unroll
            for (int pt = 0; pt < 3; ++pt) {
                u32x4 t = *reinterpret_cast<const u32x4 *>(wpart + (size_t)(pt * 8 + (n & 7)) * kPoolWPad + cw + 32 * kb + 8 * kq);
                if (!lo) t = u32x4{0u, 0u, 0u, 0u};
                af[pt] = t;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned int sel = (j & 1) ? 0x07060302u : 0x05040100u;
                u32x4 b;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) b[qd] = __builtin_amdgcn_perm(L[kb][2 * qd + 1][j >> 1], L[kb][2 * qd][j >> 1], sel);
#pragma unroll
                for (int pt = 0; pt < 3; ++pt) acc[j] = mfma16<DT>(af[pt], b, acc[j]);
            }
        }
        if (kq < 2) {                                           // rows 4 kq + r = heads
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float us = 1.0f;                                // undo this wave's weight scale
                if (DT == 2) us = *reinterpret_cast<const float *>(wpart + (size_t)(4 * kq + r) * kPoolWPad + in_dim + 2 * wid);
                float *d = partial + ((size_t)wid * heads + 4 * kq + r) * 128 + 8 * n;
                *reinterpret_cast<float4 *>(d) = make_float4(acc[0][r] * us, acc[1][r] * us, acc[2][r] * us, acc[3][r] * us);
                *reinterpret_cast<float4 *>(d + 4) = make_float4(acc[4][r] * us, acc[5][r] * us, acc[6][r] * us, acc[7][r] * us);
            }
        }
    }
    __syncthreads();

    // ---- 2. wave = head: sum the eight channel slices (fixed order), tile-local softmax numerators
    {
        const int h = wid;
        float e[2], mloc = -INFINITY, sv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = lane + 64 * u;
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += partial[((size_t)w * heads + h) * 128 + p];
            sv[u] = 128 * T + p < hw ? s + ev[u] : -INFINITY;
            mloc = fmaxf(mloc, sv[u]);
        }
        const float m = wave_max(mloc);
#pragma unroll
        for (int u = 0; u < 2; ++u) e[u] = 128 * T + lane + 64 * u < hw ? expf(sv[u] - m) : 0.0f;
        const float l = wave_sum(e[0] + e[1]);
        float *erow = a.E + ((size_t)im * heads + h) * a.EW;
        const float s0 = T == 0 ? wave_sum(sq * sk) * a.scale : 0.0f;
        if (lane == 0) {
            float *ml = a.ML + ((size_t)im * heads + h) * 5;
            ml[2 * T] = m;
            ml[2 * T + 1] = l;
            if (T == 0) { ml[4] = s0; erow[0] = 1.0f; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = lane + 64 * u;
            const int t = 1 + 128 * T + p;                      // token of this pixel; the last tile also clears the padding
            if (t <= hw || (T == (hw > 128 ? 1 : 0) && t < a.EW)) erow[t] = t <= hw ? e[u] : 0.0f;
            unsigned short q1, q2, q3;
            split3<DT>(DT == 2 ? e[u] * kPoolEScale : e[u], q1, q2, q3);
            parts[(size_t)h * kPoolPPad + p] = q1;
            parts[(size_t)(8 + h) * kPoolPPad + p] = q2;
            parts[(size_t)(16 + h) * kPoolPPad + p] = q3;
        }
    }
    __syncthreads();

    // ---- 3. weighted sums over the tile's pixels from the registers.  Load map: lane (n, kq) register i =
    // row 8 kq + i, window n.  MFMA map: A row m = (channel sub-block cr = m >> 2, window set ps = m & 3),
    // K-lane kq' = windows ps + 4 kq'; B column = (head hh = m >> 2 of the group, window set ps).  Lane
    // (m, kq') therefore takes register i of lane (n = ps + 4 kq', kq = cr): one ds_bpermute per dword.
    // r04: ONE basic block per half of the channels.  The r03 form ran a (register, head group) step at a time -- four
    // bpermutes, a wait, three DEPENDENT matrix instructions, a ?: chain that hipcc turned into nested exec-mask regions with
    // branches, a masked LDS store -- 32 strictly sequential steps of ~280 cycles, 3.3 of a unit's 12.3 us.  Now the eight
    // registers of a half are permuted in place up front (32 bpermutes in flight, one wait), four independent accumulator
    // chains (two registers x two head groups) are interleaved so that the matrix pipe never waits for its own result, the
    // lane's element of an accumulator is picked with per-lane bit masks, and the finished sums of eight consecutive
    // channels go to LDS as two 16-byte pieces per head group.
    {
        const int ps = n & 3, hh = n >> 2;                      // also cr = n >> 2 for the A side
        const int src = ((ps + 4 * kq) + 16 * (n >> 2)) * 4;    // byte index of the source lane
        u32x4 bfr[2][3];
#pragma unroll
        for (int hg = 0; hg < 2; ++hg)
#pragma unroll
            for (int pt = 0; pt < 3; ++pt)
                bfr[hg][pt] = *reinterpret_cast<const u32x4 *>(parts + (size_t)(pt * 8 + 4 * hg + hh) * kPoolPPad + 8 * (ps + 4 * kq));
        const unsigned m0 = ps == 0 ? ~0u : 0u, m1 = ps == 1 ? ~0u : 0u, m2 = ps == 2 ? ~0u : 0u, m3 = ps == 3 ? ~0u : 0u;
        float *G = partial;                                     // [heads][in_dim]; the slices are dead (barrier above)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int d = 0; d < 4; ++d) L[kb][i][d] = (unsigned int)__builtin_amdgcn_ds_bpermute(src, (int)L[kb][i][d]);
            float xs[2][8];
#pragma unroll
            for (int ig = 0; ig < 4; ++ig) {
                f32x4 c[2][2];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int hg = 0; hg < 2; ++hg) c[ii][hg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int hg = 0; hg < 2; ++hg) c[ii][hg] = mfma16<DT>(L[kb][2 * ig + ii], bfr[hg][pt], c[ii][hg]);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int hg = 0; hg < 2; ++hg) {
                        // D[row 4 kq + r][col n]: row = (channel sub-block kq, set r), col = (head hh, set ps): keep r == ps
                        const f32x4 &acc = c[ii][hg];
                        float x = __uint_as_float((__float_as_uint(acc[0]) & m0) | (__float_as_uint(acc[1]) & m1) |
                                                  (__float_as_uint(acc[2]) & m2) | (__float_as_uint(acc[3]) & m3));
                        x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                        x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                        xs[hg][2 * ig + ii] = DT == 2 ? x * (1.0f / kPoolEScale) : x;
                    }
            }
            // channels cw + 32 kb + 8 kq + (0..7) of head 4 hg + hh: one lane of the quad parks them in LDS (the score slices
            // are dead: barrier above).  (r04: stored straight from these registers -- 16 lanes x 32 B per instruction, half lines --
            // the launch was 2 us SLOWER at 4 scenes and 507 instead of 369 us at 32: the write-through path wants whole lines.)
            if (ps == 0) {
#pragma unroll
                for (int hg = 0; hg < 2; ++hg)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        *reinterpret_cast<f32x4 *>(G + (size_t)(4 * hg + hh) * in_dim + cw + 32 * kb + 8 * kq + 4 * q) =
                            f32x4{xs[hg][4 * q], xs[hg][4 * q + 1], xs[hg][4 * q + 2], xs[hg][4 * q + 3]};
            }
        }
    }
    __syncthreads();
    {
        const float *G = partial;
        float *dst = a.Gs + (size_t)slab * heads * in_dim;
        for (int i = tid * 4; i < heads * in_dim; i += 512 * 4)
        {   // write-through (sc1): the 25 MB of partials must not sit dirty in the L2s when the launch ends -- the
            // kernel boundary pays ~1 us per 6 MB of dirty lines (MI355X_MICROARCH.md, "boundary")
            // (NT: with many images -- 40 KB of partials each, 250 MB at 32 scenes -- the write-through lines are also marked
            // streaming: -11 % on this launch and +3-5 % on the step at 32 scenes per GPU, +1.7 % at 24, nothing at 12 and 16,
            // -0.5 % at 4 and 8 where the consumer still finds them in the caches (interleaved A/B on one box, PTX_POOL_NT):
            // chosen per launch from the image count)
            const f32x4 v = *reinterpret_cast<const f32x4 *>(G + i);
            if (NT) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(dst + i), "v"(v) : "memory");
            else    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");
        }
    }
}

bool img_pool_supported(int dt, int in_dim, int hw, int heads)
{
    return (dt == 1 || dt == 2) && heads == kPoolHeads && in_dim == 512 && hw > 128 && hw <= 255;
}

size_t img_pool_bytes(int nimg, int in_dim, int EW)
{
    return (size_t)nimg * kPoolHeads * (2 * (size_t)in_dim + EW + 8) * sizeof(float);
}

void img_pool_layout(float *scratch, int nimg, int in_dim, int EW, float **Gs, float **E, float **ML)
{
    *Gs = scratch;
    *E = *Gs + (size_t)nimg * 2 * kPoolHeads * in_dim;
    *ML = *E + (size_t)nimg * kPoolHeads * EW;
}

// Gs / E / ML: the three result arrays of img_pool_layout(), already offset to the first image of this launch
int launch_img_pool(const void *img, int dt, const float *we, const float *qkv0, int nimg, int in_dim, int hw, int C,
                    int KT1, int EW, float scale, float *Gs, float *E, float *ML, hipStream_t st)
{
    PTX_REQUIRE(EW % 4 == 0 && EW >= hw + 1 && hw > 128 && hw <= 255, "img pool: hw=%d EW=%d", hw, EW);
    PoolArgs pa{static_cast<const unsigned short *>(img), we, qkv0, nimg, in_dim, hw, C, KT1, EW, scale, Gs, E, ML};
    const size_t lds = sizeof(float) * 8 * kPoolHeads * 128 + sizeof(unsigned short) * 24 * (kPoolWPad + kPoolPPad);
    PTX_REQUIRE(lds <= 64 * 1024, "img pool: %zu B of LDS", lds);
    static const int nt_env = getenv("PTX_POOL_NT") ? atoi(getenv("PTX_POOL_NT")) : -1;      // A/B runs: 0 / 1 force it
    const bool nt = nt_env >= 0 ? nt_env != 0 : nimg >= kPoolNtImages;
    const dim3 grid(cdiv(nimg, 8) * 16);
    // (bench.py's roofline leg: the timing events ride on the kernel's own packet -- begin / end of the kernel, like a kernel trace)
    hipEvent_t ta = nullptr, tb = nullptr;
    (void)timing_ext_take(&ta, &tb);
    if (dt == 1) { if (nt) hipExtLaunchKernelGGL((k_img_pool<1, true>), grid, dim3(512), lds, st, ta, tb, 0, pa); else hipExtLaunchKernelGGL((k_img_pool<1, false>), grid, dim3(512), lds, st, ta, tb, 0, pa); }
    else         { if (nt) hipExtLaunchKernelGGL((k_img_pool<2, true>), grid, dim3(512), lds, st, ta, tb, 0, pa); else hipExtLaunchKernelGGL((k_img_pool<2, false>), grid, dim3(512), lds, st, ta, tb, 0, pa); }
    PTX_LAUNCHED("k_img_pool");
    return PTX_OK;
}

}  // namespace ptx
