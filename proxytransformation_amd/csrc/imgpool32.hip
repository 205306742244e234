// Attention pooling of fp32-stored image features in ONE pass over the features after the mean (r05; AttentionPool2d query 0,
// PRE:158-176; algebra in imgproxy.hip).  fp32 is the reference's own layout (DET:372-377) and cfg4's; until r05 it took THREE
// streaming passes (k_img_mean, k_img_scores, k_img_gather).  This kernel replaces the last two -- the 16-bit kernel's
// decomposition (imgpool.hip) with the arithmetic an fp32 operand allows:
//
// A work unit is (image, half of the pixels): FOUR tiles of 32 pixels x 512 channels x 4 B = 64 KB, held one after the other in
// the REGISTERS of one 8-wave work-group (8 loads of 16 B per lane: lane (n, kq) register i = channel 64 w + 8 i + kq, pixels
// 4 n .. 4 n + 3 -- eight rows x 128 contiguous bytes per instruction) and each used twice while it is there.  (64-pixel tiles --
// 64 tile registers per lane -- did not leave room for two work-groups per CU: 208 registers as written, 52 - 85 spilled when forced
// to 128; one work-group per CU ran the launch in 169 us, no faster than the two passes it replaces.)
//   1. scores   s_h(p) = sum_c w_h(c) f(c,p): plain FMAs (exact fp32 products, like k_img_scores); the eight head weights of a
//      lane's channel come from LDS, the eight channel rows of a pixel quad meet through one DPP rotation and two cross-row exchanges, the eight
//      waves' channel slices in LDS
//   2. online softmax over the unit's four tiles (wave = head, lane = pixel): running maximum / sum, the earlier tiles' numerators
//      kept in registers and rescaled, the weighted sums of stage 3 rescaled per head
//   3. weighted sums  G_h(c) += sum_p e_h(p) f(c,p) on v_mfma_f32_16x16x4_f32 (exact products).  The contraction runs over
//      PIXELS, which the load map spreads over the lanes of a row -- an MFMA contracts over lane >> 4 only -- so each wave passes
//      its tile through a wave-private 2 KB LDS patch, 16 channels at a time: written in the load map, read back as the A
//      operand (row = lane & 15 = channel, k = lane >> 4 = pixel; rows padded to 36 floats: conflict-free both ways), B = the
//      numerators; four accumulators per wave (its four channel groups) live across the four tiles.
//      (First form, r05: the tile LOADED in the A-operand map -- 16 rows x 64 B per instruction.  Parity-green and slow: 211 us per
//      launch against 167 for the two passes it replaces; with every stage's arithmetic cut to a quarter still 133 us -- the map
//      itself streams at 2.7 TB/s -- and the 16-lane DPP reductions of stage 1 cost another 50 us.)
// The unit writes what a PAIR of the 16-bit kernel's units writes for these 128 pixels -- (m, l) of the pair, the numerators
// relative to m, G -- so the o-projection GEMM merges the two halves of an image with the mean token exactly as it does for the
// 16-bit features (gemm.hip, k_gemm32<4, 1>); nothing downstream changes.
#include <hip/hip_ext.h>

#include "common.h"

namespace ptx {

typedef float p32_f4u __attribute__((ext_vector_type(4), aligned(4)));    // 16-B load at 4-B alignment (rows are hw * 4 B apart)
typedef float p32_f4 __attribute__((ext_vector_type(4)));

constexpr int kP32Heads = 8;
constexpr int kP32Tile = 32;            // pixels per tile: 8 quads (lane & 7) x 8 channel rows (lane >> 3) per load instruction
constexpr int kP32Tiles = 128 / kP32Tile;
constexpr int kP32TRow = 36;            // floats per row of a transpose patch: 32 pixels + 4 (rows 4 banks apart: reads and writes conflict-free)

struct Pool32Args {
    const float *img; const float *we, *qkv0;
    int nimg, in_dim, hw, C, KT1, EW; float scale;
    float *Gs, *E, *ML;                 // the layout of imgpool.hip's PoolArgs (img_pool_layout)
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_img_pool32(Pool32Args a)
{
    constexpr int heads = kP32Heads, TP = kP32Tile;
    extern __shared__ __attribute__((aligned(16))) float p32_sm[];
    float *W_s = p32_sm;                                // [512 channels][8 heads]; after the tiles: G [8 heads][512]
    float *S_s = W_s + 512 * heads;                     // [8 waves][8 heads][32 pixels] partial scores of the waves' channel slices
    float *E_s = S_s + 8 * heads * TP;                  // [8 heads][32] numerators of the current tile
    float *sc_s = E_s + heads * TP;                     // [16] per head: exp(m_old - m_new) of the current tile (8 .. 15: unused columns)
    float *T_s = sc_s + 16;                             // [8 waves][16 channels][36]: the waves' transpose patches (stage 3)
    const int in_dim = a.in_dim, hw = a.hw;
    // the two halves of an image back to back on the same XCD, images in reverse order (imgpool.hip)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int H = slot & 1, imr = (slot >> 1) * 8 + xcd;
    if (imr >= a.nimg) return;
    const int im = a.nimg - 1 - imr;
    const int tid = threadIdx.x, lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 7, kq = lane >> 3;                     // load map: pixel quad, channel row
    const int mr = lane & 15, mk = lane >> 4;                   // MFMA map: row (channel) / column (head), k (pixel quad)
    const float *wim = a.we + (size_t)im * heads * a.KT1;
    const float *f = a.img + (size_t)im * in_dim * hw;
    float *T_w = T_s + (size_t)wid * (16 * kP32TRow);           // this wave's transpose patch
    // ---- prologue requests: head weights of channel `tid`, positional terms of head `wid`, the mean token's q . k0
    float wv[heads];
#pragma unroll
    for (int h = 0; h < heads; ++h) wv[h] = wim[(size_t)h * a.KT1 + tid];
    float ev[kP32Tiles];
#pragma unroll
    for (int t = 0; t < kP32Tiles; ++t) {
        const int p = 128 * H + TP * t + (lane & (TP - 1));
        ev[t] = p < hw ? wim[(size_t)wid * a.KT1 + in_dim + 1 + p] : 0.0f;
    }
    float sq = 0.0f, sk = 0.0f;
    if (H == 0) {
        const int hd = a.C / heads;
        const float *qv = a.qkv0 + (size_t)im * 3 * a.C + wid * hd;
        if (lane < hd) { sq = qv[lane]; sk = qv[a.C + lane]; }
    }
    // the rows of this lane: channel 64 wid + 8 i + kq, as 32-bit byte offsets from the image's base (an image is 460 KB; as 64-bit
    // row pointers the addresses alone took 32 registers).  The very last row of the tensor must not be read past its end: the
    // quad that straddles the row's end is loaded from four floats further back and shifted (only that lane, only in that unit)
    const char *fb = reinterpret_cast<const char *>(f);
    const unsigned row_off = (unsigned)(64 * wid + kq) * (unsigned)hw * 4u, row_step = 8u * (unsigned)hw * 4u;   // register i: + i row_step
    const bool tensor_end = im == a.nimg - 1 && wid == 7 && kq == 7;      // (i == 7 below)
    {
        *reinterpret_cast<p32_f4 *>(W_s + tid * heads) = p32_f4{wv[0], wv[1], wv[2], wv[3]};
        *reinterpret_cast<p32_f4 *>(W_s + tid * heads + 4) = p32_f4{wv[4], wv[5], wv[6], wv[7]};
        if (tid < 16) sc_s[tid] = 0.0f;
    }
    p32_f4 D[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) D[q] = p32_f4{0.f, 0.f, 0.f, 0.f};
    // wave = head, lanes 0 .. 31 = the tile's pixels: running maximum / sum, the numerators of the tiles so far (rescaled as the maximum moves)
    float m_run = -INFINITY, l_run = 0.0f, e_t[kP32Tiles];
#pragma unroll
    for (int t = 0; t < kP32Tiles; ++t) e_t[t] = 0.0f;

#pragma unroll
    for (int t = 0; t < kP32Tiles; ++t) {
        const int px0 = 128 * H + TP * t;
        // ---- the tile: 8 loads per lane, all requested before the first is used
        p32_f4u L[8];
        {
            const int p = px0 + 4 * n;
            const int pc = p < hw ? p : 0;                      // quads wholly beyond the row re-read its first quad (finite; numerators 0)
            const unsigned off0 = row_off + (unsigned)pc * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float *rowp = reinterpret_cast<const float *>(fb + (off0 + (unsigned)i * row_step)) - pc;
                if (i == 7 && tensor_end && pc + 4 > hw) {
                    const p32_f4u v = *reinterpret_cast<const p32_f4u *>(rowp + hw - 4);
                    const int back = pc + 4 - hw;               // 1 .. 3 floats moved back
                    L[i] = back == 1 ? p32_f4u{v[1], v[2], v[3], 0.f} : back == 2 ? p32_f4u{v[2], v[3], 0.f, 0.f} : p32_f4u{v[3], 0.f, 0.f, 0.f};
                } else {
                    L[i] = __builtin_nontemporal_load(reinterpret_cast<const p32_f4u *>(rowp + pc));
                }
            }
        }
        __syncthreads();                                        // W_s (first trip) / S_s free again (later trips)
        // ---- 1. scores of this wave's 64 channels for the tile's 32 pixels (four heads at a time: 16 accumulators)
#pragma unroll
        for (int hb = 0; hb < heads; hb += 4) {
            float acc[4][4];
#pragma unroll
            for (int h = 0; h < 4; ++h) { acc[h][0] = acc[h][1] = acc[h][2] = acc[h][3] = 0.0f; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const p32_f4 w4 = *reinterpret_cast<const p32_f4 *>(W_s + (64 * wid + 8 * i + kq) * heads + hb);
                const p32_f4u v = L[i];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    acc[h][0] = fmaf(w4[h], v[0], acc[h][0]); acc[h][1] = fmaf(w4[h], v[1], acc[h][1]);
                    acc[h][2] = fmaf(w4[h], v[2], acc[h][2]); acc[h][3] = fmaf(w4[h], v[3], acc[h][3]);
                }
            }
            // the eight channel rows of a pixel quad (lanes 8, 16, 32 apart), fixed order
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = acc[h][c];
                    x += PTX_ROR_F(x, 8);
                    x += __shfl_xor(x, 16, 64);
                    x += __shfl_xor(x, 32, 64);
                    acc[h][c] = x;
                }
            if (kq == 0) {
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    *reinterpret_cast<p32_f4 *>(S_s + ((size_t)wid * heads + hb + h) * TP + 4 * n) = p32_f4{acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
            }
            __builtin_amdgcn_sched_barrier(0);                  // keep the next group's weight reads from being hoisted (registers)
        }
        __syncthreads();
        // ---- 2. wave = head, lane = pixel (lanes 32 .. 63 idle): the eight channel slices in a fixed order, online softmax over the tiles
        {
            const int h = wid, pl = lane & (TP - 1);
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += S_s[((size_t)w * heads + h) * TP + pl];
            const bool valid = lane < TP && px0 + pl < hw;
            const float sv = valid ? s + ev[t] : -INFINITY;
            const float m_new = fmaxf(m_run, wave_max(sv));
            const float resc = expf(m_run - m_new);              // 0 on the first tile (m_run = -inf)
            const float e = valid ? expf(sv - m_new) : 0.0f;
            l_run = l_run * resc + wave_sum(e);
            m_run = m_new;
#pragma unroll
            for (int u = 0; u < kP32Tiles; ++u) e_t[u] = u < t ? e_t[u] * resc : (u == t ? e : 0.0f);
            if (lane < TP) E_s[h * TP + lane] = e;
            if (lane == 0) sc_s[h] = resc;
            if (t == kP32Tiles - 1) {                            // the unit's results of this head
                float *erow = a.E + ((size_t)im * heads + h) * a.EW;
                // (numerators of pixels beyond the row are 0: the second half also clears the padding behind the row's last token)
#pragma unroll
                for (int u = 0; u < kP32Tiles; ++u) {
                    const int tok = 1 + 128 * H + TP * u + lane;
                    if (lane < TP && tok < a.EW) erow[tok] = e_t[u];
                }
                const float s0 = H == 0 ? wave_sum(sq * sk) * a.scale : 0.0f;
                if (lane == 0) {
                    float *ml = a.ML + ((size_t)im * heads + h) * 5;
                    ml[2 * H] = m_run; ml[2 * H + 1] = l_run;
                    if (H == 0) { ml[4] = s0; erow[0] = 1.0f; }
                }
            }
        }
        __syncthreads();
        // ---- 3. G_h(c) = G_h(c) * exp(m_old - m_new) + sum_p e_h(p) f(c,p): D[row 4 mk + r = channel][col mr = head]
        {
            const float resc = sc_s[mr];
#pragma unroll
            for (int q = 0; q < 4; ++q) { D[q][0] *= resc; D[q][1] *= resc; D[q][2] *= resc; D[q][3] *= resc; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // channels 64 w + 16 q + cl, cl = 8 (i & 1) + kq: the load map's two registers into the patch ...
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
                    *reinterpret_cast<p32_f4 *>(T_w + (8 * ii + kq) * kP32TRow + 4 * n) = p32_f4{L[2 * q + ii][0], L[2 * q + ii][1], L[2 * q + ii][2], L[2 * q + ii][3]};
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                // ... and back as the A operand: lane (row mr = channel, mk = pixel quad of a 16-pixel block)
#pragma unroll
                for (int pb = 0; pb < TP / 16; ++pb) {
                    const p32_f4 a4 = *reinterpret_cast<const p32_f4 *>(T_w + mr * kP32TRow + 16 * pb + 4 * mk);
                    p32_f4 e4 = *reinterpret_cast<const p32_f4 *>(E_s + (mr & 7) * TP + 16 * pb + 4 * mk);
                    if (mr >= heads) e4 = p32_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) D[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c], e4[c], D[q], 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    // ---- G of the unit: through LDS (whole write-through lines, imgpool.hip) into slab (im, H)
    __syncthreads();                                            // every wave is done with W_s (stage 1 of the last tile)
    float *G = W_s;
    if (mr < heads) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<p32_f4 *>(G + (size_t)mr * in_dim + 64 * wid + 16 * q + 4 * mk) = D[q];
    }
    __syncthreads();
    {
        float *dst = a.Gs + (size_t)(im * 2 + H) * heads * in_dim;
        for (int i = tid * 4; i < heads * in_dim; i += 512 * 4) {
            const p32_f4 v = *reinterpret_cast<const p32_f4 *>(G + i);
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i), "v"(v) : "memory");
        }
    }
}

bool img_pool32_supported(int dt, int in_dim, int hw, int heads)
{
    return dt == 0 && heads == kP32Heads && in_dim == 512 && hw > 128 && hw <= 255;
}

int launch_img_pool32(const float *img, const float *we, const float *qkv0, int nimg, int in_dim, int hw, int C, int KT1, int EW,
                      float scale, float *Gs, float *E, float *ML, hipStream_t st)
{
    PTX_REQUIRE(EW % 4 == 0 && EW >= hw + 1 && hw > 128 && hw <= 255 && in_dim == 512, "img pool (fp32): hw=%d EW=%d in_dim=%d", hw, EW, in_dim);
    PTX_REQUIRE((reinterpret_cast<uintptr_t>(img) & 15) == 0, "img_feat must be 16-byte aligned");
    Pool32Args pa{img, we, qkv0, nimg, in_dim, hw, C, KT1, EW, scale, Gs, E, ML};
    const size_t lds = sizeof(float) * (512 * kP32Heads + 8 * kP32Heads * kP32Tile + kP32Heads * kP32Tile + 16 + 8 * 16 * kP32TRow);
    const dim3 grid(cdiv(nimg, 8) * 16);
    // (bench.py's roofline leg: the timing events ride on the kernel's own packet, like k_img_pool's)
    hipEvent_t ta = nullptr, tb = nullptr;
    (void)timing_ext_take(&ta, &tb);
    hipExtLaunchKernelGGL(k_img_pool32, grid, dim3(512), lds, st, ta, tb, 0, pa);
    PTX_LAUNCHED("k_img_pool32");
    return PTX_OK;
}

}  // namespace ptx
