// Proxy attention (PRE:230-250) on the matrix cores, head_dim = 32 (the reference's 256 / 8) or 64 (embed_dim = 512).
//
// One kernel serves both contractions of ProxyAttention:
//   proxy as query : O = softmax_n((P*scale) K^T) V          queries = proxies, keys = cluster tokens
//   proxy as key   : O = softmax_L(mask((Q*scale) P^T)) PV   queries = tokens, keys = proxies
// A work-group owns 32 query rows of one (scene, head); its 4 waves split the key tiles (32 keys
// each) four ways, every wave runs an online softmax over its share, and the four partial
// (max, sum, O) triples are merged in fixed wave order through LDS (the chain of dependent
// tiles, not the arithmetic, is what a launch of this size waits for).  Per wave everything
// stays in registers:
//   S^T = K Q^T      v_mfma_f32_32x32x2_f32, A = key rows, B = query rows -> lane owns ONE query
//                    (col = lane & 31) and 16 of the 32 key scores of the tile, so the row
//                    max / sum of the softmax are in-lane reductions plus one xor-32 exchange;
//   O^T += V^T P^T   A = V^T (lane = output dim), B = the probabilities exactly where the first
//                    MFMA left them (key index of step s, half hh = (s&3) + 8*(s>>2) + 4*hh),
//                    so P never moves between lanes and the running rescale is per lane.
// fp32 in / fp32 accumulate: this is the parity configuration.
// (r02, 32 scenes per GPU, 113 us for both launches = 24 % of the fp32 matrix peak: neither K rows fetched coalesced and
// turned into the row-per-lane fragment through wave-private LDS (115 us), nor four waves per SIMD (128 VGPRs, 24 B of
// spill: 110 us) move it; the hardware exponential for the numerators did, 122 -> 113 us.)
#include <cstdlib>

#include "common.h"

namespace ptx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NW waves per work-group share the key tiles of one query tile: 4 for the shapes of the benchmark (<= 8 key tiles);
// 8 when there are many keys and few work-groups (proxies as queries over the 691 tokens of the reference's own
// gs = 12 configuration: 22 dependent key tiles on 32 work-groups otherwise)
template <int HD, int NW>      // HD 32 or 64: HD / 2 contraction steps for the scores, HD / 32 accumulators for O^T
__global__ __launch_bounds__(NW * 64) void k_attn32(AttnBatch ab)
{
    constexpr int NS = HD / 2;          // MFMA steps of S^T = K Q^T (each contracts 2 dims: halves hh = 0, 1)
    constexpr int NA = HD / 32;         // 32-row blocks of O^T
    const AttnProb p = ab.p[blockIdx.z];           // by value: fields live in SGPRs
    const int lane = lane_id();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * 32;
    if (q0 >= p.nq) return;
    const int ntile = (p.nk + 31) >> 5, tper = (ntile + NW - 1) / NW;
    const int kbeg = wv * tper * 32, kend = min(p.nk, (wv + 1) * tper * 32);   // this wave's keys
    const int b = blockIdx.y / ab.heads, h = blockIdx.y - b * ab.heads;
    const int li = lane & 31, hh = lane >> 5;
    const float *Q = p.Q + (size_t)b * p.sQ + h * HD;
    const float *Kp = p.K + (size_t)b * p.sK + h * HD;
    const float *Vp = p.V + (size_t)b * p.sV + h * HD;
    const uint8_t *mask = p.mask ? p.mask + (size_t)b * p.nk : nullptr;

    // B operand of S^T: this lane's query row, dims hh*NS .. hh*NS+NS-1, pre-scaled (PRE:232, 241)
    float qf[NS];
    {
        const int qi = q0 + li;
        if (qi < p.nq) {
            const float4 *src = reinterpret_cast<const float4 *>(Q + (size_t)qi * p.ldq + hh * NS);
#pragma unroll
            for (int i = 0; i < NS / 4; ++i) {
                const float4 t = src[i];
                qf[4 * i] = t.x * ab.scale; qf[4 * i + 1] = t.y * ab.scale;
                qf[4 * i + 2] = t.z * ab.scale; qf[4 * i + 3] = t.w * ab.scale;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) qf[i] = 0.0f;
        }
    }
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[a][i] = 0.0f;

    // key / value fragments of tile k0 (A operands): kf = K[k0 + li][hh*NS .. +NS-1],
    // vf[a*16 + s] = V[k0 + key(s,hh)][32 a + li] with key(s,hh) = (s&3) + 8*(s>>2) + 4*hh
    auto load_tile = [&](int k0, float (&kf)[NS], float (&vf)[16 * NA]) {
        const int ki = k0 + li;
        if (ki < p.nk) {
            const float4 *src = reinterpret_cast<const float4 *>(Kp + (size_t)ki * p.ldk + hh * NS);
#pragma unroll
            for (int i = 0; i < NS / 4; ++i) {
                const float4 t = src[i];
                kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) kf[i] = 0.0f;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int key = k0 + (s & 3) + 8 * (s >> 2) + 4 * hh;
                vf[a * 16 + s] = key < p.nk ? Vp[(size_t)key * p.ldv + 32 * a + li] : 0.0f;
            }
    };
    float kf[NS], vf[16 * NA], kn[NS], vn[16 * NA];
    if (kbeg < kend) load_tile(kbeg, kf, vf);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        // software pipeline: the next tile's loads are in flight while this tile is consumed
        // (freshly written K/V come from another XCD's L2 or HBM: ~1 us per dependent round trip)
        const bool more = k0 + 32 < kend;
        if (more) load_tile(k0 + 32, kn, vn);
        f32x16 sc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sc, 0, 0, 0);
        // sc[r] = score(key = k0 + (r&3) + 8*(r>>2) + 4*hh, query = q0 + li)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            float v = sc[r];
            if (key >= p.nk) v = -INFINITY;
            else if (mask && mask[key] == 0) v = -1e9f;               // masked_fill(-1e9), PRE:247
            sc[r] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);                       // finite: every tile has a valid key
        // (the numerators use the hardware exponential, exp2(x log2 e): ~1e-7 relative, against the 5e-5 bar of the
        //  float stages; 16 of them per lane and key tile were a third of a tile's instruction time)
        const float alpha = expf(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = __expf(sc[r] - m_new); sc[r] = e; psum += e; }
        psum += __shfl_xor(psum, 32, 64);
        l_run = fmaf(l_run, alpha, psum);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int i = 0; i < 16; ++i) o[a][i] *= alpha;
#pragma unroll
            for (int s = 0; s < 16; ++s) o[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[a * 16 + s], sc[s], o[a], 0, 0, 0);
        }
        m_run = m_new;
        if (more) {
#pragma unroll
            for (int i = 0; i < NS; ++i) kf[i] = kn[i];
#pragma unroll
            for (int i = 0; i < 16 * NA; ++i) vf[i] = vn[i];
        }
    }
    // ---- merge the NW key slices in fixed order: m = max m_w, O = sum_w O_w exp(m_w - m), l likewise
    __shared__ float s_ml[NW - 1][2][64];
    __shared__ __attribute__((aligned(16))) float s_o[NW - 1][16 * NA][64];
    if (wv > 0) {
        s_ml[wv - 1][0][lane] = m_run;
        s_ml[wv - 1][1][lane] = l_run;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_o[wv - 1][a * 16 + r][lane] = o[a][r];
    }
    __syncthreads();
    if (wv > 0) return;
    float m_all = m_run;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) m_all = fmaxf(m_all, s_ml[w][0][lane]);
    {
        const float a0 = expf(m_run - m_all);                       // wave 0 always owns >= 1 tile
        l_run *= a0;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][r] *= a0;
    }
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
        const float mw = s_ml[w][0][lane];
        const float aw = mw == -INFINITY ? 0.0f : expf(mw - m_all);   // slice without keys
        l_run = fmaf(s_ml[w][1][lane], aw, l_run);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][r] = fmaf(s_o[w][a * 16 + r][lane], aw, o[a][r]);
    }
    // o[a][r] = O^T[d = 32 a + (r&3) + 8*(r>>2) + 4*hh][query = li]
    const int qi = q0 + li;
    if (qi < p.nq) {
        const float inv = 1.0f / l_run;
        float *dst = p.O + (size_t)b * p.sO + (size_t)qi * p.ldo + h * HD;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = make_float4(o[a][4 * g] * inv, o[a][4 * g + 1] * inv, o[a][4 * g + 2] * inv, o[a][4 * g + 3] * inv);
                *reinterpret_cast<float4 *>(dst + 32 * a + 8 * g + 4 * hh) = t;
            }
    }
}

int launch_attn32(const AttnBatch &ab, hipStream_t st)
{
    int nqmax = 0, nkmax = 0;
    for (int g = 0; g < ab.n; ++g) {
        const AttnProb &p = ab.p[g];
        PTX_REQUIRE(p.Q && p.K && p.V && p.O, "attention: null operand in group %d", g);
        PTX_REQUIRE(p.nk >= 1, "attention: no keys in group %d", g);
        PTX_REQUIRE(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldo % 4 == 0 && p.sQ % 4 == 0 &&
                    p.sK % 4 == 0 && p.sO % 4 == 0, "attention: strides must be multiples of 4 floats");
        nqmax = p.nq > nqmax ? p.nq : nqmax;
        nkmax = p.nk > nkmax ? p.nk : nkmax;
    }
    if (nqmax == 0) return PTX_OK;
    PTX_REQUIRE(ab.hd == 32 || ab.hd == 64, "attention: head_dim=%d (supported: 32, 64)", ab.hd);
    const dim3 grid(cdiv(nqmax, 32), ab.B * ab.heads, ab.n);
    // waves per work-group: enough that a wave walks at most two key tiles, while the launch is small enough to be
    // waiting on that walk (with thousands of work-groups the chip is full anyway and more waves only add merge work)
    const int ntile = cdiv(nkmax, 32);
    const long wgs = (long)grid.x * grid.y * grid.z;
    int nw = 4;
    if (wgs <= 512 && ntile > 8) nw = 8;        // (16 waves: 128 VGPRs with spills and 68 KB of merge space -- not built)
    if (ab.hd == 32) {
        if (nw == 8) hipLaunchKernelGGL((k_attn32<32, 8>), grid, dim3(512), 0, st, ab);
        else hipLaunchKernelGGL((k_attn32<32, 4>), grid, dim3(256), 0, st, ab);
    } else {
        if (nw == 8) hipLaunchKernelGGL((k_attn32<64, 8>), grid, dim3(512), 0, st, ab);
        else hipLaunchKernelGGL((k_attn32<64, 4>), grid, dim3(256), 0, st, ab);
    }
    PTX_LAUNCHED("k_attn32");
    return PTX_OK;
}

}  // namespace ptx
