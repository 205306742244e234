// Multi-view depth ingest on the device: what the reference's data pipeline does on the host between
// the decoded depth maps and the (N,3) point cloud the preshape path consumes,
//   ConvertRGBDToPoints      datasets/transforms/points.py:20-98   un-project every pixel with depth != 0 (points_img2cam,
//                                                                  structures/bbox_3d/utils.py:336-368), row-major order
//   PointSample (per view)   points.py:290-420                     np.random.choice over the view's points   [host RNG]
//   AggregateMultiViewPoints multiview.py:195-253                  per view solve(global2ego, [p;1]), concatenate
//   PointSample (scene)      points.py:290-420                     np.random.choice over the concatenation   [host RNG]
//   (GlobalRotScaleTrans     augmentation.py:253-: rotate, scale, translate -- optional, parameters from the host RNG)
// The host keeps the RNG (np.random) and composes the two choices into ONE index per output point: `sel[j]` = position
// of output point j in the concatenation, over the views in order, of each view's depth != 0 pixels in row-major order
// (the reference's `grid3d[nonzero_indices]`).  The device resolves that rank to a pixel with a two-level rank / select
// index over 64-pixel groups and computes only the N points that survive -- the reference un-projects all V*H*W pixels.
//
//   k_ingest_index   one pass over the depth maps (HBM-bound, 2 or 4 B per pixel): per group of 64 pixels the bit mask of
//                    depth != 0 and its exclusive count prefix inside a chunk of 256 groups; per chunk the total
//   k_ingest_scan    exclusive prefix over the chunk totals (one work-group), per-view counts for the host (the
//                    reference's len(points) per view, which decides `replace` in np.random.choice)
//   k_ingest_gather  one thread per output point: binary search chunk -> group, select the r-th set bit of the group's
//                    mask, read ONE depth value, un-project, transform, (augment,) store; the scene's bounding box is
//                    reduced on the way and published in the encoding k_cluster reads (the forward then skips k_minmax)
#include <cstdlib>

#include "common.h"

namespace ptx {

constexpr int kGroupsPerChunk = 256;            // 16384 pixels per work-group of k_ingest_index

struct IngestLayout { size_t masks, prefix, chunk_tot, chunk_off, total; int gpv, cpv; };

static IngestLayout ingest_layout(int V, int H, int W)
{
    IngestLayout L{};
    const long hw = (long)H * W;
    L.gpv = (int)((hw + 63) / 64);
    L.cpv = (L.gpv + kGroupsPerChunk - 1) / kGroupsPerChunk;
    const size_t G = (size_t)V * L.cpv * kGroupsPerChunk, NC = (size_t)V * L.cpv;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    L.masks = take(G * 8); L.prefix = take(G * 4); L.chunk_tot = take(NC * 4); L.chunk_off = take((NC + 1) * 8);
    L.total = o;
    return L;
}

template <typename T> __device__ __forceinline__ bool depth_nonzero(T d);
template <> __device__ __forceinline__ bool depth_nonzero<float>(float d) { return d != 0.0f; }       // nonzero(): NaN counts
template <> __device__ __forceinline__ bool depth_nonzero<uint16_t>(uint16_t d) { return d != 0; }

// grid (cpv, V), 256 threads: wave w of chunk c owns groups 64 w .. 64 w + 63 of the chunk = 4096 consecutive pixels.
// VEC (r04; the view's rows are 16-byte aligned): a lane reads EIGHT consecutive pixels per step with 16-byte loads (one for
// uint16, two for float), all eight steps of the wave requested up front, turns them into one byte of "depth != 0" bits and
// drops it into a 512-byte wave-private LDS image of the wave's 64 group masks; lane g then reads group g's 64-bit mask.
// (r03: one 2-byte load per lane and step + a ballot -- 64 dependent trips per wave, 1.0 TB/s.)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_ingest_index(const T *__restrict__ depth, long hw, int gpv, int cpv,
                                                      unsigned long long *__restrict__ masks, uint32_t *__restrict__ prefix,
                                                      uint32_t *__restrict__ chunk_tot)
{
    const int v = blockIdx.y, c = blockIdx.x, lane = lane_id(), wv = threadIdx.x >> 6;
    const T *__restrict__ d = depth + (size_t)v * hw;
    const int g0 = c * kGroupsPerChunk + wv * 64;           // first group (of the view) of this wave
    unsigned long long mine = 0ull;
    __shared__ __attribute__((aligned(16))) unsigned char s_bits[4][512];
    if (VEC) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        constexpr int PER = 16 / (int)sizeof(T);            // pixels per 16-byte load: 8 (uint16) or 4 (float)
        constexpr int NL = 8 / PER;                         // loads per lane and step
        const long p0 = (long)g0 << 6;                      // first pixel of the wave
        u32x4 q[8][NL];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long p = p0 + (long)(k * 64 + lane) * 8;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const long pp = p + u * PER;                // (hw is a multiple of PER here: a 16-byte load is inside or outside)
                q[k][u] = pp < hw ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(d + pp)) : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned int bits = 0u;
#pragma unroll
            for (int u = 0; u < NL; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int w = q[k][u][e];
                    if (sizeof(T) == 2) {
                        bits |= ((w & 0xffffu) != 0u ? 1u : 0u) << (2 * e) | ((w >> 16) != 0u ? 1u : 0u) << (2 * e + 1);
                    } else {
                        bits |= (__uint_as_float(w) != 0.0f ? 1u : 0u) << (4 * u + e);      // nonzero(): NaN counts, -0.0 does not
                    }
                }
            s_bits[wv][k * 64 + lane] = (unsigned char)bits;
        }
        // wave-private image: the wave's LDS operations complete in order, no barrier needed
        mine = *reinterpret_cast<const unsigned long long *>(&s_bits[wv][8 * lane]);
    } else {
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {
            const long p = ((long)(g0 + i) << 6) + lane;
            const bool nz = p < hw && depth_nonzero<T>(d[p]);
            const unsigned long long m = __ballot(nz);
            if (lane == i) mine = m;
        }
    }
    const int cnt = __popcll(mine);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    __shared__ int s_tot[4];
    if (lane == 63) s_tot[wv] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) base += w < wv ? s_tot[w] : 0;
    const size_t g = ((size_t)v * cpv + c) * kGroupsPerChunk + wv * 64 + lane;
    masks[g] = mine;
    prefix[g] = (uint32_t)(base + incl - cnt);
    if (threadIdx.x == 0) chunk_tot[(size_t)v * cpv + c] = (uint32_t)(s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3]);
}

// one work-group: chunk_off[i] = number of depth != 0 pixels before chunk i (view-major), chunk_off[NC] = total;
// view_counts[v] (device or pinned host memory) = pixels of view v
// (r04: run by the last work-group of k_ingest_index to arrive instead of as a launch of its own -- with one arrival word for
//  ~1000 work-groups their atomics serialise, 11 -> 19 us; with per-view words + one for the views 15 us: every work-group then ends
//  with a write-through store, a wait and an atomic, which costs the index launch more than the 4 us launch it saves.  Not kept.)
__global__ __launch_bounds__(256) void k_ingest_scan(const uint32_t *__restrict__ chunk_tot, int V, int cpv,
                                                     unsigned long long *__restrict__ chunk_off, int32_t *view_counts)
{
    __shared__ unsigned long long s_w[4];
    __shared__ unsigned long long s_carry;
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const int NC = V * cpv;
    if (tid == 0) s_carry = 0ull;
    __syncthreads();
    for (int i0 = 0; i0 < NC; i0 += 256) {
        const int i = i0 + tid;
        const unsigned long long x = i < NC ? chunk_tot[i] : 0ull;
        unsigned long long incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long n = __shfl_up(incl, o, 64);
            if (lane >= o) incl += n;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned long long base = s_carry;
#pragma unroll
        for (int w = 0; w < 4; ++w) base += w < wv ? s_w[w] : 0ull;
        if (i < NC) chunk_off[i] = base + incl - x;
        __syncthreads();
        if (tid == 255) s_carry = base + incl;
        __syncthreads();
    }
    if (tid == 0) chunk_off[NC] = s_carry;
    __syncthreads();
    // per-view totals: differences of the chunk offsets at the view boundaries (written above by this work-group)
    __threadfence_block();
    for (int v = tid; v < V; v += 256) {
        const unsigned long long a = chunk_off[(size_t)v * cpv], b = chunk_off[(size_t)(v + 1) * cpv];
        __hip_atomic_store(view_counts + v, (int32_t)(b - a), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct IngestGatherArgs {
    const void *depth; float depth_scale; int V, H, W; long hw; int gpv, cpv;
    const float *inv_k;              // (V,4,4) inverse of the padded intrinsic (points_img2cam: torch.inverse(pad_cam2img))
    const float *lu; const int32_t *piv;      // (V,4,4) LU factors (unit lower | upper) of global2ego, (V,4) row order
    const long long *sel; int N;
    const float *aug;                // null, or rot_mat_T (9) | scale | trans (3)
    const unsigned long long *masks; const uint32_t *prefix; const unsigned long long *chunk_off;
    float *points; uint32_t *bbox_enc; int32_t *status;
};

template <typename T>
__global__ __launch_bounds__(256) void k_ingest_gather(IngestGatherArgs a)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int NC = a.V * a.cpv;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (j < a.N) {
        const unsigned long long flat = (unsigned long long)a.sel[j];
        const unsigned long long total = a.chunk_off[NC];
        if (flat >= total) {                                    // index beyond the scene's points: reported, point = 0
            atomicOr(reinterpret_cast<unsigned int *>(a.status), 1u);
            a.points[(size_t)j * 3] = 0.0f; a.points[(size_t)j * 3 + 1] = 0.0f; a.points[(size_t)j * 3 + 2] = 0.0f;
        } else {
            // last chunk with chunk_off <= flat, then last group of it with prefix <= r (empty ones share a value with
            // their successor and are never the last)
            int lo_c = 0, hi_c = NC - 1;
            while (lo_c < hi_c) {
                const int mid = (lo_c + hi_c + 1) >> 1;
                if (a.chunk_off[mid] <= flat) lo_c = mid; else hi_c = mid - 1;
            }
            const uint32_t r = (uint32_t)(flat - a.chunk_off[lo_c]);
            const uint32_t *pf = a.prefix + (size_t)lo_c * kGroupsPerChunk;
            int lo_g = 0, hi_g = kGroupsPerChunk - 1;
            while (lo_g < hi_g) {
                const int mid = (lo_g + hi_g + 1) >> 1;
                if (pf[mid] <= r) lo_g = mid; else hi_g = mid - 1;
            }
            unsigned long long m = a.masks[(size_t)lo_c * kGroupsPerChunk + lo_g];
            int q = (int)(r - pf[lo_g]), bit = 0;
#pragma unroll
            for (int w = 32; w >= 1; w >>= 1) {                 // position of the q-th set bit
                const int c = __popcll(m & ((1ull << w) - 1ull));
                if (q >= c) { q -= c; bit += w; m >>= w; }
            }
            const int v = lo_c / a.cpv;
            const long pix = ((long)((lo_c - v * a.cpv) * kGroupsPerChunk + lo_g) << 6) + bit;
            const float dval = (float)static_cast<const T *>(a.depth)[(size_t)v * a.hw + pix];
            const float d = sizeof(T) == 2 ? dval / a.depth_scale : dval;       // LoadDepthFromFile: astype(float32) / depth_shift
            const float us = (float)(pix % a.W), vs = (float)(pix / a.W);
            // points_img2cam: homo = [u d, v d, d, 1];  p = homo @ inverse(pad_cam2img)^T  (first three columns)
            const float h0 = us * d, h1 = vs * d;
            const float *ik = a.inv_k + (size_t)v * 16;
            float p[4];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                p[k] = fmaf(1.0f, ik[k * 4 + 3], fmaf(d, ik[k * 4 + 2], fmaf(h1, ik[k * 4 + 1], h0 * ik[k * 4])));
            p[3] = 1.0f;
            // AggregateMultiViewPoints: x = solve(global2ego, [p;1]) through the LU factors (P A = L U)
            const float *lu = a.lu + (size_t)v * 16;
            const int32_t *pv = a.piv + (size_t)v * 4;
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int src = pv[i];
                y[i] = src == 0 ? p[0] : src == 1 ? p[1] : src == 2 ? p[2] : p[3];
            }
#pragma unroll
            for (int i = 1; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < i; ++k) y[i] = __fsub_rn(y[i], __fmul_rn(lu[i * 4 + k], y[k]));
#pragma unroll
            for (int i = 3; i >= 0; --i) {
#pragma unroll
                for (int k = 3; k > i; --k) y[i] = __fsub_rn(y[i], __fmul_rn(lu[i * 4 + k], y[k]));
                y[i] = __fdiv_rn(y[i], lu[i * 4 + i]);
            }
            float x = y[0], yy = y[1], z = y[2];
            if (a.aug != nullptr) {
                // GlobalRotScaleTrans on the points: p @ rot_mat_T, then * scale, then + trans (three separate fp32 steps)
                const float *R = a.aug;
                const float rx = (x * R[0] + yy * R[3]) + z * R[6], ry = (x * R[1] + yy * R[4]) + z * R[7],
                            rz = (x * R[2] + yy * R[5]) + z * R[8];
                x = rx * R[9] + R[10]; yy = ry * R[9] + R[11]; z = rz * R[9] + R[12];
            }
            a.points[(size_t)j * 3] = x; a.points[(size_t)j * 3 + 1] = yy; a.points[(size_t)j * 3 + 2] = z;
            lo[0] = hi[0] = x; lo[1] = hi[1] = yy; lo[2] = hi[2] = z;
        }
    }
    if (a.bbox_enc == nullptr) return;
    // bounding box of the scene in k_minmax's encoding: [0..2] = ~ord(min), [3..5] = ord(max) (atomicMax on zeroed words)
    __shared__ float red[4][6];
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) { lo[dd] = wave_min(lo[dd]); hi[dd] = wave_max(hi[dd]); }
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) { red[w][dd] = lo[dd]; red[w][3 + dd] = hi[dd]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int dd = threadIdx.x;
        float val = red[0][dd];
        for (int i = 1; i < 4; ++i) val = dd < 3 ? fminf(val, red[i][dd]) : fmaxf(val, red[i][dd]);
        if (dd < 3) { if (val != INFINITY) atomicMax(&a.bbox_enc[dd], ~f2ord(val)); }
        else        { if (val != -INFINITY) atomicMax(&a.bbox_enc[dd], f2ord(val)); }
    }
}

}  // namespace ptx

using namespace ptx;

extern "C" {

size_t ptx_ingest_workspace_bytes(int V, int H, int W)
{
    if (V < 1 || H < 1 || W < 1 || (long)V * H * W > (1l << 40)) return 0;
    return ingest_layout(V, H, W).total;
}

int ptx_ingest_index(const void *depth, int depth_dtype, int V, int H, int W, void *workspace, size_t ws_bytes,
                     int32_t *view_counts, void *stream)
{
    PTX_REQUIRE(depth && workspace && view_counts, "ptx_ingest_index: null argument");
    PTX_REQUIRE(depth_dtype == 0 || depth_dtype == 1, "ptx_ingest_index: depth_dtype=%d (0 float32, 1 uint16)", depth_dtype);
    PTX_REQUIRE(V >= 1 && H >= 1 && W >= 1, "ptx_ingest_index: V=%d H=%d W=%d", V, H, W);
    const IngestLayout L = ingest_layout(V, H, W);
    if (ws_bytes < L.total) { set_error("ptx_ingest_index: workspace too small: %zu < %zu bytes", ws_bytes, L.total); return PTX_ENOSPACE; }
    PTX_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "ptx_ingest_index: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    auto *masks = reinterpret_cast<unsigned long long *>(ws + L.masks);
    auto *prefix = reinterpret_cast<uint32_t *>(ws + L.prefix);
    auto *ctot = reinterpret_cast<uint32_t *>(ws + L.chunk_tot);
    auto *coff = reinterpret_cast<unsigned long long *>(ws + L.chunk_off);
    const long hw = (long)H * W;
    const dim3 grid(L.cpv, V);
    // 16-byte loads where every view starts on a 16-byte boundary (480 x 640 maps do), the element form otherwise
    const size_t esz = depth_dtype == 0 ? 4 : 2;
    const bool vec = (reinterpret_cast<uintptr_t>(depth) & 15) == 0 && ((size_t)hw * esz) % 16 == 0;
    if (depth_dtype == 0) {
        const float *dp = static_cast<const float *>(depth);
        if (vec) hipLaunchKernelGGL((k_ingest_index<float, true>), grid, dim3(256), 0, st, dp, hw, L.gpv, L.cpv, masks, prefix, ctot);
        else     hipLaunchKernelGGL((k_ingest_index<float, false>), grid, dim3(256), 0, st, dp, hw, L.gpv, L.cpv, masks, prefix, ctot);
    } else {
        const uint16_t *dp = static_cast<const uint16_t *>(depth);
        if (vec) hipLaunchKernelGGL((k_ingest_index<uint16_t, true>), grid, dim3(256), 0, st, dp, hw, L.gpv, L.cpv, masks, prefix, ctot);
        else     hipLaunchKernelGGL((k_ingest_index<uint16_t, false>), grid, dim3(256), 0, st, dp, hw, L.gpv, L.cpv, masks, prefix, ctot);
    }
    PTX_LAUNCHED("k_ingest_index");
    hipLaunchKernelGGL(k_ingest_scan, dim3(1), dim3(256), 0, st, ctot, V, L.cpv, coff, view_counts);
    PTX_LAUNCHED("k_ingest_scan");
    return PTX_OK;
}

int ptx_ingest_gather(const void *depth, int depth_dtype, float depth_shift, int V, int H, int W, const float *inv_intrinsic,
                      const float *lu, const int32_t *piv, const int64_t *sel, int N, const float *aug, float *points,
                      uint32_t *bbox_enc, int32_t *status, const void *workspace, size_t ws_bytes, void *stream)
{
    PTX_REQUIRE(depth && inv_intrinsic && lu && piv && sel && points && status && workspace, "ptx_ingest_gather: null argument");
    PTX_REQUIRE(depth_dtype == 0 || depth_dtype == 1, "ptx_ingest_gather: depth_dtype=%d (0 float32, 1 uint16)", depth_dtype);
    PTX_REQUIRE(V >= 1 && H >= 1 && W >= 1 && N >= 1, "ptx_ingest_gather: V=%d H=%d W=%d N=%d", V, H, W, N);
    PTX_REQUIRE(depth_dtype == 0 || depth_shift > 0.0f, "ptx_ingest_gather: depth_shift=%g", (double)depth_shift);
    const IngestLayout L = ingest_layout(V, H, W);
    if (ws_bytes < L.total) { set_error("ptx_ingest_gather: workspace too small: %zu < %zu bytes", ws_bytes, L.total); return PTX_ENOSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *ws = static_cast<const char *>(workspace);
    IngestGatherArgs a{depth, depth_shift, V, H, W, (long)H * W, L.gpv, L.cpv, inv_intrinsic, lu, piv,
                       reinterpret_cast<const long long *>(sel), N, aug,
                       reinterpret_cast<const unsigned long long *>(ws + L.masks),
                       reinterpret_cast<const uint32_t *>(ws + L.prefix),
                       reinterpret_cast<const unsigned long long *>(ws + L.chunk_off), points, bbox_enc, status};
    PTX_HIP(hipMemsetAsync(status, 0, 4, st));
    if (bbox_enc) PTX_HIP(hipMemsetAsync(bbox_enc, 0, 24, st));
    if (depth_dtype == 0) hipLaunchKernelGGL(k_ingest_gather<float>, dim3(cdiv(N, 256)), dim3(256), 0, st, a);
    else                  hipLaunchKernelGGL(k_ingest_gather<uint16_t>, dim3(cdiv(N, 256)), dim3(256), 0, st, a);
    PTX_LAUNCHED("k_ingest_gather");
    return PTX_OK;
}

}  // extern "C"
