// Voxel quantisation of the preshaped point clouds -- the step right after the path in the reference's detector
// (detectors/sparse_featfusion_grounder_preshape.py:388-397):
//     coordinates, features = ME.utils.batch_sparse_collate([(p[:, :3] / voxel_size, p) for p in points])
//     x = ME.SparseTensor(coordinates=coordinates, features=features)
// i.e. per point the integer voxel floor(p / voxel_size) with the scene index in front (batch_sparse_collate), and one
// row per occupied voxel (SparseTensor's default quantisation keeps ONE point of every voxel).  MinkowskiEngine is not
// vendored in the reference: which duplicate survives and the row order are implementation details of its coordinate
// map ("random subsample") -- pinned here to the FIRST point of every voxel in (scene, point) order, rows in that
// order.  "Parity unpinned" against ME itself; bit-exact against the CPU restatement the tests hold.
//
// HBM-bound integer work: one hash insert per point (64-bit CAS on a table of 2x the points, linear probing), a
// count pass and an ordered emit pass over 2048-point tiles (same compaction scheme as k_affine).
#include "common.h"

namespace ptx {

constexpr unsigned long long kVoxEmpty = ~0ull;
constexpr int kVoxBias = 1 << 18;          // voxel indices in [-2^18, 2^18): +-2.6 km at 1 cm

struct VoxArgs {
    const float *points; const int32_t *counts; int B, Ncap; float voxel_size;
    unsigned long long *keys; int32_t *minidx; int32_t *slot_of; int32_t *row_of_slot; int32_t *tile_counts;
    unsigned int mask;
    int32_t *coords; float *feats; int32_t *inverse; int32_t *nvox; int32_t *overflow;
};

__device__ __forceinline__ bool vox_key(const VoxArgs &a, int b, int i, int (&v)[3], unsigned long long &key)
{
    const float *p = a.points + ((size_t)b * a.Ncap + i) * 3;
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        v[d] = (int)floorf(__fdiv_rn(p[d], a.voxel_size));              // torch: floor(p / voxel_size), fp32
        ok = ok && v[d] >= -kVoxBias && v[d] < kVoxBias;
    }
    key = ((unsigned long long)b << 57) | ((unsigned long long)(v[0] + kVoxBias) << 38) |
          ((unsigned long long)(v[1] + kVoxBias) << 19) | (unsigned long long)(v[2] + kVoxBias);
    return ok;
}

__device__ __forceinline__ unsigned int vox_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned int)k;
}

// pass 1: every valid point claims (or finds) the slot of its voxel and lowers the slot's owner to its own index
__global__ __launch_bounds__(256) void k_vox_insert(VoxArgs a)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.counts[b]) return;
    int v[3]; unsigned long long key;
    if (!vox_key(a, b, i, v, key)) { atomicAdd(a.overflow, 1); }
    const int gi = b * a.Ncap + i;
    unsigned int slot = vox_hash(key) & a.mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&a.keys[slot], kVoxEmpty, key);
        if (prev == kVoxEmpty || prev == key) break;
        slot = (slot + 1) & a.mask;
    }
    atomicMin(&a.minidx[slot], gi);
    a.slot_of[gi] = (int)slot;
}

// pass 2: representatives (first point of a voxel) per 2048-point tile
__global__ __launch_bounds__(256) void k_vox_count(VoxArgs a)
{
    const int b = blockIdx.y, tile = blockIdx.x, nb = a.counts[b];
    int c = 0;
#pragma unroll
    for (int r = 0; r < kTilePts / 256; ++r) {
        const int i = tile * kTilePts + r * 256 + threadIdx.x;
        if (i < nb) { const int gi = b * a.Ncap + i; c += a.minidx[a.slot_of[gi]] == gi; }
    }
    c = wave_sum(c);
    __shared__ int red[4];
    if (lane_id() == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_counts[b * gridDim.x + tile] = red[0] + red[1] + red[2] + red[3];
}

// pass 3: ordered emit: row = number of representatives before this point in (scene, point) order
__global__ __launch_bounds__(256) void k_vox_emit(VoxArgs a)
{
    const int b = blockIdx.y, tile = blockIdx.x, ntiles = gridDim.x, nb = a.counts[b];
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    constexpr int R = kTilePts / 256;
    __shared__ int s_cnt[R][4];
    __shared__ int s_base;
    int acc = 0;
    const int before_tiles = b * ntiles + tile;
    for (int t = tid; t < before_tiles; t += 256) acc += a.tile_counts[t];
    acc = wave_sum(acc);
    if (lane == 0) s_cnt[0][wid] = acc;
    __syncthreads();
    if (tid == 0) {
        s_base = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
        if (b == (int)gridDim.y - 1 && tile == ntiles - 1) *a.nvox = s_base + a.tile_counts[before_tiles];
    }
    __syncthreads();
    const int base = s_base;
    __syncthreads();
    bool rep[R]; unsigned long long bal[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tile * kTilePts + r * 256 + tid;
        rep[r] = false;
        if (i < nb) { const int gi = b * a.Ncap + i; rep[r] = a.minidx[a.slot_of[gi]] == gi; }
        bal[r] = __ballot(rep[r]);
        if (lane == 0) s_cnt[r][wid] = __popcll(bal[r]);
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    int run = base;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int before = 0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) before += ww < wid ? s_cnt[r][ww] : 0;
        if (rep[r]) {
            const int i = tile * kTilePts + r * 256 + tid, gi = b * a.Ncap + i;
            const int row = run + before + __popcll(bal[r] & lt);
            int v[3]; unsigned long long key;
            vox_key(a, b, i, v, key);
            int32_t *c = a.coords + (size_t)row * 4;
            c[0] = b; c[1] = v[0]; c[2] = v[1]; c[3] = v[2];
            const float *p = a.points + (size_t)gi * 3;
            float *f = a.feats + (size_t)row * 3;
            f[0] = p[0]; f[1] = p[1]; f[2] = p[2];
            a.row_of_slot[a.slot_of[gi]] = row;
        }
        run += s_cnt[r][0] + s_cnt[r][1] + s_cnt[r][2] + s_cnt[r][3];
    }
}

// pass 4 (optional): voxel row of every input point
__global__ __launch_bounds__(256) void k_vox_inverse(VoxArgs a)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.Ncap) return;
    const int gi = b * a.Ncap + i;
    a.inverse[gi] = i < a.counts[b] ? a.row_of_slot[a.slot_of[gi]] : -1;
}

// last launch: the row count and the overflow count with system scope (nvox_overflow may be device-mapped pinned host
// memory preset to -1: the host spins on it with ptx_wait_counts instead of draining the stream and copying)
__global__ void k_vox_publish(const int32_t *acc, int32_t *nvox_overflow)
{
    __hip_atomic_store(nvox_overflow + 1, acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(nvox_overflow, acc[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct VoxLayout { size_t keys, minidx, slot_of, row_of_slot, tile_counts, overflow, total; unsigned int slots; };
static VoxLayout vox_layout(int B, int Ncap)
{
    VoxLayout L{};
    const size_t total = (size_t)B * Ncap;
    unsigned int slots = 1024;
    while ((size_t)slots < 2 * total) slots <<= 1;
    L.slots = slots;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    L.keys = take((size_t)slots * 8); L.minidx = take((size_t)slots * 4);
    L.slot_of = take(total * 4); L.row_of_slot = take((size_t)slots * 4);
    L.tile_counts = take((size_t)B * cdiv(Ncap, kTilePts) * 4); L.overflow = take(8);
    L.total = o;
    return L;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

size_t ptx_voxel_workspace_bytes(int B, int Ncap)
{
    if (B < 1 || Ncap < 1 || B > 64 || (long)B * Ncap > (1l << 30)) return 0;
    return vox_layout(B, Ncap).total;
}

int ptx_voxelize(const float *points, const int32_t *counts, int B, int Ncap, float voxel_size, int32_t *coords,
                 float *feats, int32_t *inverse, int32_t *nvox_overflow, void *workspace, size_t ws_bytes, void *stream)
{
    PTX_REQUIRE(points && counts && coords && feats && nvox_overflow && workspace, "ptx_voxelize: null argument");
    PTX_REQUIRE(B >= 1 && B <= 64 && Ncap >= 1 && (long)B * Ncap <= (1l << 30) && voxel_size > 0.0f,
                "ptx_voxelize: B=%d Ncap=%d voxel_size=%g", B, Ncap, voxel_size);
    const VoxLayout L = vox_layout(B, Ncap);
    if (ws_bytes < L.total) { set_error("ptx_voxelize: workspace too small: %zu < %zu bytes", ws_bytes, L.total); return PTX_ENOSPACE; }
    PTX_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "ptx_voxelize: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    VoxArgs a{points, counts, B, Ncap, voxel_size,
              reinterpret_cast<unsigned long long *>(ws + L.keys), reinterpret_cast<int32_t *>(ws + L.minidx),
              reinterpret_cast<int32_t *>(ws + L.slot_of), reinterpret_cast<int32_t *>(ws + L.row_of_slot),
              reinterpret_cast<int32_t *>(ws + L.tile_counts), L.slots - 1, coords, feats, inverse,
              reinterpret_cast<int32_t *>(ws + L.overflow), reinterpret_cast<int32_t *>(ws + L.overflow) + 1};
    PTX_HIP(hipMemsetAsync(ws + L.keys, 0xFF, (size_t)L.slots * 8, st));
    PTX_HIP(hipMemsetAsync(ws + L.minidx, 0x7F, (size_t)L.slots * 4, st));
    PTX_HIP(hipMemsetAsync(ws + L.overflow, 0, 8, st));
    const dim3 per_point(cdiv(Ncap, 256), B), per_tile(cdiv(Ncap, kTilePts), B);
    hipLaunchKernelGGL(k_vox_insert, per_point, dim3(256), 0, st, a);
    PTX_LAUNCHED("k_vox_insert");
    hipLaunchKernelGGL(k_vox_count, per_tile, dim3(256), 0, st, a);
    PTX_LAUNCHED("k_vox_count");
    hipLaunchKernelGGL(k_vox_emit, per_tile, dim3(256), 0, st, a);
    PTX_LAUNCHED("k_vox_emit");
    hipLaunchKernelGGL(k_vox_publish, dim3(1), dim3(1), 0, st, reinterpret_cast<const int32_t *>(ws + L.overflow), nvox_overflow);
    PTX_LAUNCHED("k_vox_publish");
    if (inverse) {
        hipLaunchKernelGGL(k_vox_inverse, per_point, dim3(256), 0, st, a);
        PTX_LAUNCHED("k_vox_inverse");
    }
    return PTX_OK;
}

}  // extern "C"
