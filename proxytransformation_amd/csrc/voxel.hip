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
// HBM-bound integer work.  r04 (r03: three memsets + insert + count + emit + publish = seven launches, 134 us for 399k
// points): ONE memset (the hash table stores key + 1 and the complemented owner index, so that zero means empty for both; the
// tile words and the overflow counter lie in the same region), then
//   k_vox_insert  one hash insert per point (64-bit CAS on a table of >= 2x the points, linear probing) + atomicMax of the
//                 complemented point index on the slot: the FIRST point of the voxel in (scene, point) order owns it
//   k_vox_emit    per 2048-point tile: representatives counted, the tile's count published (agent scope) at once, the
//                 counts of ALL tiles in front summed as soon as they appear (every work-group publishes before it waits
//                 and only waits for lower-numbered tiles, which were dispatched earlier: no order assumption beyond
//                 that), rows emitted in order; the last tile publishes the row count / overflow count to the host
#include "common.h"

namespace ptx {

constexpr int kVoxBias = 1 << 18;          // voxel indices in [-2^18, 2^18): +-2.6 km at 1 cm

struct VoxArgs {
    const float *points; const int32_t *counts; int B, Ncap; float voxel_size;
    unsigned long long *keys; uint32_t *owner; int32_t *first; int32_t *slot_of; int32_t *row_of_slot; unsigned long long *tile_word;
    unsigned int mask;
    int32_t *coords; float *feats; int32_t *inverse; int32_t *overflow; int32_t *nvox_overflow;
    int32_t *scene_end;                    // optional (ptx_voxelize_ex): rows written up to and including scene b
    // coarsening mode (ptx_voxel_coarsen): the "points" are the integer voxel rows of a finer level -- coords_in (rows,4) int32
    // (scene, x, y, z), scene b's rows [in_end[b-1], in_end[b]) -- and the voxel of a row is its coordinate >> shift (floor division
    // by the power-of-two stride); the emitted row carries floor(c / s) * s and the "feature" that coordinate times voxel_size
    const int32_t *coords_in; int shift; int32_t in_end[64];
};

__device__ __forceinline__ int vox_count(const VoxArgs &a, int b)
{
    return a.coords_in == nullptr ? a.counts[b] : a.in_end[b] - (b > 0 ? a.in_end[b - 1] : 0);
}

__device__ __forceinline__ bool vox_key(const VoxArgs &a, int b, int i, int (&v)[3], unsigned long long &key)
{
    bool ok = true;
    if (a.coords_in != nullptr) {
        const int32_t *c = a.coords_in + ((size_t)(b > 0 ? a.in_end[b - 1] : 0) + i) * 4;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d] = c[1 + d] >> a.shift;                                   // arithmetic shift = floor division by the stride
            ok = ok && v[d] >= -kVoxBias && v[d] < kVoxBias;
        }
    } else {
        const float *p = a.points + ((size_t)b * a.Ncap + i) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d] = (int)floorf(__fdiv_rn(p[d], a.voxel_size));            // torch: floor(p / voxel_size), fp32
            ok = ok && v[d] >= -kVoxBias && v[d] < kVoxBias;
        }
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

// pass 1: every valid point claims (or finds) the slot of its voxel.  ONE device-scope atomic per point in the common case
// (r04: CAS + atomicMax per point were 46 us for 399k points -- the table is shared by the whole chip, its atomics execute at
// the memory side): a slot is probed with a plain load first (slots only ever go empty -> key, so a stale "empty" is put
// right by the CAS and an occupied slot of another voxel is skipped without an atomic); the point whose CAS claims the slot
// stores its index with a plain store into `first`, and only points that FIND their voxel already there (duplicates: rare at
// 1 cm) raise the complemented minimum in `owner`.  The representative of a voxel is min(first, ~owner).
__global__ __launch_bounds__(256) void k_vox_insert(VoxArgs a)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= vox_count(a, b)) return;
    int v[3]; unsigned long long key;
    if (!vox_key(a, b, i, v, key)) { atomicAdd(a.overflow, 1); }
    const int gi = b * a.Ncap + i;
    const unsigned long long k1 = key + 1ull;               // 0 = empty (the table is cleared with one memset)
    unsigned int slot = vox_hash(key) & a.mask;
    for (;;) {
        unsigned long long cur = a.keys[slot];
        bool claimed = false;
        if (cur == 0ull) {
            cur = atomicCAS(&a.keys[slot], 0ull, k1);       // (a stale "empty" is put right here)
            claimed = cur == 0ull;
        }
        if (claimed) { a.first[slot] = gi; break; }         // the one plain store of this slot
        if (cur == k1) { atomicMax(&a.owner[slot], ~(uint32_t)gi); break; }     // my voxel was there already: a duplicate
        slot = (slot + 1) & a.mask;
    }
    a.slot_of[gi] = (int)slot;
}

// pass 2: representatives (first point of a voxel) counted and emitted in (scene, point) order, tile by tile
__global__ __launch_bounds__(256) void k_vox_emit(VoxArgs a)
{
    const int b = blockIdx.y, tile = blockIdx.x, ntiles = gridDim.x, nb = vox_count(a, b);
    const int tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    constexpr int R = kTilePts / 256;
    __shared__ int s_cnt[R][4];
    __shared__ int s_red[4];
    const int me = b * ntiles + tile;
    bool rep[R]; unsigned long long bal[R]; int slot[R];
    int mine = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tile * kTilePts + r * 256 + tid;
        slot[r] = i < nb ? a.slot_of[b * a.Ncap + i] : -1;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tile * kTilePts + r * 256 + tid;
        // first point of the voxel in (scene, point) order = min(the claimer's index, the smallest duplicate arrival)
        rep[r] = false;
        if (slot[r] >= 0) {
            const uint32_t dup = ~a.owner[slot[r]];         // owner 0 (no duplicate) -> 0xffffffff
            rep[r] = min((uint32_t)a.first[slot[r]], dup) == (uint32_t)(b * a.Ncap + i);
        }
        bal[r] = __ballot(rep[r]);
        if (lane == 0) { s_cnt[r][wid] = __popcll(bal[r]); }
        mine += rep[r];
    }
    mine = wave_sum(mine);
    if (lane == 0) s_red[wid] = mine;
    __syncthreads();
    const int total = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    // publish this tile's count at once (bit 63 = present), then add up everything in front as it appears
    if (tid == 0) __hip_atomic_store(a.tile_word + me, 0x8000000000000000ull | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                        // s_red is re-used below
    long long before = 0;
    for (int t = tid; t < me; t += 256) {
        unsigned long long w;
        unsigned spins = 0;
        while (((w = __hip_atomic_load(a.tile_word + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63) == 0ull) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) {
                // a lower tile that never publishes: give up rather than hang the device -- but not silently: the sticky
                // word next to the overflow counter turns the published row count into PTX_VOX_BROKEN
                __hip_atomic_store(a.overflow + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        before += (long long)(w & 0x7fffffffull);
    }
    int acc = (int)before;
    acc = wave_sum(acc);
    if (lane == 0) s_red[wid] = acc;
    __syncthreads();
    const int base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    if (tid == 0 && b == (int)gridDim.y - 1 && tile == ntiles - 1) {
        // the row count and the overflow count with system scope (nvox_overflow may be device-mapped pinned host memory preset
        // to -1: the host spins on it with ptx_wait_counts instead of draining the stream); rows first: the count releases them
        __hip_atomic_store(a.nvox_overflow + 1, __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the last tile of a scene knows where the scene's rows end (ptx_voxelize_ex; system scope like the total: the buffer may be
    // device-mapped pinned host memory that the host polls)
    if (a.scene_end != nullptr && tid == 0 && tile == ntiles - 1)
        __hip_atomic_store(a.scene_end + b, base + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int run = base;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int front = 0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) front += ww < wid ? s_cnt[r][ww] : 0;
        if (rep[r]) {
            const int i = tile * kTilePts + r * 256 + tid, gi = b * a.Ncap + i;
            const int row = run + front + __popcll(bal[r] & lt);
            int v[3]; unsigned long long key;
            vox_key(a, b, i, v, key);
            int32_t *c = a.coords + (size_t)row * 4;
            float *f = a.feats + (size_t)row * 3;
            if (a.coords_in != nullptr) {       // a coarser level: the coordinate in the finest level's units and its position
                const int sm = 1 << a.shift, cx = v[0] * sm, cy = v[1] * sm, cz = v[2] * sm;
                *reinterpret_cast<int4 *>(c) = make_int4(b, cx, cy, cz);
                f[0] = __fmul_rn((float)cx, a.voxel_size); f[1] = __fmul_rn((float)cy, a.voxel_size); f[2] = __fmul_rn((float)cz, a.voxel_size);
            } else {
                *reinterpret_cast<int4 *>(c) = make_int4(b, v[0], v[1], v[2]);
                const float *p = a.points + (size_t)gi * 3;
                f[0] = p[0]; f[1] = p[1]; f[2] = p[2];
            }
            a.row_of_slot[slot[r]] = row;
        }
        run += s_cnt[r][0] + s_cnt[r][1] + s_cnt[r][2] + s_cnt[r][3];
    }
    if (b == (int)gridDim.y - 1 && tile == ntiles - 1) {
        // every row of the call is written when the LAST tile's rows are?  No: other tiles may still be emitting.  The count is
        // only a size; the rows themselves are ordered on the stream like any other result (ptx_voxelize's contract).
        __syncthreads();
        // (this tile has waited for EVERY tile in front: whoever gave up on a tile did so no later than this one)
        if (tid == 0) {
            const int broken = __hip_atomic_load(a.overflow + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.nvox_overflow, broken ? 0x7fffffff : base + total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// pass 3 (optional): voxel row of every input point
__global__ __launch_bounds__(256) void k_vox_inverse(VoxArgs a)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.Ncap) return;
    const int gi = b * a.Ncap + i;
    a.inverse[gi] = i < vox_count(a, b) ? a.row_of_slot[a.slot_of[gi]] : -1;
}

struct VoxLayout { size_t zero_begin, keys, owner, tile_word, overflow, zero_bytes, first, slot_of, row_of_slot, total; unsigned int slots; };
static VoxLayout vox_layout(int B, int Ncap)
{
    VoxLayout L{};
    const size_t total = (size_t)B * Ncap;
    unsigned int slots = 1024;
    while ((size_t)slots < 2 * total) slots <<= 1;
    L.slots = slots;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes, 256); return r; };
    L.zero_begin = o;                                       // cleared by ONE memset per call
    L.keys = take((size_t)slots * 8); L.owner = take((size_t)slots * 4);
    L.tile_word = take((size_t)B * cdiv(Ncap, kTilePts) * 8); L.overflow = take(8);
    L.zero_bytes = o - L.zero_begin;
    L.first = take((size_t)slots * 4); L.slot_of = take(total * 4); L.row_of_slot = take((size_t)slots * 4);
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
    return ptx_voxelize_ex(points, counts, B, Ncap, voxel_size, coords, feats, inverse, nvox_overflow, nullptr, workspace, ws_bytes,
                           stream);
}

int ptx_voxelize_ex(const float *points, const int32_t *counts, int B, int Ncap, float voxel_size, int32_t *coords,
                    float *feats, int32_t *inverse, int32_t *nvox_overflow, int32_t *scene_end, void *workspace, size_t ws_bytes,
                    void *stream)
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
              reinterpret_cast<unsigned long long *>(ws + L.keys), reinterpret_cast<uint32_t *>(ws + L.owner),
              reinterpret_cast<int32_t *>(ws + L.first), reinterpret_cast<int32_t *>(ws + L.slot_of), reinterpret_cast<int32_t *>(ws + L.row_of_slot),
              reinterpret_cast<unsigned long long *>(ws + L.tile_word), L.slots - 1, coords, feats, inverse,
              reinterpret_cast<int32_t *>(ws + L.overflow), nvox_overflow, scene_end, nullptr, 0, {}};
    PTX_HIP(hipMemsetAsync(ws + L.zero_begin, 0, L.zero_bytes, st));
    const dim3 per_point(cdiv(Ncap, 256), B), per_tile(cdiv(Ncap, kTilePts), B);
    hipLaunchKernelGGL(k_vox_insert, per_point, dim3(256), 0, st, a);
    PTX_LAUNCHED("k_vox_insert");
    hipLaunchKernelGGL(k_vox_emit, per_tile, dim3(256), 0, st, a);
    PTX_LAUNCHED("k_vox_emit");
    if (inverse) {
        hipLaunchKernelGGL(k_vox_inverse, per_point, dim3(256), 0, st, a);
        PTX_LAUNCHED("k_vox_inverse");
    }
    return PTX_OK;
}

int ptx_voxel_coarsen(const int32_t *coords_in, const int32_t *in_scene_end, int B, int stride, float voxel_size,
                      int32_t *coords, float *points, int32_t *nvox_overflow, int32_t *scene_end, void *workspace, size_t ws_bytes,
                      void *stream)
{
    PTX_REQUIRE(coords_in && in_scene_end && coords && points && nvox_overflow && workspace, "ptx_voxel_coarsen: null argument");
    PTX_REQUIRE(B >= 1 && B <= 64 && stride >= 1 && (stride & (stride - 1)) == 0 && stride <= (1 << 16) && voxel_size > 0.0f,
                "ptx_voxel_coarsen: B=%d stride=%d voxel_size=%g (stride: a power of two)", B, stride, voxel_size);
    int ncap = 1, prev = 0;
    for (int b = 0; b < B; ++b) {
        PTX_REQUIRE(in_scene_end[b] >= prev, "ptx_voxel_coarsen: scene ends must not decrease");
        ncap = in_scene_end[b] - prev > ncap ? in_scene_end[b] - prev : ncap;
        prev = in_scene_end[b];
    }
    const VoxLayout L = vox_layout(B, ncap);
    if (ws_bytes < L.total) { set_error("ptx_voxel_coarsen: workspace too small: %zu < %zu bytes", ws_bytes, L.total); return PTX_ENOSPACE; }
    PTX_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "ptx_voxel_coarsen: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    VoxArgs a{nullptr, nullptr, B, ncap, voxel_size,
              reinterpret_cast<unsigned long long *>(ws + L.keys), reinterpret_cast<uint32_t *>(ws + L.owner),
              reinterpret_cast<int32_t *>(ws + L.first), reinterpret_cast<int32_t *>(ws + L.slot_of), reinterpret_cast<int32_t *>(ws + L.row_of_slot),
              reinterpret_cast<unsigned long long *>(ws + L.tile_word), L.slots - 1, coords, points, nullptr,
              reinterpret_cast<int32_t *>(ws + L.overflow), nvox_overflow, scene_end, coords_in, 0, {}};
    while ((1 << a.shift) < stride) ++a.shift;
    for (int b = 0; b < B; ++b) a.in_end[b] = in_scene_end[b];
    PTX_HIP(hipMemsetAsync(ws + L.zero_begin, 0, L.zero_bytes, st));
    const dim3 per_point(cdiv(ncap, 256), B), per_tile(cdiv(ncap, kTilePts), B);
    hipLaunchKernelGGL(k_vox_insert, per_point, dim3(256), 0, st, a);
    PTX_LAUNCHED("k_vox_insert");
    hipLaunchKernelGGL(k_vox_emit, per_tile, dim3(256), 0, st, a);
    PTX_LAUNCHED("k_vox_emit");
    return PTX_OK;
}

}  // extern "C"
