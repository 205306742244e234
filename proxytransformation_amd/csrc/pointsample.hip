// Image feature -> point sampling: batch_point_sample of the reference
// (models/layers/fusion_layers/point_fusion.py:208-313) as its detector calls it after the sparse backbone
// (detectors/sparse_featfusion_grounder_preshape.py:428-444: aligned=False -> nearest, zeros padding, align_corners=True,
// valid_flag=True): every point is projected into all V views, the nearest feature-map pixel of every view is gathered,
// the samples are summed and divided by the number of views in which the point is inside the (padded) image with
// positive depth.
//
// Layout: the feature maps arrive channels-first (V,C,H,W); gathering C channels of one pixel from that layout touches C
// cache lines.  k_feat_transpose makes one channels-last copy (V,H*W,C) (HBM-bound, V*C*H*W*4 B each way, caller-owned
// workspace); k_point_sample then runs one wave per point: lanes = views for the projection (ballot of the views that
// hit a pixel), lanes = channels for the gather (256 B contiguous per view and 64 channels).
#include "common.h"

namespace ptx {

__device__ __forceinline__ float ps_load(const void *base, size_t off, int dt)
{
    if (dt == 0) return static_cast<const float *>(base)[off];
    const unsigned short u = static_cast<const unsigned short *>(base)[off];
    if (dt == 1) return __uint_as_float((unsigned int)u << 16);
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}

// (V, C, HW) -> (V, HW, C), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_feat_transpose(const void *__restrict__ in, int dt, int C, int HW, float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int v = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r, p = p0 + tx;
        tile[ty + 8 * r][tx] = (c < C && p < HW) ? ps_load(in, ((size_t)v * C + c) * HW + p, dt) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = p0 + ty + 8 * r, c = c0 + tx;
        if (p < HW && c < C) out[((size_t)v * HW + p) * C + c] = tile[tx][ty + 8 * r];
    }
}

struct PsArgs {
    const float *points; int N;
    const float *featT; int V, C, H, W;
    const float *proj; const float *pre;
    float sx, sy, cx, cy; int flip; float ori_w, pad_h, pad_w;
    float *out; int32_t *valid_num;
    int bilinear;       // aligned=True: F.grid_sample(mode='bilinear'); 0: 'nearest' (what the detector asks for)
};

constexpr int kPsMaxQ = 8;      // C <= 512

__global__ __launch_bounds__(256) void k_point_sample(PsArgs a)
{
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (n >= a.N) return;
    const int lane = lane_id();
    float x = a.points[(size_t)n * 3], y = a.points[(size_t)n * 3 + 1], z = a.points[(size_t)n * 3 + 2];
    if (a.pre) {        // reverse 3D augmentation, composed by the host into one affine
        const float *A = a.pre;
        const float nx = fmaf(A[2], z, fmaf(A[1], y, A[0] * x)) + A[3];
        const float ny = fmaf(A[6], z, fmaf(A[5], y, A[4] * x)) + A[7];
        const float nz = fmaf(A[10], z, fmaf(A[9], y, A[8] * x)) + A[11];
        x = nx; y = ny; z = nz;
    }
    float acc[kPsMaxQ];
#pragma unroll
    for (int q = 0; q < kPsMaxQ; ++q) acc[q] = 0.0f;
    int nvalid = 0;
    const int HW = a.H * a.W;
    for (int v0 = 0; v0 < a.V; v0 += 64) {
        const int v = v0 + lane;
        bool inb = false, valid = false;
        int pix = 0;
        float wx1 = 0.0f, wy1 = 0.0f;          // bilinear: weights of the +1 neighbours; pix = (y0 * W + x0), possibly outside
        int cx0 = 0, cy0 = 0;
        if (v < a.V) {
            const float *P = a.proj + (size_t)v * 16;
            // q = [x y z 1] P^T (structures/bbox_3d/utils.py:322-327), one rounding per step, in this order
            float q[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                q[r] = __fadd_rn(fmaf(z, P[4 * r + 2], fmaf(y, P[4 * r + 1], __fmul_rn(x, P[4 * r]))), P[4 * r + 3]);
            const float zc = fmaxf(q[2], 1e-3f);
            float cx = __fsub_rn(__fmul_rn(__fdiv_rn(q[0], zc), a.sx), a.cx);        // scale -> crop (point_fusion.py:266-267)
            const float cy = __fsub_rn(__fmul_rn(__fdiv_rn(q[1], zc), a.sy), a.cy);
            if (a.flip) cx = __fsub_rn(a.ori_w, cx);                                  // horizontal flip (:276)
            const float nx = __fsub_rn(__fmul_rn(__fdiv_rn(cx, a.pad_w), 2.0f), 1.0f);
            const float ny = __fsub_rn(__fmul_rn(__fdiv_rn(cy, a.pad_h), 2.0f), 1.0f);
            // grid_sample, align_corners=True (unnormalise: ((g + 1) / 2) * (size - 1)), zeros padding
            const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(nx, 1.0f), 2.0f), (float)(a.W - 1));
            const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(ny, 1.0f), 2.0f), (float)(a.H - 1));
            if (!a.bilinear) {                                  // nearest: round half to even
                const float fx = rintf(ix), fy = rintf(iy);
                inb = fx >= 0.0f && fx <= (float)(a.W - 1) && fy >= 0.0f && fy <= (float)(a.H - 1);
                if (inb) pix = (int)fy * a.W + (int)fx;
            } else {                                            // the four neighbours, those outside the map count as 0
                const float x0 = floorf(ix), y0 = floorf(iy);
                wx1 = __fsub_rn(ix, x0); wy1 = __fsub_rn(iy, y0);
                // a point whose four neighbours are all outside contributes nothing (also covers inf / nan coordinates)
                inb = x0 >= -1.0f && x0 <= (float)(a.W - 1) && y0 >= -1.0f && y0 <= (float)(a.H - 1);
                if (inb) { cx0 = (int)x0; cy0 = (int)y0; }
            }
            valid = cx < a.pad_w && cx > 0.0f && cy < a.pad_h && cy > 0.0f && q[2] > 0.0f;     // :300-301
        }
        nvalid += __popcll(__ballot(valid));
        unsigned long long hit = __ballot(inb);
        while (hit) {                                           // views in ascending order (the reference sums dim 0)
            const int l = __ffsll((long long)hit) - 1;
            hit &= hit - 1;
            if (!a.bilinear) {
                const int px = __builtin_amdgcn_readlane(pix, l);
                const float *f = a.featT + ((size_t)(v0 + l) * HW + px) * a.C;
#pragma unroll
                for (int q = 0; q < kPsMaxQ; ++q) {
                    const int c = lane + 64 * q;
                    if (c < a.C) acc[q] += f[c];
                }
            } else {
                const int x0 = __builtin_amdgcn_readlane(cx0, l), y0 = __builtin_amdgcn_readlane(cy0, l);
                const float fx = PTX_LANE_F(wx1, l), fy = PTX_LANE_F(wy1, l);
                const float gx = __fsub_rn(1.0f, fx), gy = __fsub_rn(1.0f, fy);
                // weights as torch's grid sampler forms them: nw = (x1 - ix)(y1 - iy), ne = (ix - x0)(y1 - iy), ...
                const float w[4] = {__fmul_rn(gx, gy), __fmul_rn(fx, gy), __fmul_rn(gx, fy), __fmul_rn(fx, fy)};
                const float *fv = a.featT + (size_t)(v0 + l) * HW * a.C;
                float s[kPsMaxQ];
#pragma unroll
                for (int q = 0; q < kPsMaxQ; ++q) s[q] = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {                   // nw, ne, sw, se
                    const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
                    if (xx < 0 || xx >= a.W || yy < 0 || yy >= a.H) continue;      // wave-uniform
                    const float *f = fv + ((size_t)yy * a.W + xx) * a.C;
#pragma unroll
                    for (int q = 0; q < kPsMaxQ; ++q) {
                        const int c = lane + 64 * q;
                        if (c < a.C) s[q] = __fadd_rn(s[q], __fmul_rn(f[c], w[k]));
                    }
                }
#pragma unroll
                for (int q = 0; q < kPsMaxQ; ++q) acc[q] += s[q];
            }
        }
    }
    const float den = (float)(nvalid > 1 ? nvalid : 1);
#pragma unroll
    for (int q = 0; q < kPsMaxQ; ++q) {
        const int c = lane + 64 * q;
        if (c < a.C) a.out[(size_t)n * a.C + c] = nvalid > 0 ? __fdiv_rn(acc[q], den) : 0.0f;
    }
    if (a.valid_num && lane == 0) a.valid_num[n] = nvalid;
}

}  // namespace ptx

using namespace ptx;

extern "C" {

/* out (cols, rows) = in (rows, cols)^T, fp32 (train mode: transposed weights for the input-gradient GEMMs) */
int ptx_op_transpose(const float *in, int rows, int cols, float *out, void *stream)
{
    PTX_REQUIRE(in && out && rows >= 1 && cols >= 1, "ptx_op_transpose: bad arguments");
    hipLaunchKernelGGL(k_feat_transpose, dim3(cdiv(cols, 32), cdiv(rows, 32), 1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       in, 0, rows, cols, out);
    PTX_LAUNCHED("k_feat_transpose");
    return PTX_OK;
}

size_t ptx_point_sample_workspace_bytes(int V, int C, int H, int W)
{
    if (V < 1 || C < 1 || C > 64 * kPsMaxQ || H < 1 || W < 1) return 0;
    return align_up((size_t)V * C * H * W * sizeof(float), 256);
}

/* the channels-last copy alone (ABI 12): what ptx_point_sample does first.  A caller that knows the feature maps before it knows the
 * points (the detector: the 2D backbone runs before the neck) makes the copies early, on another stream, and samples with feats = NULL */
int ptx_point_sample_prepare(const void *feats, int feat_dtype, int V, int C, int H, int W, void *workspace, size_t ws_bytes, void *stream)
{
    PTX_REQUIRE(feats && workspace && V >= 1 && C >= 1 && C <= 64 * kPsMaxQ && H >= 1 && W >= 1 && feat_dtype >= 0 && feat_dtype <= 2,
                "ptx_point_sample_prepare: V=%d C=%d (<= %d) H=%d W=%d dtype=%d", V, C, 64 * kPsMaxQ, H, W, feat_dtype);
    const size_t need = ptx_point_sample_workspace_bytes(V, C, H, W);
    if (ws_bytes < need) { set_error("ptx_point_sample_prepare: workspace too small: %zu < %zu bytes", ws_bytes, need); return PTX_ENOSPACE; }
    hipLaunchKernelGGL(k_feat_transpose, dim3(cdiv(H * W, 32), cdiv(C, 32), V), dim3(256), 0, static_cast<hipStream_t>(stream), feats,
                       feat_dtype, C, H * W, static_cast<float *>(workspace));
    PTX_LAUNCHED("k_feat_transpose");
    return PTX_OK;
}

int ptx_point_sample(const float *points, int N, const void *feats, int feat_dtype, int V, int C, int H, int W,
                     const float *proj, const float *pre, float scale_w, float scale_h, float crop_w, float crop_h, int flip,
                     float ori_w, float pad_h, float pad_w, int bilinear, float *out, int32_t *valid_num, void *workspace,
                     size_t ws_bytes, void *stream)
{
    PTX_REQUIRE(points && proj && out && workspace, "ptx_point_sample: null argument");       // feats == NULL: workspace prepared
    PTX_REQUIRE(N >= 1 && V >= 1 && C >= 1 && C <= 64 * kPsMaxQ && H >= 1 && W >= 1 && feat_dtype >= 0 && feat_dtype <= 2 &&
                pad_h > 0.0f && pad_w > 0.0f, "ptx_point_sample: N=%d V=%d C=%d (<= %d) H=%d W=%d dtype=%d", N, V, C,
                64 * kPsMaxQ, H, W, feat_dtype);
    const size_t need = ptx_point_sample_workspace_bytes(V, C, H, W);
    if (ws_bytes < need) { set_error("ptx_point_sample: workspace too small: %zu < %zu bytes", ws_bytes, need); return PTX_ENOSPACE; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *featT = static_cast<float *>(workspace);
    if (feats != nullptr) {
        hipLaunchKernelGGL(k_feat_transpose, dim3(cdiv(H * W, 32), cdiv(C, 32), V), dim3(256), 0, st, feats, feat_dtype, C, H * W, featT);
        PTX_LAUNCHED("k_feat_transpose");
    }
    PsArgs a{points, N, featT, V, C, H, W, proj, pre, scale_w, scale_h, crop_w, crop_h, flip, ori_w, pad_h, pad_w, out, valid_num, bilinear ? 1 : 0};
    hipLaunchKernelGGL(k_point_sample, dim3(cdiv(N, 4)), dim3(256), 0, st, a);
    PTX_LAUNCHED("k_point_sample");
    return PTX_OK;
}

}  // extern "C"
