// Lab: how many L2 requests, and how long, do the load maps of an attention-pool unit take when nothing else happens?
// 784 images x 512 rows x 225 bf16 (rows of 450 B, 2-byte aligned), three copies rotated (cold), loads + an xor fold only.
//   tile      : the shipped map -- unit (image, 128-pixel tile), 8 waves, a load instruction = 4 rows x 256 B (rows 8 apart)
//   rowpair16 : whole rows -- unit = image, 16 waves; wave = (row class c mod 8, half of the 64 row groups); an instruction = 2 rows
//               of one class (c, c + 8: 3600 B apart, so their 16-B windows line up) x 29 windows of 16 B starting at the 16-B
//               boundary below the row; one work-group per CU
//   rowpair8  : the same map as two 8-wave work-groups per image (the two halves of the row groups), two per CU
//   contig    : 1 KB contiguous per instruction, 16 KB per wave (what the mean pass does): the floor
// Run alone for times, and under  rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum  for requests.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4u2 __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ unsigned fold(const u32x4 &v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; }
constexpr int HW = 225, CH = 512;

__global__ __launch_bounds__(512) void k_tile(const unsigned short *img, int nimg, float *out)
{
    extern __shared__ unsigned char smem[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, T = slot & 1, imr = (slot >> 1) * 8 + xcd;
    if (imr >= nimg) return;
    const int im = nimg - 1 - imr, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    const unsigned short *f = img + (size_t)im * CH * HW;
    unsigned acc = 0;
    u32x4 L[16];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            L[8 * kb + i] = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(f + (size_t)(64 * wid + 32 * kb + 8 * kq + i) * HW + 128 * T + 8 * n));
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= fold(L[i]);
    if (acc == 0x12345678u) out[blockIdx.x] = smem[0];
}

template <int WAVES>      // 16: unit = image; 8: unit = (image, half of the row groups)
__global__ __launch_bounds__(WAVES * 64) void k_rowpair(const unsigned short *img, int nimg, float *out)
{
    extern __shared__ unsigned char smem[];
    const int per = WAVES == 16 ? 1 : 2;
    const int u = blockIdx.x;
    const int xcd = u & 7, slot = u >> 3;
    const int sub = per == 2 ? (slot & 1) : 0, imr = (per == 2 ? (slot >> 1) : slot) * 8 + xcd;
    if (imr >= nimg) return;
    const int im = nimg - 1 - imr, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rho = wid & 7, kt = WAVES == 16 ? (wid >> 3) : sub;
    const int n = lane & 31, h = lane >> 5;
    const unsigned char *f = reinterpret_cast<const unsigned char *>(img) + (size_t)im * CH * HW * 2;
    unsigned acc = 0;
    u32x4 L[16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = 32 * kt + 16 * s + 2 * i + h;                 // row group: rows 8 t .. 8 t + 7
            L[8 * s + i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(f + (size_t)3600 * t + 448 * rho + 16 * min(n, 28)));
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= fold(L[i]);
    if (acc == 0x12345678u) out[blockIdx.x] = smem[0];
}

// whole rows at their own 2-byte alignment (what k_img_mean16 does): an instruction = 2 adjacent rows x 29 windows of 16 B
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_rows2(const unsigned short *img, int nimg, float *out)
{
    extern __shared__ unsigned char smem[];
    const size_t wave = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);       // 32 rows per wave
    const size_t nrows = (size_t)nimg * CH;
    if (wave * 32 >= nrows) return;
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    unsigned acc = 0;
    u32x4 L[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        L[i] = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(img + (wave * 32 + 2 * i + h) * HW + (n < 28 ? 8 * n : HW - 8)));
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= fold(L[i]);
    if (acc == 0x12345678u) out[blockIdx.x] = smem[0];
}

__global__ __launch_bounds__(512) void k_contig(const unsigned short *img, int nimg, float *out)
{
    extern __shared__ unsigned char smem[];
    // unit = 128 KB contiguous (as much as a tile unit): wave = 16 KB, instruction = 1 KB
    const size_t total = (size_t)nimg * CH * HW * 2;
    const size_t base = (size_t)blockIdx.x * 131072 + (size_t)(threadIdx.x >> 6) * 16384 + (threadIdx.x & 63) * 16;
    unsigned acc = 0;
    u32x4 L[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        size_t o = base + 1024 * i;
        if (o + 16 > total) o = total - 16;
        L[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(img) + o));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= fold(L[i]);
    if (acc == 0x12345678u) out[blockIdx.x] = smem[0];
}

int main(int argc, char **argv)
{
    const int nimg = argc > 1 ? atoi(argv[1]) : 784;
    const size_t bytes = (size_t)nimg * CH * HW * 2;
    unsigned short *img[3]; float *out;
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&img[k], bytes + 4096)); CK(hipMemset(img[k], 0x3f, bytes + 4096)); }
    CK(hipMalloc(&out, 1 << 22));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 30;
    for (int v = 0; v < 9; ++v) {
        const char *name = v == 0 ? "tile      (8 waves, 2 per CU)" : v == 1 ? "rowpair16 (16 waves, 1 per CU)" : v == 2 ? "rowpair8  (8 waves, 2 per CU)"
                         : v == 3 ? "contig    (8 waves, 2 per CU)" : v == 4 ? "rowpair8  (8 waves, 4 per CU: 32 KB LDS)"
                         : v == 5 ? "tile      (8 waves, 1 per CU: 128 KB LDS)" : v == 6 ? "rowpair8  (8 waves, 1 per CU: 128 KB LDS)"
                         : v == 7 ? "rows2     (8 waves, 2-B aligned whole rows)" : "rows2     (4 waves, no LDS: the mean pass)";
        auto launch = [&](int k) {
            const unsigned short *p = img[k % 3];
            if (v == 0) hipLaunchKernelGGL(k_tile, dim3((nimg + 7) / 8 * 16), dim3(512), 65408, 0, p, nimg, out);
            else if (v == 1) hipLaunchKernelGGL(k_rowpair<16>, dim3((nimg + 7) / 8 * 8), dim3(1024), 131072, 0, p, nimg, out);
            else if (v == 2) hipLaunchKernelGGL(k_rowpair<8>, dim3((nimg + 7) / 8 * 16), dim3(512), 65408, 0, p, nimg, out);
            else if (v == 3) hipLaunchKernelGGL(k_contig, dim3((unsigned)((bytes + 131071) / 131072)), dim3(512), 65408, 0, p, nimg, out);
            else if (v == 4) hipLaunchKernelGGL(k_rowpair<8>, dim3((nimg + 7) / 8 * 16), dim3(512), 32768, 0, p, nimg, out);
            else if (v == 5) hipLaunchKernelGGL(k_tile, dim3((nimg + 7) / 8 * 16), dim3(512), 131072, 0, p, nimg, out);
            else if (v == 6) hipLaunchKernelGGL(k_rowpair<8>, dim3((nimg + 7) / 8 * 16), dim3(512), 131072, 0, p, nimg, out);
            else if (v == 7) hipLaunchKernelGGL(k_rows2<8>, dim3((unsigned)(((size_t)nimg * CH / 32 + 7) / 8)), dim3(512), 65408, 0, p, nimg, out);
            else hipLaunchKernelGGL(k_rows2<4>, dim3((unsigned)(((size_t)nimg * CH / 32 + 3) / 4)), dim3(256), 0, 0, p, nimg, out);
        };
        for (int k = 0; k < 6; ++k) launch(k);
        CK(hipEventRecord(a));
        for (int k = 0; k < iters; ++k) launch(k);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const float us = ms / iters * 1e3f;
        printf("%-44s %7.1f us per launch  %6.2f TB/s\n", name, us, bytes / us * 1e-6);
    }
    return 0;
}
