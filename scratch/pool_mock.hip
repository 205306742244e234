// Lab (r04): does a pooling unit's compute phase overlap other units' loads?  Units load their feature rows, then "compute" for D us
// (a bounded spin on s_memrealtime: no memory traffic), then store their partials.  Variants:
//   tile8x2   : (image, 128-pixel tile) units, 8 waves, two work-groups per CU            -- the shipped decomposition
//   img16     : whole images, 16 waves, one work-group per CU, rows at their own alignment (2 adjacent rows per instruction)
//   img16p    : the same, persistent (256 work-groups, image b, b + 256, ...): the next image's rows are requested BEFORE the
//               compute phase of the current one (upper bound of an in-work-group prefetch: needs a second register set)
//   img16h    : persistent, the next image requested after HALF of the compute phase (registers free up in the last stage)
// usage: pool_mock <nimg> <D_us x10>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4u2 __attribute__((ext_vector_type(4), aligned(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned fold(const u32x4 &v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; }
constexpr int HW = 225, CH = 512;
static int LDSB = 43000;
__device__ __forceinline__ void spin(unsigned ticks) {            // 100 MHz ticks
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

__global__ __launch_bounds__(512) void k_tile(const unsigned short *img, int nimg, float *out, unsigned ticks, int store_f4, int mode, const float *we)
{
    extern __shared__ unsigned char smem[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, T = slot & 1, imr = (slot >> 1) * 8 + xcd;
    if (imr >= nimg) return;
    const int im = nimg - 1 - imr, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    const unsigned short *f = img + (size_t)im * CH * HW;
    unsigned acc = 0;
    u32x4 L[16];
    float wv[8];
    if (mode == 3) {
#pragma unroll
        for (int h = 0; h < 8; ++h) wv[h] = we[((size_t)im * 8 + h) * 738 + threadIdx.x];
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            L[8 * kb + i] = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(f + (size_t)(64 * wid + 32 * kb + 8 * kq + i) * HW + 128 * T + 8 * n));
    if (mode == 3) {
        unsigned short *wp = reinterpret_cast<unsigned short *>(smem);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const unsigned u = __float_as_uint(wv[h]);
            wp[h * 544 + threadIdx.x] = (unsigned short)(u >> 16); wp[(8 + h) * 544 + threadIdx.x] = (unsigned short)u; wp[(16 + h) * 544 + threadIdx.x] = (unsigned short)(u >> 8);
        }
        __syncthreads();
        acc ^= *reinterpret_cast<const unsigned *>(wp + (lane & 7) * 544 + 64 * wid + 8 * kq);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= fold(L[i]);
    __syncthreads();
    float *dst = out + (size_t)(im * 2 + T) * store_f4 * 4;
    const f32x4 v = {(float)acc, 1.f, 2.f, 3.f};
    if (mode == 1)
        for (int i = threadIdx.x; i < store_f4 / 2; i += 512) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + 4 * i), "v"(v) : "memory");
    spin(ticks);
    __syncthreads();
    for (int i = threadIdx.x + (mode == 1 ? store_f4 / 2 : 0); i < store_f4; i += 512) {
        if (mode == 2) *reinterpret_cast<f32x4 *>(dst + 4 * i) = v;
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + 4 * i), "v"(v) : "memory");
    }
}

template <int MODE>      // 0: one image per work-group; 1: persistent, next image requested before the compute phase; 2: after half of it
__global__ __launch_bounds__(1024) void k_img16(const unsigned short *img, int nimg, float *out, unsigned ticks, int store_f4)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    const int step = MODE == 0 ? nimg : gridDim.x;
    unsigned acc = 0;
    u32x4 L[16];
    auto req = [&](int im, u32x4 (&R)[16]) {
        const unsigned short *f = img + (size_t)im * CH * HW;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            R[i] = __builtin_nontemporal_load(reinterpret_cast<const u4u2 *>(f + (size_t)(32 * wid + 2 * i + h) * HW + (n < 28 ? 8 * n : HW - 8)));
    };
    int imr = blockIdx.x;
    if (imr >= nimg) return;
    req(nimg - 1 - imr, L);
    for (; imr < nimg; imr += step) {
        const int im = nimg - 1 - imr, nxt = imr + step;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= fold(L[i]);
        __syncthreads();
        if (MODE == 1 && nxt < nimg) req(nimg - 1 - nxt, L);     // (L is dead: folded above)
        if (MODE == 2) { spin(ticks / 2); if (nxt < nimg) req(nimg - 1 - nxt, L); spin(ticks - ticks / 2); }
        else spin(ticks);
        __syncthreads();
        float *dst = out + (size_t)im * store_f4 * 4;
        for (int i = threadIdx.x; i < store_f4; i += 1024) {
            const f32x4 v = {(float)acc, 1.f, 2.f, 3.f};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + 4 * i), "v"(v) : "memory");
        }
    }
}

int main(int argc, char **argv)
{
    const int nimg = argc > 1 ? atoi(argv[1]) : 784;
    if (argc > 2) LDSB = atoi(argv[2]);
    const size_t bytes = (size_t)nimg * CH * HW * 2;
    unsigned short *img[3]; float *out;
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&img[k], bytes + 4096)); CK(hipMemset(img[k], 0x3f, bytes + 4096)); }
    CK(hipMalloc(&out, (size_t)nimg * 2 * 24 * 1024));
    float *we; CK(hipMalloc(&we, (size_t)nimg * 8 * 738 * 4 + 4096)); CK(hipMemset(we, 0, (size_t)nimg * 8 * 738 * 4 + 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 30;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_img16<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_img16<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_img16<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
    for (int d10 : {0, 20, 35, 50}) {
        const unsigned ticks = d10 * 10;                  // 0.1 us = 10 ticks of 10 ns
        for (int v = 0; v < 8; ++v) {
            // compute per unit: a tile unit does half an image's work -> D; an image unit -> 1.6 D (one soft-max, one weight prologue)
            const char *name = v == 0 ? "tile8x2 (20 KB partials per unit)" : v == 1 ? "img16   (24 KB per image)" : v == 2 ? "img16p  (persistent, full prefetch)"
                             : v == 3 ? "img16h  (persistent, prefetch at half)" : v == 4 ? "tile8x2 no stores" : v == 5 ? "tile8x2 half of the stores before D"
                             : v == 6 ? "tile8x2 plain stores" : "tile8x2 + weight prologue";
            if (v == 2 || v == 3) continue;
            const unsigned tk = (v != 1) ? ticks : (unsigned)(ticks * 1.6);
            auto launch = [&](int k) {
                const unsigned short *p = img[k % 3];
                if (v == 0) hipLaunchKernelGGL(k_tile, dim3((nimg + 7) / 8 * 16), dim3(512), LDSB, 0, p, nimg, out, tk, 1280, 0, we);
                else if (v == 4) hipLaunchKernelGGL(k_tile, dim3((nimg + 7) / 8 * 16), dim3(512), LDSB, 0, p, nimg, out, tk, 0, 0, we);
                else if (v == 5) hipLaunchKernelGGL(k_tile, dim3((nimg + 7) / 8 * 16), dim3(512), LDSB, 0, p, nimg, out, tk, 1280, 1, we);
                else if (v == 6) hipLaunchKernelGGL(k_tile, dim3((nimg + 7) / 8 * 16), dim3(512), LDSB, 0, p, nimg, out, tk, 1280, 2, we);
                else if (v == 7) hipLaunchKernelGGL(k_tile, dim3((nimg + 7) / 8 * 16), dim3(512), LDSB, 0, p, nimg, out, tk, 1280, 3, we);
                else if (v == 1) hipLaunchKernelGGL(k_img16<0>, dim3(nimg), dim3(1024), 160000, 0, p, nimg, out, tk, 1536);
                else if (v == 2) hipLaunchKernelGGL(k_img16<1>, dim3(256), dim3(1024), 160000, 0, p, nimg, out, tk, 1536);
                else hipLaunchKernelGGL(k_img16<2>, dim3(256), dim3(1024), 160000, 0, p, nimg, out, tk, 1536);
            };
            for (int k = 0; k < 6; ++k) launch(k);
            CK(hipEventRecord(a));
            for (int k = 0; k < iters; ++k) launch(k);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const float us = ms / iters * 1e3f;
            printf("D=%.1f us  %-40s %7.1f us per launch  %5.2f TB/s\n", d10 / 10.0, name, us, bytes / us * 1e-6);
        }
    }
    return 0;
}
