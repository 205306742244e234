// Lab (r05): the pooling pass with its feature tile in LDS instead of registers -- filled by LDS-DMA (global_load_lds_dwordx4),
// double-buffered inside ONE work-group per image (or half image), 64-pixel tiles of 512 channels = 64 KB.  A unit loads its
// tiles, "computes" for D us per tile (a bounded spin: no memory traffic, no issue slots; plus one full read of the tile from
// LDS as the checksum), then stores its partials.  Question: does the launch reach the load floor (~32 us for 784 images) with
// the compute hidden behind the NEXT tile's DMA inside the work-group, and what does the 784-images-on-256-CUs tail cost?
//   dma4 : one work-group per image, 4 tiles          dma2 : per half image, 2 tiles (two work-groups per image)
// usage: pool_dma_mock <nimg>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef HWPX
#define HWPX 225
#endif
constexpr int HW = HWPX, CH = 512, TPX = 64, TILE_B = CH * TPX * 2;
__device__ __forceinline__ void spin(unsigned ticks) {            // 100 MHz ticks
    if (ticks == 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// LDS image of a tile: [channel][8 pieces of 16 B], piece slot s of channel c holds window s ^ (c & 7)
template <int TPW, int NW>      // tiles per work-group, waves per work-group
__global__ __launch_bounds__(NW * 64) void k_dma(const unsigned short *img, int nimg, float *out, unsigned ticks, int store_f4, unsigned *chk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int U = 4 / TPW;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, part = slot % U, imr = (slot / U) * 8 + xcd;
    if (imr >= nimg) return;
    const int im = nimg - 1 - imr, lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned short *f = img + (size_t)im * CH * HW;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    constexpr int IPW = 64 / NW;                                  // DMA instructions per wave and tile (64 x 1 KiB)
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int q = 0; q < IPW; ++q) {
            const int I = wid * IPW + q, c = 8 * I + (lane >> 3), w = (lane & 7) ^ (c & 7);
            glds16(f + (size_t)c * HW + TPX * t + 8 * w, lds0 + buf * TILE_B + I * 1024);
        }
    };
    unsigned acc = 0;
    issue(part * TPW, 0);
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int buf = tt & 1;
        if (tt + 1 < TPW) {
            issue(part * TPW + tt + 1, buf ^ 1);
            if (IPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (IPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // "compute": one full read of the tile from LDS (checksum) + D us
        const u32x4 *tile = reinterpret_cast<const u32x4 *>(smem + buf * TILE_B);
#pragma unroll
        for (int i = 0; i < TILE_B / 16 / (NW * 64); ++i) {
            const u32x4 v = tile[i * NW * 64 + threadIdx.x];
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
        spin(ticks);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // everyone is done with buf before tile tt + 2 lands in it
    }
    if (chk) atomicXor(chk + (im * U + part), acc);
    float *dst = out + (size_t)(im * U + part) * store_f4 * 4;
    const f32x4 v = {(float)acc, 1.f, 2.f, 3.f};
    for (int i = threadIdx.x; i < store_f4; i += NW * 64) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + 4 * i), "v"(v) : "memory");
}

int main(int argc, char **argv)
{
    const int nimg = argc > 1 ? atoi(argv[1]) : 784;
    const size_t elems = (size_t)nimg * CH * HW, bytes = elems * 2;
    std::vector<unsigned short> h(elems + 4096);
    unsigned s = 12345u;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (unsigned short)(s >> 16); }
    unsigned short *img[3]; float *out; unsigned *chk;
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&img[k], bytes + 8192)); CK(hipMemcpy(img[k], h.data(), bytes + 8192, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&out, (size_t)nimg * 4 * 24 * 1024));
    CK(hipMalloc(&chk, (size_t)nimg * 4 * 4)); CK(hipMemset(chk, 0, (size_t)nimg * 4 * 4));
    const int LDS = 2 * TILE_B;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dma<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dma<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dma<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dma<4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    // correctness of the LDS-DMA image (2-byte-aligned sources, windows running on into the next row): checksum of image 3
    hipLaunchKernelGGL((k_dma<4, 8>), dim3((nimg + 7) / 8 * 8), dim3(512), LDS, 0, img[0], nimg, out, 0u, 16, chk);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> hc(nimg);
    CK(hipMemcpy(hc.data(), chk, nimg * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int im : {0, 3, nimg / 2, nimg - 1}) {
        unsigned ref = 0;
        for (int t = 0; t < 4; ++t) for (int c = 0; c < CH; ++c) for (int w = 0; w < 8; ++w) for (int k = 0; k < 8; k += 2) {
            const size_t e = (size_t)im * CH * HW + (size_t)c * HW + TPX * t + 8 * w + k;
            ref ^= (unsigned)h[e] | ((unsigned)h[e + 1] << 16);
        }
        if (ref != hc[im]) { ++bad; printf("checksum MISMATCH image %d: %08x vs %08x\n", im, hc[im], ref); }
    }
    printf("LDS-DMA image checksum: %s\n", bad ? "BAD" : "ok (4 images)");
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 30;
    for (int d10 : {0, 5, 10, 15, 20, 30}) {
        const unsigned ticks = d10 * 10;
        for (int v = 0; v < 5; ++v) {
            const char *name = v == 0 ? "dma4 x 8 waves (image per work-group, 20 KB out)" : v == 1 ? "dma2 x 8 waves (half image, 20 KB out each)"
                             : v == 2 ? "dma4 x 4 waves" : v == 3 ? "dma4 x 16 waves" : "dma4 x 8 waves, no stores";
            auto launch = [&](int k) {
                const unsigned short *p = img[k % 3];
                if (v == 0) hipLaunchKernelGGL((k_dma<4, 8>), dim3((nimg + 7) / 8 * 8), dim3(512), LDS, 0, p, nimg, out, ticks, 1280, nullptr);
                else if (v == 1) hipLaunchKernelGGL((k_dma<2, 8>), dim3((nimg + 7) / 8 * 16), dim3(512), LDS, 0, p, nimg, out, ticks, 1280, nullptr);
                else if (v == 2) hipLaunchKernelGGL((k_dma<4, 4>), dim3((nimg + 7) / 8 * 8), dim3(256), LDS, 0, p, nimg, out, ticks, 1280, nullptr);
                else if (v == 3) hipLaunchKernelGGL((k_dma<4, 16>), dim3((nimg + 7) / 8 * 8), dim3(1024), LDS, 0, p, nimg, out, ticks, 1280, nullptr);
                else hipLaunchKernelGGL((k_dma<4, 8>), dim3((nimg + 7) / 8 * 8), dim3(512), LDS, 0, p, nimg, out, ticks, 0, nullptr);
            };
            for (int k = 0; k < 6; ++k) launch(k);
            CK(hipEventRecord(a));
            for (int k = 0; k < iters; ++k) launch(k);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const float us = ms / iters * 1e3f;
            printf("D=%.1f us/tile  %-52s %7.1f us per launch  %5.2f TB/s\n", d10 / 10.0, name, us, bytes / us * 1e-6);
        }
    }
    return 0;
}
