
// What does __builtin_readcyclecounter() count on gfx950, and how fast is the shader clock under a streaming load?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4 *in, unsigned long long *out, int n)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64(), m0 = clock64();
    float4 acc = {0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64(), m1 = clock64();
    if (acc.x == 1.2345f) out[0] = 1;
    if (threadIdx.x == 0) { out[blockIdx.x * 3 + 1] = c1 - c0; out[blockIdx.x * 3 + 2] = w1 - w0; out[blockIdx.x * 3 + 3] = m1 - m0; }
}
int main()
{
    const int n = 64 << 20;   // 1 GiB of float4
    float4 *in; unsigned long long *out;
    hipMalloc(&in, (size_t)n * 16); hipMemset(in, 0, (size_t)n * 16);
    hipMalloc(&out, 8 * (3 * 2048 + 4));
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    int cc = 0; hipDeviceGetAttribute(&cc, hipDeviceAttributeClockRate, 0);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, in, out, n);
    hipDeviceSynchronize();
    unsigned long long h[3 * 2048 + 4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0, w = 0, m = 0;
    for (int b = 0; b < 2048; ++b) { c += h[b * 3 + 1]; w += h[b * 3 + 2]; m += h[b * 3 + 3]; }
    printf("wall clock rate attr: %d kHz, clock rate attr: %d kHz\n", wc, cc);
    printf("per block: readcyclecounter %.0f, wall_clock64 %.0f, clock64 %.0f\n", c / 2048, w / 2048, m / 2048);
    printf("=> readcyclecounter ticks at %.1f MHz, clock64 at %.1f MHz (wall clock = %d kHz)\n",
           c / w * wc / 1e3, m / w * wc / 1e3, wc);
    return 0;
}
