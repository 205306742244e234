// Fixed cost of a dependent kernel: empty / scalar-argument chain / struct-argument chain (rocprofv3 durations).
#include <hip/hip_runtime.h>
#include <cstdio>
struct Prob { const float *a; const float *w; float *c; int r, n, k, lda; float pad[20]; };
struct Batch { Prob p[8]; int n; };
__global__ void k_empty(const float *a, float *c) {}
__global__ void k_scalar(const float *a, float *c, int n) { c[threadIdx.x + blockIdx.x * 64] = a[(threadIdx.x + blockIdx.x * 64) % n] + 1.0f; }
__global__ void k_struct(Batch b) { const Prob pr = b.p[blockIdx.z]; pr.c[threadIdx.x + blockIdx.x * 64] = pr.a[(threadIdx.x + blockIdx.x * 64) % pr.n] + 1.0f; }
__global__ void k_chain2(const float *a, const int *idx, float *c, int n) { const int i = idx[(threadIdx.x + blockIdx.x * 64) % n]; c[threadIdx.x + blockIdx.x * 64] = a[i] + 1.0f; }
__global__ void k_stream(const float4 *a, float4 *c, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i]; }
int main() {
    float *a, *c; int *idx; float4 *big, *big2;
    hipMalloc(&a, 1 << 20); hipMalloc(&c, 1 << 20); hipMalloc(&idx, 1 << 20); hipMemset(idx, 0, 1 << 20); hipMemset(a, 0, 1 << 20);
    hipMalloc(&big, 256u << 20); hipMalloc(&big2, 256u << 20);
    Batch b{}; b.n = 2; for (int i = 0; i < 8; ++i) { b.p[i].a = a; b.p[i].c = c; b.p[i].n = 1024; }
    hipStream_t st; hipStreamCreate(&st);
    for (int it = 0; it < 40; ++it) {
        if (it >= 20) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, st, big, big2, (size_t)(192u << 20) / 16);   // cold half
        hipLaunchKernelGGL(k_empty, dim3(64), dim3(64), 0, st, a, c);
        hipLaunchKernelGGL(k_scalar, dim3(64), dim3(64), 0, st, a, c, 1024);
        hipLaunchKernelGGL(k_struct, dim3(64, 1, 2), dim3(64), 0, st, b);
        hipLaunchKernelGGL(k_chain2, dim3(64), dim3(64), 0, st, a, idx, c, 1024);
    }
    hipStreamSynchronize(st);
    printf("done\n");
    return 0;
}
