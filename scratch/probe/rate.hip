// Back-to-back dependent launch cost without a profiler: stream launches vs a captured graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_empty(const float *a, float *c) {}
__global__ void k_scalar(const float *a, float *c, int n) { c[threadIdx.x + blockIdx.x * 64] = a[(threadIdx.x + blockIdx.x * 64) % n] + 1.0f; }
__global__ void k_big(const float *a, float *c, int n) { c[threadIdx.x + blockIdx.x * 256] = a[(threadIdx.x + blockIdx.x * 256) % n] + 1.0f; }
int main() {
    float *a, *c; CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&c, 64 << 20)); CK(hipMemset(a, 0, 64 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(64), dim3(64), 0, st, a, c);
                else if (variant == 1) hipLaunchKernelGGL(k_scalar, dim3(64), dim3(64), 0, st, a, c, 1024);
                else hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 0, st, a, c, 1 << 20);
            }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("stream variant %d: %.2f us per launch\n", variant, ms * 1e3 / N);
        }
    }
    // graph of 20 dependent kernels
    for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 20; ++i) {
            if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(64), dim3(64), 0, st, a, c);
            else if (variant == 1) hipLaunchKernelGGL(k_scalar, dim3(64), dim3(64), 0, st, a, c, 1024);
            else hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 0, st, a, c, 1 << 20);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 100; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("graph variant %d: %.2f us per kernel node\n", variant, ms * 1e3 / 2000);
        }
    }
    return 0;
}
