#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short us8 __attribute__((ext_vector_type(8), aligned(2)));
__global__ void k(const unsigned short* x, int off, unsigned int* out){
  const us8 v = *reinterpret_cast<const us8*>(x + off + threadIdx.x * 8);
  unsigned int s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
}
int main(){
  unsigned short* d; unsigned int* o; hipMalloc(&d, 1<<16); hipMalloc(&o, 64*4);
  unsigned short h[4096]; for (int i=0;i<4096;++i) h[i]=i;
  hipMemcpy(d,h,sizeof(h),hipMemcpyHostToDevice);
  for (int off : {0,1,2,3,5,7}) {
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, off, o); 
    unsigned int r[64]; hipError_t e = hipMemcpy(r,o,sizeof(r),hipMemcpyDeviceToHost);
    unsigned int exp0=0; for(int i=0;i<8;++i) exp0+=off+i;
    unsigned int exp5=0; for(int i=0;i<8;++i) exp5+=off+40+i;
    printf("off=%d err=%d lane0=%u (exp %u) lane5=%u (exp %u)\n", off, (int)e, r[0], exp0, r[5], exp5);
  }
  return 0;
}
