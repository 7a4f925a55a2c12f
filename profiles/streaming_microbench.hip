#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
template<int LDSW, int ROWS>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, uint32_t* out, int P16, int doload){
  __shared__ uint32_t l[LDSW];
  uint32_t acc=0;
  if (doload){
    const uint4* p = in + (size_t)blockIdx.x*4*ROWS*P16 + (threadIdx.x>>6)*ROWS*P16 + (threadIdx.x&63);
    #pragma unroll
    for(int k=0;k<ROWS+2;++k){ uint4 v=p[k*P16]; acc|=v.x|v.y|v.z|v.w; }
  }
  if (LDSW>1 && acc==0x12345) l[threadIdx.x%LDSW]=acc;
  if (acc==0xdeadbeef) out[blockIdx.x]=acc + (LDSW>1? l[0]:0);
}
template<int LDSW,int ROWS> float run(const uint4* in, uint32_t* out, int blocks, int doload){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int i=0;i<3;++i) hipLaunchKernelGGL((k<LDSW,ROWS>),dim3(blocks),dim3(256),0,0,in,out,64,doload);
  hipEventRecord(a,0);
  for(int i=0;i<20;++i) hipLaunchKernelGGL((k<LDSW,ROWS>),dim3(blocks),dim3(256),0,0,in,out,64,doload);
  hipEventRecord(b,0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); return ms/20*1000;
}
int main(){
  size_t bytes=(size_t)256*1026*1024+ (1<<20);
  uint4* in; uint32_t* out; CHK(hipMalloc(&in,bytes)); CHK(hipMalloc(&out,1<<20)); CHK(hipMemset(in,0,bytes));
  printf("empty  lds0   32768 blk: %.1f us\n", run<1,2>(in,out,32768,0));
  printf("empty  lds16K 32768 blk: %.1f us\n", run<4100,2>(in,out,32768,0));
  printf("empty  lds16K  8192 blk: %.1f us\n", run<4100,8>(in,out,8192,0));
  printf("load rb2 lds0   32768 blk: %.1f us\n", run<1,2>(in,out,32768,1));
  printf("load rb2 lds16K 32768 blk: %.1f us\n", run<4100,2>(in,out,32768,1));
  printf("load rb8 lds0    8192 blk: %.1f us\n", run<1,8>(in,out,8192,1));
  printf("load rb8 lds16K  8192 blk: %.1f us\n", run<4100,8>(in,out,8192,1));
  printf("load rb16 lds16K 4096 blk: %.1f us\n", run<4100,16>(in,out,4096,1));
  printf("load rb32 lds16K 2048 blk: %.1f us\n", run<4100,32>(in,out,2048,1));
  return 0;
}
