// How many bytes must be in flight per CU to saturate HBM reads? (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
template<int U>
__global__ void __launch_bounds__(256) rd(const double* __restrict__ val,const int* __restrict__ col,int64_t nnz,double* out){
  extern __shared__ double pad[];
  double s=0; const int64_t stride=(int64_t)gridDim.x*256*U;
  for(int64_t i=(int64_t)blockIdx.x*256*U+threadIdx.x;i<nnz;i+=stride){
    double v[U]; int c[U];
#pragma unroll
    for(int u=0;u<U;++u){int64_t j=i+u*256; j=j<nnz?j:nnz-1; v[u]=val[j]; c[u]=col[j];}
#pragma unroll
    for(int u=0;u<U;++u) s+=v[u]*(double)c[u];
  }
  if(s==123.456) out[0]=s+pad[0];
}
int main(){
  int64_t nnz=117047296; double *val,*out; int *col;
  CK(hipMalloc(&val,8*nnz)); CK(hipMalloc(&col,4*nnz)); CK(hipMalloc(&out,64)); CK(hipMemset(val,0,8*nnz)); CK(hipMemset(col,0,4*nnz));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto run=[&](auto kern,int U,int wavesPerCU){
    // limit occupancy with dynamic LDS: blocks/CU = wavesPerCU/4
    int blocksPerCU=wavesPerCU/4; size_t lds=(160*1024/blocksPerCU) - 1024; if(lds>64*1024){ CK(hipFuncSetAttribute((const void*)kern,hipFuncAttributeMaxDynamicSharedMemorySize,(int)lds)); }
    int grid=256*blocksPerCU;
    for(int i=0;i<3;i++) kern<<<grid,256,lds>>>(val,col,nnz,out);
    hipEventRecord(a); for(int i=0;i<20;i++) kern<<<grid,256,lds>>>(val,col,nnz,out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms,a,b); ms/=20; CK(hipGetLastError());
    printf("U=%d waves/CU=%2d inflight/CU=%4d KB : %.4f ms %.0f GB/s\n",U,wavesPerCU,(int)(wavesPerCU*U*64*12/1024),ms,12.0*nnz/ms/1e6);
  };
  for(int w: {4,8,16,32}){ run(rd<1>,1,w); run(rd<2>,2,w); run(rd<4>,4,w); run(rd<8>,8,w); run(rd<16>,16,w);}
  return 0;
}
