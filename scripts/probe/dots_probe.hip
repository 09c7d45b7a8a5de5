// Variants of the reorthogonalisation GEMV kernels on a resident basis (diagnostic, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "../../lanczos_amd/csrc/kernels.hip.h"
using namespace lz;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

typedef double d2v __attribute__((ext_vector_type(2)));
template<bool NT> __device__ __forceinline__ double2 ld2(const double2* p){ if(NT){ d2v t=__builtin_nontemporal_load((const d2v*)p); return make_double2(t.x,t.y);} return *p; }
template<int EPT,int UL,bool NT>
__global__ void __launch_bounds__(256) dots_v(const double2* __restrict__ V,int64_t ldv2,int l0,int l1,const double2* __restrict__ w,int64_t n2,double* __restrict__ hpart,int hstride){
  extern __shared__ double hs[];
  const int nl=l1-l0; const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  for(int j=threadIdx.x;j<nl*4;j+=256) hs[j]=0.0;
  __syncthreads();
  double* myhs=hs+wave*nl;
  constexpr int TILE=256*EPT;
  const int64_t tiles=(n2+TILE-1)/TILE;
  for(int64_t t=blockIdx.x;t<tiles;t+=gridDim.x){
    const int64_t base=t*TILE+threadIdx.x;
    double2 wt[EPT];
#pragma unroll
    for(int u=0;u<EPT;++u) wt[u]=w[base+u*256];
    const double2* vl=V+(int64_t)l0*ldv2+base;
#pragma unroll UL
    for(int l=0;l<nl;++l,vl+=ldv2){
      double p=0.0;
#pragma unroll
      for(int u=0;u<EPT;++u){ const double2 q=ld2<NT>(vl+u*256); p+=q.x*wt[u].x+q.y*wt[u].y; }
      p=wave_sum(p);
      if(lane==0) myhs[l]+=p;
    }
  }
  __syncthreads();
  for(int j=threadIdx.x;j<nl;j+=256) hpart[(size_t)j*hstride+blockIdx.x]=(hs[j]+hs[nl+j])+(hs[2*nl+j]+hs[3*nl+j]);
}
int main(){
  int64_t n=16777216, ldv=n; int k=96; double *V,*hpart; 
  CK(hipMalloc(&V,8*ldv*(k+1))); CK(hipMalloc(&hpart,8*4096*128)); CK(hipMemset(V,0,8*ldv*(k+1)));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  double bytes=8.0*n*k+8.0*n;
  auto run=[&](const char* name,auto kern,int grid){
    for(int i=0;i<2;i++) kern<<<grid,256,8*4*k>>>((const double2*)V,ldv/2,0,k,(const double2*)(V+ldv*k),ldv/2,hpart,4096);
    hipEventRecord(a); for(int i=0;i<5;i++) kern<<<grid,256,8*4*k>>>((const double2*)V,ldv/2,0,k,(const double2*)(V+ldv*k),ldv/2,hpart,4096); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms,a,b); ms/=5; CK(hipGetLastError());
    printf("grid %4d %-22s %.3f ms %.0f GB/s\n",grid,name,ms,bytes/ms/1e6);
  };
  for(int grid: {1024,2048,4096}){
    run("product k_dots",k_dots,grid);
    run("EPT4 UL2",dots_v<4,2,false>,grid);
    run("EPT4 UL4",dots_v<4,4,false>,grid);
    run("EPT4 UL4 nt",dots_v<4,4,true>,grid);
    run("EPT2 UL4",dots_v<2,4,false>,grid);
    run("EPT2 UL8",dots_v<2,8,false>,grid);
    run("EPT8 UL2",dots_v<8,2,false>,grid);
    run("EPT8 UL2 nt",dots_v<8,2,true>,grid);
    run("EPT1 UL8",dots_v<1,8,false>,grid);
  }
  return 0;
}
