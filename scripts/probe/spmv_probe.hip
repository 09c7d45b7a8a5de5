// Ablation probe for the SpMV kernel on a 256^3 7-point Laplacian (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "../../lanczos_amd/csrc/kernels.hip.h"
using namespace lz;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void gen_rp(int N, int* rp_cnt){ int64_t n=(int64_t)N*N*N; for(int64_t i=blockIdx.x*(int64_t)blockDim.x+threadIdx.x;i<n;i+=(int64_t)gridDim.x*blockDim.x){int x=i%N,y=(i/N)%N,z=i/((int64_t)N*N); rp_cnt[i]=1+(x>0)+(x<N-1)+(y>0)+(y<N-1)+(z>0)+(z<N-1);} }
__global__ void gen_fill(int N,const int* rp,int* col,double* val){ int64_t n=(int64_t)N*N*N; for(int64_t i=blockIdx.x*(int64_t)blockDim.x+threadIdx.x;i<n;i+=(int64_t)gridDim.x*blockDim.x){int x=i%N,y=(i/N)%N,z=i/((int64_t)N*N); int p=rp[i];
  if(z>0){col[p]=i-N*N;val[p++]=-1;} if(y>0){col[p]=i-N;val[p++]=-1;} if(x>0){col[p]=i-1;val[p++]=-1;} col[p]=i;val[p++]=6; if(x<N-1){col[p]=i+1;val[p++]=-1;} if(y<N-1){col[p]=i+N;val[p++]=-1;} if(z<N-1){col[p]=i+N*N;val[p++]=-1;} } }

// variant kernel: MODE bit0 = gather x, bit1 = LDS row sum, bit2 = skip col load
template<int MODE>
__global__ void __launch_bounds__(256) probe(int64_t n,const int* __restrict__ rp,const int* __restrict__ col,const double* __restrict__ val,const double* __restrict__ x,double* __restrict__ y){
  __shared__ double prod[4][kTile];
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  const int64_t groups=(n+63)>>6; const int64_t total=(groups+3)/4; const int64_t per_xcd=(total+7)>>3;
  const int64_t xcd=blockIdx.x&7, slot=blockIdx.x>>3, slots=gridDim.x>>3; int64_t c_end=(xcd+1)*per_xcd; if(c_end>total)c_end=total;
  double* myprod=prod[wave];
  int64_t cbeg=xcd*per_xcd+slot, cstep=slots; if(MODE&32){cbeg=blockIdx.x; cstep=gridDim.x; c_end=total;}
  for(int64_t c=cbeg;c<c_end;c+=cstep){
    const int64_t row=((c*4+wave)<<6)+lane;
    int my_lo,my_hi;
    if(MODE&16){ int64_t r0=row<n?row:n, r1=row+1<n?row+1:n; my_lo=(int)(r0*6+ (r0>>4)); my_hi=(int)(r1*6+(r1>>4)); }
    else { my_lo=rp[row<n?row:n]; my_hi=rp[row+1<n?row+1:n]; }
    const int span_lo=__shfl(my_lo,0), span_hi=__shfl(my_hi,63);
    double sum=0;
    for(int t0=span_lo;t0<span_hi;t0+=kTile){
      const int last=span_hi-1; int cj[kTileU]; double vj[kTileU],xj[kTileU];
#pragma unroll
      for(int u=0;u<kTileU;++u){int j=t0+u*64+lane; j=j<last?j:last; cj[u]=(MODE&4)?j:col[j]; vj[u]=val[j];}
#pragma unroll
      for(int u=0;u<kTileU;++u) xj[u]=(MODE&1)?x[cj[u]]:(double)cj[u];
      if(MODE&2){
#pragma unroll
        for(int u=0;u<kTileU;++u) myprod[u*64+lane]=vj[u]*xj[u];
        __builtin_amdgcn_wave_barrier();
        const int t1=(span_hi-t0>kTile)?t0+kTile:span_hi; const int lo=my_lo>t0?my_lo:t0; const int hi=my_hi<t1?my_hi:t1;
        for(int j=lo;j<hi;++j) sum+=myprod[j-t0];
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
        for(int u=0;u<kTileU;++u) sum+=vj[u]*xj[u];
      }
    }
    if(MODE&8){ if(sum==1.2345e-300) y[row<n?row:0]=sum; } else if(row<n) y[row]=sum;
  }
}
// pure streaming reference: read val (8B/lane) + col (4B/lane) fully coalesced grid-stride, write y
__global__ void __launch_bounds__(256) stream_ref(int64_t nnz,const int* __restrict__ col,const double* __restrict__ val,double* __restrict__ y,int64_t n){
  double s=0; int64_t stride=(int64_t)gridDim.x*256;
#pragma unroll 8
  for(int64_t i=blockIdx.x*256+threadIdx.x;i<nnz;i+=stride){ s+=val[i]*(double)col[i]; }
  int64_t i=blockIdx.x*256+threadIdx.x; if(i<n) y[i]=s;
}
__global__ void __launch_bounds__(256) stream_ref16(int64_t nnz2,const int2* __restrict__ col,const double2* __restrict__ val,double* __restrict__ y,int64_t n){
  double s=0; int64_t stride=(int64_t)gridDim.x*256;
#pragma unroll 8
  for(int64_t i=blockIdx.x*256+threadIdx.x;i<nnz2;i+=stride){ double2 v=val[i]; int2 c=col[i]; s+=v.x*(double)c.x+v.y*(double)c.y; }
  int64_t i=blockIdx.x*256+threadIdx.x; if(i<n) y[i]=s;
}

int main(int argc,char**argv){
  int N=256; int64_t n=(int64_t)N*N*N; int *rp,*col; double *val,*x,*y,*part;
  CK(hipMalloc(&rp,4*(n+1))); CK(hipMemset(rp,0,4));
  gen_rp<<<2048,256>>>(N,rp+1);
  std::vector<int> h(n+1); CK(hipMemcpy(h.data(),rp,4*(n+1),hipMemcpyDeviceToHost)); for(int64_t i=0;i<n;i++)h[i+1]+=h[i]; int64_t nnz=h[n];
  CK(hipMemcpy(rp,h.data(),4*(n+1),hipMemcpyHostToDevice));
  CK(hipMalloc(&col,4*nnz)); CK(hipMalloc(&val,8*nnz)); CK(hipMalloc(&x,8*n)); CK(hipMalloc(&y,8*n)); CK(hipMalloc(&part,8*4096));
  gen_fill<<<2048,256>>>(N,rp,col,val); CK(hipMemset(x,0,8*n)); CK(hipDeviceSynchronize());
  double bytes=12.0*nnz+4.0*(n+1)+16.0*n; printf("n=%lld nnz=%lld B_spmv=%.0f\n",(long long)n,(long long)nnz,bytes);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto timeit=[&](const char* name, auto fn){ for(int i=0;i<3;i++)fn(); hipEventRecord(a); for(int i=0;i<20;i++)fn(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); ms/=20; printf("%-28s %.4f ms  %.0f GB/s (algorithmic)\n",name,ms,bytes/ms/1e6); };
  for(int grid : {2048}){
    printf("grid %d\n",grid);
    timeit("full(gather+lds)",[&]{probe<3><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("gather,no lds",[&]{probe<1><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("no gather,lds",[&]{probe<2><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("no gather,no lds",[&]{probe<0><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("no col,no gather,no lds",[&]{probe<4><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("A base (no gather,no lds)",[&]{probe<0><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("B  A + no store",[&]{probe<8><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("C  A + synthetic rp",[&]{probe<16><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("D  A + no xcd map",[&]{probe<32><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("E  A + no store+synth rp",[&]{probe<24><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("F  full + no xcd map",[&]{probe<35><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("G  gather only + noxcd",[&]{probe<33><<<grid,256>>>(n,rp,col,val,x,y);});
    timeit("stream_ref 8B/4B",[&]{stream_ref<<<grid,256>>>(nnz,col,val,y,n);});
    timeit("stream_ref 16B/8B",[&]{stream_ref16<<<grid,256>>>(nnz/2,(const int2*)col,(const double2*)val,y,n);});
    int np=0; (void)np;
    timeit("product pipe64",[&]{k_spmv_pipe<int,6,false><<<grid,256>>>(n,rp,col,val,x,x,y,part);});
    timeit("product stream",[&]{k_spmv_stream<int><<<grid,256>>>(n,rp,col,val,x,x,y,part);});
  }
  return 0;
}
