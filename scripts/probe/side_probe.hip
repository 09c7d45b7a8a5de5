// Cost of small side streams (row offsets, y stores) next to the big col/val stream (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
// wave-per-group structure, no data dependencies: group g covers entries [448 g, 448 g + 448) (7 slices of 64; 8th slice clamped)
// MODE bit0: load rp-like ints (2 per lane), bit1: store y (nt if bit2), bit3: read x[row] (8B/lane contiguous)
template<int MODE>
__global__ void __launch_bounds__(256) k(const double* __restrict__ val,const int* __restrict__ col,const int* __restrict__ rp,const double* __restrict__ x,double* __restrict__ y,int64_t groups,int64_t nnz,double* out){
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6; double s=0;
  for(int64_t g=(int64_t)blockIdx.x*4+wave; g<groups; g+=(int64_t)gridDim.x*4){
    const int64_t row=g*64+lane; const int64_t base=g*448;
    double v[8]; int c[8];
#pragma unroll
    for(int u=0;u<8;++u){ int64_t j=base+u*64+lane; j=j<nnz?j:nnz-1; c[u]=col[j]; v[u]=val[j]; }
    int a=0,b=0; if(MODE&1){ a=rp[row]; b=rp[row+1]; }
    double xv=0; if(MODE&8) xv=x[row];
    double t=xv+(double)(a+b);
#pragma unroll
    for(int u=0;u<8;++u) t+=v[u]*(double)c[u];
    if(MODE&16){ double t2=__shfl_down(t,1); if(!(lane&1)){ double2 w=make_double2(t,t2); *(double2*)&y[row]=w; } }
    else if(MODE&32){ double t2=__shfl_down(t,1), t3=__shfl_down(t,2), t4=__shfl_down(t,3); if(!(lane&3)){ double4 w=make_double4(t,t2,t3,t4); *(double4*)&y[row]=w; } }
    else if(MODE&2){ if(MODE&4) __builtin_nontemporal_store(t,&y[row]); else y[row]=t; } else s+=t;
  }
  if(s==123.456) out[0]=s;
}
template<int MODE>
__global__ void __launch_bounds__(256) k2(const double* __restrict__ val,const int* __restrict__ col,double* __restrict__ y,int64_t groups2,int64_t nnz,double* out){
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  for(int64_t g=(int64_t)blockIdx.x*4+wave; g<groups2; g+=(int64_t)gridDim.x*4){
    const int64_t base=g*896; double t=0,t2=0;
#pragma unroll
    for(int h=0;h<2;++h){
    double v[8]; int c[8];
#pragma unroll
    for(int u=0;u<8;++u){ int64_t j=base+h*448+u*64+lane; j=j<nnz?j:nnz-1; c[u]=col[j]; v[u]=val[j]; }
#pragma unroll
    for(int u=0;u<8;++u) { if(h==0) t+=v[u]*(double)c[u]; else t2+=v[u]*(double)c[u]; }
    }
    *(double2*)&y[g*128+lane*2]=make_double2(t,t2);
  }
}
int main(){
  int64_t n=16777216, groups=n/64, nnz=groups*448; double *val,*out,*x,*y; int *col,*rp;
  CK(hipMalloc(&val,8*nnz)); CK(hipMalloc(&col,4*nnz)); CK(hipMalloc(&out,64)); CK(hipMalloc(&rp,4*(n+64))); CK(hipMalloc(&x,8*n)); CK(hipMalloc(&y,8*n));
  CK(hipMemset(val,0,8*nnz)); CK(hipMemset(col,0,4*nnz)); CK(hipMemset(rp,0,4*(n+64))); CK(hipMemset(x,0,8*n));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto run=[&](const char* name, auto kern, double bytes, int grid){
    for(int i=0;i<3;i++) kern<<<grid,256>>>(val,col,rp,x,y,groups,nnz,out);
    hipEventRecord(a); for(int i=0;i<20;i++) kern<<<grid,256>>>(val,col,rp,x,y,groups,nnz,out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms,a,b); ms/=20; CK(hipGetLastError());
    printf("grid %5d %-28s %.4f ms  actual %.0f GB/s\n",grid,name,ms,bytes/ms/1e6);
  };
  double m=12.0*nnz;
  for(int grid: {2048}){
  run("matrix only",k<0>,m,grid);
  run("+rp",k<1>,m+4.0*n,grid);
  run("+y store",k<2>,m+8.0*n,grid);
  run("+y store nt",k<6>,m+8.0*n,grid);
  run("+x read",k<8>,m+8.0*n,grid);
  run("+y store 16B x32 lanes",k<16>,m+8.0*n,grid);
  run("+y store 32B x16 lanes",k<32>,m+8.0*n,grid);
  { int g2=grid; for(int i=0;i<3;i++) k2<0><<<g2,256>>>(val,col,y,groups/2,nnz,out); hipEventRecord(a); for(int i=0;i<20;i++) k2<0><<<g2,256>>>(val,col,y,groups/2,nnz,out); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); ms/=20; printf("grid %5d %-28s %.4f ms  actual %.0f GB/s\n",grid,"2 groups/iter, 16B x64 store",ms,(m+8.0*n)/ms/1e6);}
  run("+rp +y",k<3>,m+12.0*n,grid);
  run("+rp +y +x",k<11>,m+20.0*n,grid);
  run("+rp +y(nt) +x",k<15>,m+20.0*n,grid);
  }
  return 0;
}
