// Cost of one 134 MB write stream next to NR 134 MB read streams (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef double d2v __attribute__((ext_vector_type(2)));
template<int NR,int WR,bool NTL>
__global__ void __launch_bounds__(256) k(const double2* __restrict__ in,int64_t n2,double2* __restrict__ out,double* sink){
  double2 acc=make_double2(0,0); const int64_t stride=(int64_t)gridDim.x*256;
  for(int64_t i=(int64_t)blockIdx.x*256+threadIdx.x;i<n2;i+=stride){
    double2 t=make_double2(0,0);
#pragma unroll
    for(int r=0;r<NR;++r){ double2 v; if(NTL){ d2v q=__builtin_nontemporal_load((const d2v*)&in[(int64_t)r*n2+i]); v=make_double2(q.x,q.y);} else v=in[(int64_t)r*n2+i]; t.x+=v.x; t.y+=v.y; }
    if(WR==1) out[i]=t; else if(WR==2){ d2v q; q.x=t.x; q.y=t.y; __builtin_nontemporal_store(q,(d2v*)&out[i]); } else {acc.x+=t.x; acc.y+=t.y;}
  }
  if(acc.x==1.234) sink[0]=acc.y;
}
int main(){
  int64_t n=16777216, n2=n/2; double2 *in,*out; double* sink;
  CK(hipMalloc(&in,16*n2*12)); CK(hipMalloc(&out,16*n2)); CK(hipMalloc(&sink,64)); CK(hipMemset(in,0,16*n2*12));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto run=[&](const char* name, auto kern, double bytes){
    for(int i=0;i<3;i++) kern<<<2048,256>>>(in,n2,out,sink);
    hipEventRecord(a); for(int i=0;i<20;i++) kern<<<2048,256>>>(in,n2,out,sink); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms,a,b); ms/=20; CK(hipGetLastError());
    printf("%-18s %.4f ms  %.0f GB/s\n",name,ms,bytes/ms/1e6);
  };
  double v=8.0*n;
  run("read12",k<12,0,false>,12*v); run("read12 ntl",k<12,0,true>,12*v);
  run("read12+write",k<12,1,false>,13*v); run("read12+write nts",k<12,2,false>,13*v);
  run("read12 ntl+write",k<12,1,true>,13*v); run("read12 ntl+write nts",k<12,2,true>,13*v);
  run("read4+write",k<4,1,false>,5*v); run("read4 ntl+write nts",k<4,2,true>,5*v);
  run("read2+write",k<2,1,false>,3*v); run("read2 ntl+write nts",k<2,2,true>,3*v);
  run("read1+write",k<1,1,false>,2*v); run("read1 ntl+write nts",k<1,2,true>,2*v);
  return 0;
}
