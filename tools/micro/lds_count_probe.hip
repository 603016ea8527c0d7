// Micro-benchmark: where does the time go in LDS-table counting?  (tools only, not shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
__device__ __forceinline__ uint32_t fmix32(uint32_t h){h^=h>>16;h*=0x85EBCA6Bu;h^=h>>13;h*=0xC2B2AE35u;h^=h>>16;return h;}
__global__ void gen(int32_t* k, uint64_t n, double card, double s, uint32_t seed){
  uint64_t i=blockIdx.x*(uint64_t)blockDim.x+threadIdx.x; uint64_t st=(uint64_t)gridDim.x*blockDim.x;
  for(;i<n;i+=st){ uint32_t r=fmix32((uint32_t)i*2654435761u+seed); double u=(r+0.5)/4294967296.0;
    double x=pow((pow(card,1.0-s)-1.0)*u+1.0,1.0/(1.0-s)); int64_t v=(int64_t)floor(x); if(v<1)v=1; if(v>card)v=(int64_t)card;
    k[i]=(int32_t)((v*2654435761ull)%2147483648ull);} }
constexpr int SLOTS=4096; constexpr int EMPTY=INT32_MIN;
template<int MODE> __global__ __launch_bounds__(256) void cnt(const int32_t* __restrict__ keys, uint64_t n, unsigned* out){
  __shared__ int lkeys[SLOTS]; __shared__ unsigned lcnt[SLOTS];
  for(int i=threadIdx.x;i<SLOTS;i+=256){lkeys[i]=EMPTY;lcnt[i]=0;} __syncthreads();
  unsigned acc=0; const int4* vk=(const int4*)keys; uint64_t nv=n/4, st=(uint64_t)gridDim.x*256;
  for(uint64_t v=blockIdx.x*256ull+threadIdx.x; v<nv; v+=st){ int4 p=vk[v]; int k[4]={p.x,p.y,p.z,p.w};
    if(MODE==0){ acc+=p.x^p.y^p.z^p.w; }
    else if (MODE==4) {
      // wave sort + run-length: one LDS add per distinct key per wave
      #pragma unroll
      for(int j=0;j<4;++j){ int key=k[j];
        // bitonic sort across 64 lanes
        unsigned lane=__lane_id();
        #pragma unroll
        for(int size=2; size<=64; size<<=1){
          #pragma unroll
          for(int stride=size>>1; stride>0; stride>>=1){
            int other=__shfl_xor(key,stride,64);
            bool up=((lane&size)==0); bool lower=((lane&stride)==0);
            int mn=key<other?key:other, mx=key<other?other:key;
            key=(up==lower)?mn:mx; } }
        int prev=__shfl_up(key,1,64); bool head=(lane==0)||(prev!=key);
        unsigned long long hm=__ballot(head);
        if(head){ unsigned long long higher = hm & ~((2ull<<lane)-1ull); // heads above me
          int nxt = higher? __ffsll((long long)higher)-1 : 64; unsigned run=nxt-lane;
          uint32_t h=fmix32((uint32_t)key)>>17;
          for(int pr=0;pr<4;++pr){ uint32_t s=(h+pr)&(SLOTS-1); int cur=lkeys[s]; if(cur==EMPTY){cur=atomicCAS(&lkeys[s],EMPTY,key); if(cur==EMPTY)cur=key;} if(cur==key){atomicAdd(&lcnt[s],run);break;} } }
      }
    }
    else {
      #pragma unroll
      for(int j=0;j<4;++j){ int key=k[j]; uint32_t h=fmix32((uint32_t)key)>>17;
        for(int pr=0;pr<4;++pr){ uint32_t s=(h+pr)&(SLOTS-1); int cur=lkeys[s];
          if(cur==EMPTY){cur=atomicCAS(&lkeys[s],EMPTY,key); if(cur==EMPTY)cur=key;}
          if(cur==key){ if(MODE==1) acc+=s; else if(MODE==2) atomicAdd(&lcnt[s],1u); else if(MODE==3){ unsigned o=atomicAdd(&lcnt[s],1u); acc+=o;} break; } } }
    }
  }
  __syncthreads();
  for(int i=threadIdx.x;i<SLOTS;i+=256) acc+=lcnt[i];
  if(acc==0x12345678) out[0]=acc;
  if(threadIdx.x==0 && blockIdx.x==0){ unsigned t=0; for(int i=0;i<SLOTS;i++) t+=lcnt[i]; out[1]=t; }
}
template<int MODE> float run(const int32_t* k, uint64_t n, unsigned* out){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  cnt<MODE><<<1024,256>>>(k,n,out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for(int r=0;r<3;r++) cnt<MODE><<<1024,256>>>(k,n,out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/3*1000; }
int main(){ uint64_t n=45000000; int32_t* k; unsigned* out; CK(hipMalloc(&k,n*4)); CK(hipMalloc(&out,64));
  double cards[]={3,36,155,976,3000}; 
  printf("%8s %10s %10s %10s %10s %10s\n","card","read","probe","add","add_rtn","wavesort");
  for(double c: cards){ gen<<<2048,256>>>(k,n,c,1.1,(uint32_t)c); CK(hipDeviceSynchronize());
    float t0=run<0>(k,n,out), t1=run<1>(k,n,out), t2=run<2>(k,n,out), t3=run<3>(k,n,out), t4=run<4>(k,n,out);
    printf("%8.0f %10.1f %10.1f %10.1f %10.1f %10.1f us\n",c,t0,t1,t2,t3,t4); }
  return 0; }
