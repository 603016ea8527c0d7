// Micro-benchmark: LDS-table counting throughput vs table size / workgroup shape / extras.
// (tools only, not shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)
__device__ __forceinline__ uint32_t fmix32(uint32_t h){h^=h>>16;h*=0x85EBCA6Bu;h^=h>>13;h*=0xC2B2AE35u;h^=h>>16;return h;}
__global__ void gen(int32_t* k, uint64_t n, double card, double s, uint32_t seed){
  uint64_t i=blockIdx.x*(uint64_t)blockDim.x+threadIdx.x; uint64_t st=(uint64_t)gridDim.x*blockDim.x;
  for(;i<n;i+=st){ uint32_t r=fmix32((uint32_t)i*2654435761u+seed); double u=(r+0.5)/4294967296.0;
    double x=pow((pow(card,1.0-s)-1.0)*u+1.0,1.0/(1.0-s)); int64_t v=(int64_t)floor(x); if(v<1)v=1; if(v>card)v=(int64_t)card;
    k[i]=(int32_t)((v*2654435761ull)%2147483648ull);} }
constexpr int EMPTY=INT32_MIN;
// EXTRA bit0: validity bitmap; bit1: lfill counter + check; bit2: filter 1/4 (split); bit3: U=4 software pipeline
template<int SLOTS,int BS,int EXTRA> __global__ __launch_bounds__(BS) void cnt(const int32_t* __restrict__ keys, const uint8_t* __restrict__ valid, uint64_t n, unsigned* out){
  __shared__ int lkeys[SLOTS]; __shared__ unsigned lcnt[SLOTS]; __shared__ unsigned lfill;
  for(int i=threadIdx.x;i<SLOTS;i+=BS){lkeys[i]=EMPTY;lcnt[i]=0;} if(threadIdx.x==0) lfill=0; __syncthreads();
  unsigned acc=0; const int4* vk=(const int4*)keys; uint64_t nv=n/4, st=(uint64_t)gridDim.x*BS;
  const unsigned q = blockIdx.x & 3;
  auto add=[&](int key){ uint32_t h=fmix32((uint32_t)key);
      if((EXTRA&4) && (h&3)!=q) return;
      h>>=17;
      for(int pr=0;pr<64;++pr){ uint32_t s=(h+pr)&(SLOTS-1); int cur=lkeys[s];
        if(cur==EMPTY){cur=atomicCAS(&lkeys[s],EMPTY,key); if(cur==EMPTY){cur=key; if(EXTRA&2) atomicAdd(&lfill,1u);} }
        if(cur==key){ atomicAdd(&lcnt[s],1u); break; } } };
  if(EXTRA&8){
    constexpr int U=4; int4 np[U]; unsigned nb[U];
    auto issue=[&](uint64_t v0){
      #pragma unroll
      for(int u=0;u<U;++u){ uint64_t v=v0+u*st; nb[u]=0x10000; if(v<nv){ np[u]=vk[v]; nb[u]=(EXTRA&1)? valid[(v*4)>>3]:0xFF; } } };
    issue(blockIdx.x*(uint64_t)BS+threadIdx.x);
    for(uint64_t v0=blockIdx.x*(uint64_t)BS+threadIdx.x; v0<nv; v0+=st*U){
      if((EXTRA&2) && lfill>SLOTS*3/4) break;
      int4 p[U]; unsigned b[U];
      #pragma unroll
      for(int u=0;u<U;++u){p[u]=np[u]; b[u]=nb[u];}
      issue(v0+st*U);
      #pragma unroll
      for(int u=0;u<U;++u){ if(b[u]&0x10000) continue; unsigned vb=(b[u]>>(((v0+u*st)*4)&7))&15; int k[4]={p[u].x,p[u].y,p[u].z,p[u].w};
        #pragma unroll
        for(int j=0;j<4;++j) if((vb>>j)&1) add(k[j]); else acc++; }
    }
  } else {
    for(uint64_t v=blockIdx.x*(uint64_t)BS+threadIdx.x; v<nv; v+=st){ int4 p=vk[v]; int k[4]={p.x,p.y,p.z,p.w};
      unsigned vb=15; if(EXTRA&1) vb=(valid[(v*4)>>3]>>((v*4)&7))&15;
      if((EXTRA&2) && lfill>SLOTS*3/4) break;
      #pragma unroll
      for(int j=0;j<4;++j) if((vb>>j)&1) add(k[j]); else acc++; }
  }
  __syncthreads();
  for(int i=threadIdx.x;i<SLOTS;i+=BS) acc+=lcnt[i];
  if(acc==0x12345678) out[0]=acc;
}
template<int SLOTS,int BS,int EXTRA> float run(int grid,const int32_t* k,const uint8_t* valid,uint64_t n,unsigned* out){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  cnt<SLOTS,BS,EXTRA><<<grid,BS>>>(k,valid,n,out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for(int r=0;r<3;r++) cnt<SLOTS,BS,EXTRA><<<grid,BS>>>(k,valid,n,out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/3*1000; }
int main(){ uint64_t n=45000000; int32_t* k; unsigned* out; uint8_t* valid; CK(hipMalloc(&k,n*4)); CK(hipMalloc(&out,64)); CK(hipMalloc(&valid,n/8+16)); CK(hipMemset(valid,0xFF,n/8+16));
  double cards[]={3,36,976,7420};
  printf("%6s | %8s %8s %8s %8s | %8s %8s %8s %8s %8s | %8s %8s\n","card","4k/256","8k/512","16k/1024","16k/512","+valid","+fill","+v+f","+v+f+U4","all/512","split4","split4U");
  for(double c: cards){ gen<<<2048,256>>>(k,n,c,1.1,(uint32_t)c); CK(hipDeviceSynchronize());
    float a0=run<4096,256,0>(1024,k,valid,n,out);
    float a1=run<8192,512,0>(512,k,valid,n,out);
    float a2=run<16384,1024,0>(256,k,valid,n,out);
    float a3=run<16384,512,0>(256,k,valid,n,out);
    float b0=run<16384,1024,1>(256,k,valid,n,out);
    float b1=run<16384,1024,2>(256,k,valid,n,out);
    float b2=run<16384,1024,3>(256,k,valid,n,out);
    float b3=run<16384,1024,11>(256,k,valid,n,out);
    float b4=run<16384,512,11>(256,k,valid,n,out);
    float d0=run<8192,1024,0>(512,k,valid,n,out);   // 2 WGs x 16 waves per CU
    float d1=run<8192,1024,11>(512,k,valid,n,out);
    float d2=run<4096,1024,0>(1024,k,valid,n,out);  // up to 4 WGs per CU (wave slots permitting)
    float c0=run<16384,1024,4>(256,k,valid,n,out);
    float c1=run<16384,1024,15>(256,k,valid,n,out);
    printf("%6.0f | %8.1f %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f %8.1f %8.1f | %8.1f %8.1f us || 8k/1024x2: %.1f  +v+f+U4: %.1f  4k/1024x4: %.1f\n",c,a0,a1,a2,a3,b0,b1,b2,b3,b4,c0,c1,d0,d1,d2); }
  return 0; }
