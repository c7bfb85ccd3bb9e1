#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u32 ld32u(const u8* p){ u32 v; __builtin_memcpy(&v,p,4); return v; }
// MODE 0: dependent unaligned dword load per step (stride from data: 4..11 bytes)
// MODE 1: same + a 2-byte store per step to a private output stream
// MODE 2: same as 0 but region data staged in LDS first (LDS dependent chain)
template<int MODE> __global__ void chase(const u8* buf, u32 region, int steps, u16* outp, u32* sink, u64* cyc){
  __shared__ u8 lds[64*512+16];
  u32 gl = blockIdx.x*64+threadIdx.x;
  const u8* p = buf + (size_t)gl*region;
  u16* o = outp + (size_t)gl*8192;
  u32 pos=0, acc=0;
  if (MODE==2){ for(int i=0;i<512;i++) lds[threadIdx.x*512+i]=p[i]; __syncthreads(); }
  u64 t0=clock64();
  for(int i=0;i<steps;i++){
    u32 w = (MODE==2) ? ld32u(lds + threadIdx.x*512 + (pos & 255)) : ld32u(p+pos);
    acc += w;
    if (MODE==1) o[i] = (u16)pos;
    pos += 4 + (w & 7);
    if (pos + 16 > region) pos = 0;
  }
  u64 t1=clock64();
  sink[gl]=acc; if(threadIdx.x==0) cyc[blockIdx.x]=t1-t0;
}
int main(){
  const u32 region=36*1024; const int nb=2048; size_t N=(size_t)nb*64*region;
  u8* d; hipMalloc(&d,N); std::vector<u8> h(N); u32 x=12345; for(size_t i=0;i<N;i++){ x=x*1664525u+1013904223u; h[i]=x>>24; }
  hipMemcpy(d,h.data(),N,hipMemcpyHostToDevice);
  u16* o; hipMalloc(&o,(size_t)nb*64*8192*2); u32* sink; hipMalloc(&sink,nb*64*4); u64* cyc; hipMalloc(&cyc,nb*8);
  std::vector<u64> hc(nb);
  int steps=4000;
  for(int blocks : {256, 1024, 2048}) for(int mode=0;mode<3;mode++){
    for(int rep=0;rep<2;rep++){
      if(mode==0) chase<0><<<blocks,64>>>(d,region,steps,o,sink,cyc);
      if(mode==1) chase<1><<<blocks,64>>>(d,region,steps,o,sink,cyc);
      if(mode==2) chase<2><<<blocks,64>>>(d,region,steps,o,sink,cyc);
      hipDeviceSynchronize();
    }
    hipMemcpy(hc.data(),cyc,blocks*8,hipMemcpyDeviceToHost); double a=0; for(int i=0;i<blocks;i++) a+=hc[i];
    printf("blocks=%4d (waves/CU=%.1f) mode=%d: %.0f cycles/step\n", blocks, blocks/256.0, mode, a/blocks/steps);
  }
  return 0;
}
