// microbenchmark: do unaligned LDS / global dword accesses work on gfx950, and what do they cost?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;

template <typename T> __device__ __forceinline__ T ldu(const u8* p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }
template <typename T> __device__ __forceinline__ void stu(u8* p, T v) { __builtin_memcpy(p, &v, sizeof(T)); }

// mode 0: byte copy; 1: b32 unaligned read + b32 unaligned write; 2: b64/b64
// each lane copies LEN bytes from lds[src + lane*STRIDE + shift] to lds[dst + lane*STRIDE + shift2]
template <int MODE>
__global__ void lds_copy(u8* out, int iters, int stride, int shift_src, int shift_dst, u64* cycles) {
  __shared__ __attribute__((aligned(16))) u8 lds[32768];
  int lane = threadIdx.x;
  for (int i = lane; i < 32768; i += blockDim.x) lds[i] = (u8)(i * 7 + 3);
  __syncthreads();
  u8* s = lds + (lane & 63) * stride + shift_src + (lane >> 6) * 4096;
  u8* d = lds + 16384 + (lane & 63) * stride + shift_dst + (lane >> 6) * 4096;
  u64 t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { for (int k = 0; k < 8; k++) d[k] = s[k]; }
    if (MODE == 1) { stu<u32>(d, ldu<u32>(s)); stu<u32>(d + 4, ldu<u32>(s + 4)); }
    if (MODE == 2) { stu<u64>(d, ldu<u64>(s)); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  }
  u64 t1 = clock64();
  __syncthreads();
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
  if (blockIdx.x == 0) for (int i = lane; i < 32768; i += blockDim.x) out[i] = lds[i];
}

template <int MODE>
__global__ void glb_copy(const u8* in, u8* out, int iters, int stride, int shift_src, int shift_dst, u64* cycles) {
  int lane = threadIdx.x;
  const u8* s = in + (size_t)blockIdx.x * 65536 + lane * stride + shift_src;
  u8* d = out + (size_t)blockIdx.x * 65536 + lane * stride + shift_dst;
  u64 t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { for (int k = 0; k < 8; k++) d[k] = s[k]; }
    if (MODE == 1) { stu<u32>(d, ldu<u32>(s)); stu<u32>(d + 4, ldu<u32>(s + 4)); }
    if (MODE == 2) { stu<u64>(d, ldu<u64>(s)); }
    s += 1024; d += 1024;
  }
  __builtin_amdgcn_s_waitcnt(0);
  u64 t1 = clock64();
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  u8 *dout; u64* dcy; hipMalloc(&dout, 32768); hipMalloc(&dcy, 8 * 4096);
  std::vector<u8> h(32768); u64 cy[4096];
  int iters = 2000;
  for (int stride : {8, 13}) for (int ss : {0, 1, 3}) for (int sd : {0, 2, 3}) {
    printf("LDS stride=%2d shift_src=%d shift_dst=%d:", stride, ss, sd);
    for (int mode = 0; mode < 3; mode++) {
      hipMemset(dout, 0, 32768);
      if (mode == 0) lds_copy<0><<<1, 64>>>(dout, iters, stride, ss, sd, dcy);
      if (mode == 1) lds_copy<1><<<1, 64>>>(dout, iters, stride, ss, sd, dcy);
      if (mode == 2) lds_copy<2><<<1, 64>>>(dout, iters, stride, ss, sd, dcy);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), dout, 32768, hipMemcpyDeviceToHost); hipMemcpy(cy, dcy, 8, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; l++) for (int k = 0; k < 8; k++) {
        int si = l * stride + ss + k, di = 16384 + l * stride + sd + k;
        // expected value: original pattern at src (dst region may be overwritten by neighbours when stride<8: stride>=8 here)
        if (h[di] != (u8)(si * 7 + 3)) bad++;
      }
      printf("  mode%d %6.1f cyc/iter bad=%d", mode, (double)cy[0] / iters, bad);
    }
    printf("\n");
  }
  // global
  size_t N = 256 * 65536 * 4; u8 *gin, *gout; hipMalloc(&gin, N); hipMalloc(&gout, N);
  std::vector<u8> hin(N); for (size_t i = 0; i < N; i++) hin[i] = (u8)(i * 131 + 7);
  hipMemcpy(gin, hin.data(), N, hipMemcpyHostToDevice);
  std::vector<u8> hout(N);
  for (int stride : {8, 13}) for (int ss : {0, 1}) for (int sd : {0, 3}) {
    printf("GLB stride=%2d shift_src=%d shift_dst=%d:", stride, ss, sd);
    for (int mode = 0; mode < 3; mode++) {
      hipMemset(gout, 0, N);
      int it = 32;
      if (mode == 0) glb_copy<0><<<1024, 64>>>(gin, gout, it, stride, ss, sd, dcy);
      if (mode == 1) glb_copy<1><<<1024, 64>>>(gin, gout, it, stride, ss, sd, dcy);
      if (mode == 2) glb_copy<2><<<1024, 64>>>(gin, gout, it, stride, ss, sd, dcy);
      hipDeviceSynchronize();
      hipMemcpy(hout.data(), gout, N, hipMemcpyDeviceToHost); hipMemcpy(cy, dcy, 8 * 1024, hipMemcpyDeviceToHost);
      int bad = 0; double avg = 0;
      for (int b = 0; b < 1024; b++) { avg += cy[b];
        for (int i = 0; i < it; i++) for (int l = 0; l < 64; l++) for (int k = 0; k < 8; k++) {
          size_t si = (size_t)b * 65536 + i * 1024 + l * stride + ss + k, di = (size_t)b * 65536 + i * 1024 + l * stride + sd + k;
          if (hout[di] != hin[si]) bad++; } }
      printf("  mode%d %7.1f cyc/iter bad=%d", mode, avg / 1024 / it, bad);
    }
    printf("\n");
  }
  return 0;
}
