#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <string.h>
static double now(){ struct timespec ts; clock_gettime(CLOCK_MONOTONIC,&ts); return ts.tv_sec+ts.tv_nsec*1e-9; }
int main(){
  hipFree(0);
  for (size_t mb : {64, 256, 1024}) {
    void* p; double t0=now(); hipHostMalloc(&p, mb<<20, hipHostMallocNonCoherent); double t1=now();
    memset(p, 1, mb<<20); double t2=now();
    void* d; hipMalloc(&d, mb<<20); double t3=now();
    hipMemcpy(d,p,mb<<20,hipMemcpyHostToDevice); double t4=now();
    hipMemcpy(d,p,mb<<20,hipMemcpyHostToDevice); double t5=now();
    hipMemcpy(p,d,mb<<20,hipMemcpyDeviceToHost); double t6=now();
    hipHostFree(p); double t7=now(); hipFree(d);
    void* q = malloc(mb<<20); memset(q,2,mb<<20); double t8=now(); void* r = malloc(mb<<20); memset(r,3,mb<<20); double t9=now(); memcpy(r,q,mb<<20); double t10=now();
    printf("%4zu MiB: hostmalloc %.1f ms, first-touch memset %.1f ms, hipMalloc %.1f ms, H2D %.1f/%.1f ms (%.1f GB/s), D2H %.1f ms (%.1f GB/s), hostfree %.1f ms, memcpy host %.1f ms (%.1f GB/s)\n", mb,
      (t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t5-t4)*1e3,(mb/1024.0)/(t5-t4),(t6-t5)*1e3,(mb/1024.0)/(t6-t5),(t7-t6)*1e3,(t10-t9)*1e3,(mb/1024.0)/(t10-t9));
    free(q); free(r);
  }
  return 0;
}
