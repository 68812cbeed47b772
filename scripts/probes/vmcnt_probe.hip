// Probe: do vector-memory STORES take part in vmcnt on gfx950?  One lane issues a cold load (HBM latency), then a store,
// then waits with vmcnt(1) and reads the clock.  If stores count, vmcnt(1) lets the (younger) store stay outstanding but
// has to wait for the load (~microseconds); if they do not, one outstanding load is allowed and the wait is free.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* __restrict__ cold, float* __restrict__ sink, long long* t, float* outv) {
  const long long t0 = __builtin_readcyclecounter();
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(cold + threadIdx.x * 1024) : "memory");
  asm volatile("global_store_dword %0, %1, off" :: "v"(sink + threadIdx.x), "v"(1.0f) : "memory");
  asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_readcyclecounter();
  outv[threadIdx.x] = v;
  if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t0; }
}
int main() {
  float *cold, *sink, *outv; long long *t, h[2];
  hipMalloc(&cold, 64 << 20); hipMalloc(&sink, 4096); hipMalloc(&outv, 4096); hipMalloc(&t, 16);
  hipMemset(cold, 0, 64 << 20);
  for (int rep = 0; rep < 3; ++rep) {
    k<<<1, 64>>>(cold + rep * (4 << 20), sink, t, outv);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("after vmcnt(1): %lld cycles, after vmcnt(0): %lld cycles\n", h[0], h[1]);
  }
  return 0;
}
