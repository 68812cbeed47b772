// Probe: lane mapping of the wave-wide DPP shifts on gfx950 (0x130 wave_shl:1, 0x138 wave_shr:1) with a fallback value.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  const int v = 100 + lane;
  out[lane] = __builtin_amdgcn_update_dpp(-7, v, 0x130, 0xF, 0xF, false);
  out[64 + lane] = __builtin_amdgcn_update_dpp(-9, v, 0x138, 0xF, 0xF, false);
}
int main() {
  int *d, h[128];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("0x130: lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane62<-%d lane63<-%d\n", h[0], h[1], h[15], h[16], h[62], h[63]);
  printf("0x138: lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane62<-%d lane63<-%d\n", h[64], h[65], h[79], h[80], h[126], h[127]);
  return 0;
}
