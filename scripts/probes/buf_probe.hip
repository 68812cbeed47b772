// Probe: raw buffer load semantics on gfx950 — per-dword range checking of dwordx2/x4 loads, unaligned (4-byte aligned)
// wide loads, and the behaviour of offsets near 2^32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const float* p, int W, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, W * 4, 0x00020000);
  int x0 = (int)threadIdx.x - 4;  // -4 .. W+3
  unsigned off = (unsigned)x0 << 2;
  v2f a = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
  v4f b = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
  float c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
  float* o = out + threadIdx.x * 8;
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = b.z; o[5] = b.w; o[6] = c; o[7] = (float)x0;
}
int main() {
  const int W = 10;
  float h[32];
  for (int i = 0; i < 32; ++i) h[i] = 100.0f + i;   // row occupies h[8..17]; neighbours are 100+ values too
  float *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 20 * 8 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 18>>>(d + 8, W, o);
  float r[20 * 8];
  hipMemcpy(r, o, 18 * 8 * 4, hipMemcpyDeviceToHost);
  for (int t = 0; t < 18; ++t)
    printf("x0=%3d  b64=(%5.0f,%5.0f)  b128=(%5.0f,%5.0f,%5.0f,%5.0f)  b32=%5.0f\n", (int)r[t*8+7], r[t*8], r[t*8+1], r[t*8+2], r[t*8+3], r[t*8+4], r[t*8+5], r[t*8+6]);
  return 0;
}
