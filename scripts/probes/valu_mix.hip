// Probe: cycles per wave64 VALU instruction on gfx950 by instruction kind, at 1 / 4 / 8 waves per SIMD and for independent vs
// dependent streams (round 5: is a wave64 fp32 instruction 2 or 4 cycles of its SIMD?  MI355X_MICROARCH.md says v_fma_f32
// = 2, the sweep kernels' ablations say ~4 per instruction of their mix).  Clock from s_memtime around the loop of wave 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
constexpr int ITERS = 2048;

template <int KIND, int ILP>
__global__ void k(float* out, unsigned long long* ticks, float seed) {
  float a[ILP];
  for (int i = 0; i < ILP; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
  const float c = seed * 0.999f, d = seed * 1e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (KIND == 0) a[i] = __builtin_fmaf(a[i], c, d);            // v_fma_f32
      if (KIND == 1) a[i] = a[i] * c;                              // v_mul_f32
      if (KIND == 2) a[i] = a[i] + d;                              // v_add_f32
      if (KIND == 3) a[i] = fabsf(a[i] - c) + d;                   // v_sub + v_add |.|
      if (KIND == 4) a[i] = __builtin_amdgcn_fmed3f(a[i], 0.01f, 1.0f) + d;   // v_med3 + v_add
      if (KIND == 5) a[i] = __builtin_amdgcn_exp2f(a[i]);          // v_exp_f32
      if (KIND == 6) a[i] = __builtin_amdgcn_rcpf(a[i]);           // v_rcp_f32
      if (KIND == 7) a[i] = fmaxf(a[i], c) + d;                    // v_max (+ canonicalise?) + v_add
      if (KIND == 8) { float t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(a[i])); a[i] = t; }   // v_mov_b32
      if (KIND == 9) a[i] = (a[i] > c) ? a[i] : d;                 // v_cmp + v_cndmask
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < ILP; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND, int ILP>
int run(const char* name, int ops) {
  float* out; unsigned long long* ticks; CK(hipMalloc(&out, 64)); CK(hipMalloc(&ticks, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 4, 8}) {
    const int blocks = 256, threads = 256 * wps;   // one block per CU, wps waves per SIMD
    k<KIND, ILP><<<blocks, threads>>>(out, ticks, 1.0f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<KIND, ILP><<<blocks, threads>>>(out, ticks, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    unsigned long long tk; CK(hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost));
    const double instr_per_simd = (double)wps * ITERS * ILP * ops;
    printf("%-28s ILP %d  %d waves/SIMD: %7.3f ms, %8llu ticks -> %.2f ticks per wave-instruction per SIMD (%.2f GHz if ticks are cycles)\n",
           name, ILP, wps, ms, tk, (double)tk / instr_per_simd, tk / (ms * 1e6));
  }
  return 0;
}
int main() {
  run<0, 8>("v_fma_f32", 1); run<0, 1>("v_fma_f32 dependent", 1);
  run<1, 8>("v_mul_f32", 1); run<2, 8>("v_add_f32", 1); run<2, 1>("v_add_f32 dependent", 1);
  run<3, 8>("v_sub + v_add|.|", 2); run<4, 8>("v_med3 + v_add", 2);
  run<5, 8>("v_exp_f32", 1); run<6, 8>("v_rcp_f32", 1); run<7, 8>("fmaxf + v_add", 2);
  run<8, 8>("v_mov_b32", 1); run<9, 8>("v_cmp + v_cndmask", 2);
  return 0;
}
