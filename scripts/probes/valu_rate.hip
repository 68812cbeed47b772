// Probe: issue rate of scalar vs packed fp32 VALU ops and of the transcendental / conversion ops the sweep kernels use.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
constexpr int ITERS = 4096, ILP = 8;

template <int KIND>
__global__ void k(float* out, float seed) {
  float a[ILP];
  v2f p[ILP];
  for (int i = 0; i < ILP; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = v2f{a[i], a[i] + 0.5f}; }
  const float c = seed * 0.999f, d = seed * 1e-3f;
  const v2f c2 = {c, c * 1.0001f}, d2 = {d, d * 0.5f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (KIND == 0) a[i] = __builtin_fmaf(a[i], c, d);                       // v_fma_f32
      if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], c2, d2);          // v_pk_fma_f32
      if (KIND == 2) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.5f;             // v_exp_f32 + mul
      if (KIND == 3) a[i] = __builtin_amdgcn_rcpf(a[i]) + 1.0f;              // v_rcp_f32 + add
      if (KIND == 4) a[i] = floorf(a[i] * c) + d;                            // v_floor + fma-ish
      if (KIND == 5) a[i] = (a[i] > c) ? a[i] * c : a[i] + d;                // cmp + cndmask + ...
      if (KIND == 6) a[i] = __builtin_amdgcn_fmed3f(a[i] * c, 0.01f, 1.0f) + d;  // mul + med3 + add
      if (KIND == 7) a[i] = (float)((int)(a[i] * c)) + d;                    // cvt i32 <-> f32
      if (KIND == 8) p[i] = p[i] * c2 + d2;                                   // v_pk_mul + v_pk_add (or pk_fma)
    }
  }
  float s = 0;
  for (int i = 0; i < ILP; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
int run(const char* name, int ops_per_iter) {
  float* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
  k<KIND><<<blocks, threads>>>(out, 1.0f); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) k<KIND><<<blocks, threads>>>(out, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double wave_instr = (double)blocks * (threads / 64) * ITERS * ILP * ops_per_iter;
  const double per_simd_per_clk = wave_instr / (ms * 1e-3) / 1024 / 2.4e9;
  printf("%-34s %7.3f ms  %6.1f G wave-instr/s  -> %.3f instr/clk/SIMD (=%.1f clk per instr @2.4GHz)\n", name, ms,
         wave_instr / (ms * 1e-3) / 1e9, per_simd_per_clk, 1.0 / per_simd_per_clk);
  return 0;
}
int main() {
  run<0>("v_fma_f32", 1);
  run<1>("v_pk_fma_f32", 1);
  run<8>("pk mul+add", 1);
  run<2>("v_exp_f32 + v_mul", 2);
  run<3>("v_rcp_f32 + v_add", 2);
  run<4>("v_mul + v_floor + v_add", 3);
  run<5>("cmp + 2 alu + cndmask", 4);
  run<6>("v_mul + v_med3 + v_add", 3);
  run<7>("mul + cvt_i32 + cvt_f32 + add", 4);
  return 0;
}
