// Probe: HBM throughput of the BACKWARD's traffic shape without its arithmetic: per workgroup (one image row, 4 waves)
// and plane, 8-byte shifted loads from two tensors and 4-byte shifted (ring-wrapped) stores to two other tensors, two
// planes per group, one group prefetched.  Compare with the load-only pattern (inflight_probe) and with the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t Rsrc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ Rsrc rsrc(const float* p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ v2f ld2(Rsrc r, unsigned off) { return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }
__device__ __forceinline__ void st1(Rsrc r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0); }

template <int U, bool LOAD, bool STORE>
__global__ __launch_bounds__(256) void pattern(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ GA,
                                               float* __restrict__ GB, float* __restrict__ out, int N, int H, int W) {
  extern __shared__ float dummy[];
  const int b = blockIdx.y, y = blockIdx.x;
  const long HW = (long)H * W;
  float acc = 0.f;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    v2f ca[U], cb[U], na[U], nb[U];
    auto issue = [&](v2f* va, v2f* vb, int n0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = min(n0 + u, N - 1);
        const unsigned off = (unsigned)(x + n * 6 + 1) << 2;
        if (LOAD) {
          va[u] = ld2(rsrc(A + ((long)b * N + n) * HW + (long)y * W, W * 4), off);
          vb[u] = ld2(rsrc(Bt + ((long)b * N + n) * HW + (long)y * W, W * 4), off);
        } else { va[u] = v2f{1.f, 2.f}; vb[u] = v2f{3.f, 4.f}; }
      }
    };
    issue(ca, cb, 0);
    for (int n0 = 0; n0 < N; n0 += U) {
      issue(na, nb, min(n0 + U, N - 1));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = min(n0 + u, N - 1);
        const float va = ca[u].x * 1.0001f + ca[u].y, vb = cb[u].x + cb[u].y;
        acc += va + vb;
        if (STORE) {
          unsigned xs = (unsigned)(x + n * 6 + 1);
          xs = xs < (unsigned)W ? xs : xs - W;   // ring of W slots, as the gather-form adjoint writes them
          st1(rsrc(GA + ((long)b * N + n) * HW + (long)y * W, W * 4), xs << 2, va);
          st1(rsrc(GB + ((long)b * N + n) * HW + (long)y * W, W * 4), xs << 2, vb);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { ca[u] = na[u]; cb[u] = nb[u]; }
    }
  }
  if (acc == 123.456f) out[0] = acc + dummy[threadIdx.x];
}

template <class F> static double time_ms(F f, int iters = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

template <int U, bool LOAD, bool STORE> static void run(const char* what, const float* A, const float* Bt, float* GA, float* GB,
                                                        float* out, int B, int N, int H, int W, double gb) {
  const size_t lds = (size_t)(160 * 1024) / 3 - 1024;  // 3 workgroups per CU, like the kernels
  CK(hipFuncSetAttribute((const void*)pattern<U, LOAD, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double ms = time_ms([&] { pattern<U, LOAD, STORE><<<dim3(H, B), 256, lds>>>(A, Bt, GA, GB, out, N, H, W); });
  printf("%-28s U=%d  %7.3f ms  %7.1f GB/s\n", what, U, ms, gb / (ms * 1e-3));
}

int main() {
  const int B = 8, N = 49, H = 192, W = 640;
  const size_t n = (size_t)B * N * H * W;
  float *A, *Bt, *GA, *GB, *out;
  CK(hipMalloc(&A, n * 4 + 4096)); CK(hipMalloc(&Bt, n * 4 + 4096)); CK(hipMalloc(&GA, n * 4 + 4096)); CK(hipMalloc(&GB, n * 4 + 4096));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(A, 0, n * 4)); CK(hipMemset(Bt, 0, n * 4));
  const double gb2 = 2.0 * n * 4 / 1e9;
  run<2, true, false>("loads only (2 tensors)", A, Bt, GA, GB, out, B, N, H, W, gb2);
  run<2, false, true>("stores only (2 tensors)", A, Bt, GA, GB, out, B, N, H, W, gb2);
  run<2, true, true>("loads + stores (4 tensors)", A, Bt, GA, GB, out, B, N, H, W, 2 * gb2);
  run<4, true, true>("loads + stores (4 tensors)", A, Bt, GA, GB, out, B, N, H, W, 2 * gb2);
  return 0;
}
