// Probe: HBM throughput of the plane-sweep access pattern as a function of the bytes in flight per CU.
// One 4-wave workgroup per image row (like the row kernels), the number of resident workgroups per CU capped by a dummy
// LDS allocation, each lane keeping U planes x 2 tensors x 8 bytes in flight (software-pipelined: the next batch is
// issued before the current one is consumed).  Answers: how much must be in flight to reach 4.5-5 TB/s?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t Rsrc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ Rsrc rsrc(const float* p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ v2f ld2(Rsrc r, unsigned off) { return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }

template <int U>
__global__ __launch_bounds__(256) void pattern(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ out,
                                               int N, int H, int W) {
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
        va[u] = ld2(rsrc(A + ((long)b * N + n) * HW + (long)y * W, W * 4), off);
        vb[u] = ld2(rsrc(Bt + ((long)b * N + n) * HW + (long)y * W, W * 4), off);
      }
    };
    issue(ca, cb, 0);
    for (int n0 = 0; n0 < N; n0 += U) {
      issue(na, nb, n0 + U);
#pragma unroll
      for (int u = 0; u < U; ++u) { acc += ca[u].x * 1.0001f + ca[u].y + cb[u].x + cb[u].y; }
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

template <int U> static void run(const float* A, const float* Bt, float* out, int B, int N, int H, int W, double gb) {
  for (int k : {2, 3, 4, 6, 8}) {
    const size_t lds = (size_t)(160 * 1024) / k - 1024;
    CK(hipFuncSetAttribute((const void*)pattern<U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const double ms = time_ms([&] { pattern<U><<<dim3(H, B), 256, lds>>>(A, Bt, out, N, H, W); });
    printf("U=%2d planes/batch, %d workgroups/CU (%2d waves/CU): in flight <= %4.0f KB/CU  %7.3f ms  %7.1f GB/s\n", U, k, 4 * k,
           2.0 * U * 2 * 8 * 64 * 4 * k / 1024.0, ms, gb / (ms * 1e-3));
  }
}

int main() {
  const int B = 8, N = 49, H = 192, W = 640;
  const size_t n = (size_t)B * N * H * W;
  float *A, *Bt, *out;
  CK(hipMalloc(&A, n * 4 + 4096)); CK(hipMalloc(&Bt, n * 4 + 4096)); CK(hipMalloc(&out, 64));
  CK(hipMemset(A, 0, n * 4)); CK(hipMemset(Bt, 0, n * 4));
  const double gb = 2.0 * n * 4 / 1e9;
  run<2>(A, Bt, out, B, N, H, W, gb);
  run<4>(A, Bt, out, B, N, H, W, gb);
  run<8>(A, Bt, out, B, N, H, W, gb);
  run<16>(A, Bt, out, B, N, H, W, gb);
  return 0;
}
