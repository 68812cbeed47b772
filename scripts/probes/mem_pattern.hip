// Probe: HBM throughput of the plane-sweep access pattern on gfx950, isolated from the arithmetic.
// Two tensors [B,N,H,W] fp32 are read the way the row-shift forward reads them (one workgroup per image row, a loop
// over N planes, each lane reading 8 bytes at a per-plane column shift), against variants that change one thing at a
// time.  Prints GB/s per variant.  Usage: mem_pattern [B N H W]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t Rsrc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ Rsrc rsrc(const float* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ v2f ld2(Rsrc r, unsigned off) { return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }
__device__ __forceinline__ v4f ld4(Rsrc r, unsigned off) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); }
__device__ __forceinline__ float ld1(Rsrc r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); }

// V0: the kernel's pattern.  grid (H*R? , B): one block per row, threads = W/CHUNKS, U planes per batch of loads.
template <int U, int WIDTH /*1,2*/, bool SHIFT>
__global__ void row_pattern(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ out, int N,
                            int H, int W, int rows_per_block) {
  const int b = blockIdx.y;
  const long HW = (long)H * W;
  float acc = 0.f;
  for (int rr = 0; rr < rows_per_block; ++rr) {
    const int y = blockIdx.x * rows_per_block + rr;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
      for (int n0 = 0; n0 < N; n0 += U) {
        float va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int n = min(n0 + u, N - 1);
          const int k = SHIFT ? (n * 6 + 1) : 0;  // a different column shift per plane
          const float* pa = A + ((long)b * N + n) * HW + (long)y * W;
          const float* pb = Bt + ((long)b * N + n) * HW + (long)y * W;
          const unsigned off = (unsigned)(x + k) << 2;
          if (WIDTH == 2) {
            const v2f a = ld2(rsrc(pa, W * 4), off), c = ld2(rsrc(pb, W * 4), off);
            va[u] = a.x + a.y; vb[u] = c.x + c.y;
          } else {
            va[u] = ld1(rsrc(pa, W * 4), off); vb[u] = ld1(rsrc(pb, W * 4), off);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += va[u] * 1.0001f + vb[u];
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

// V1: each lane owns 4 consecutive pixels of the row and reads 16 aligned bytes per plane and tensor.
template <int U>
__global__ void row_pattern_x4(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ out,
                               int N, int H, int W, int rows_per_block) {
  const int b = blockIdx.y;
  const long HW = (long)H * W;
  float acc = 0.f;
  for (int rr = 0; rr < rows_per_block; ++rr) {
    const int y = blockIdx.x * rows_per_block + rr;
    for (int x4 = threadIdx.x; x4 * 4 < W; x4 += blockDim.x) {
      for (int n0 = 0; n0 < N; n0 += U) {
        v4f va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int n = min(n0 + u, N - 1);
          const float* pa = A + ((long)b * N + n) * HW + (long)y * W;
          const float* pb = Bt + ((long)b * N + n) * HW + (long)y * W;
          va[u] = ld4(rsrc(pa, W * 4), (unsigned)x4 << 4);
          vb[u] = ld4(rsrc(pb, W * 4), (unsigned)x4 << 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += va[u].x + va[u].y + va[u].z + va[u].w + vb[u].x + vb[u].y + vb[u].z + vb[u].w;
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

// V4: plain streaming read of both tensors (float4 per lane, grid-stride).
__global__ void stream_read(const v4f* __restrict__ A, const v4f* __restrict__ Bt, float* __restrict__ out, long n4) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const v4f a = A[i], c = Bt[i];
    acc += a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <class F>
static double time_ms(F f, int iters = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  int B = 8, N = 49, H = 192, W = 640;
  if (argc >= 5) { B = atoi(argv[1]); N = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); }
  const size_t n = (size_t)B * N * H * W;
  float *A, *Bt, *out;
  CK(hipMalloc(&A, n * 4 + 4096)); CK(hipMalloc(&Bt, n * 4 + 4096)); CK(hipMalloc(&out, 64));
  CK(hipMemset(A, 0, n * 4)); CK(hipMemset(Bt, 0, n * 4));
  const double gb = 2.0 * n * 4 / 1e9;
  printf("B=%d N=%d H=%d W=%d  bytes read per pass = %.1f MB\n", B, N, H, W, gb * 1e3);
#define RUN(name, ...) do { double ms = time_ms([&] { __VA_ARGS__; }); printf("%-58s %8.3f ms  %7.1f GB/s\n", name, ms, gb / (ms * 1e-3)); } while (0)
  RUN("stream float4 (2048 blocks x 256)", (stream_read<<<2048, 256>>>((const v4f*)A, (const v4f*)Bt, out, (long)(n / 4))));
  RUN("row/block, 320 thr, b64 shifted, U=4", (row_pattern<4, 2, true><<<dim3(H, B), 320>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 320 thr, b64 shifted, U=8", (row_pattern<8, 2, true><<<dim3(H, B), 320>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 320 thr, b64 shifted, U=16", (row_pattern<16, 2, true><<<dim3(H, B), 320>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 640 thr, b64 shifted, U=8", (row_pattern<8, 2, true><<<dim3(H, B), 640>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 320 thr, b64 unshifted, U=8", (row_pattern<8, 2, false><<<dim3(H, B), 320>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 320 thr, b32 shifted, U=8", (row_pattern<8, 1, true><<<dim3(H, B), 320>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 320 thr, b32 unshifted, U=8", (row_pattern<8, 1, false><<<dim3(H, B), 320>>>(A, Bt, out, N, H, W, 1)));
  RUN("2 rows/block, 320 thr, b64 shifted, U=8", (row_pattern<8, 2, true><<<dim3(H / 2, B), 320>>>(A, Bt, out, N, H, W, 2)));
  RUN("4 rows/block, 320 thr, b64 shifted, U=8", (row_pattern<8, 2, true><<<dim3(H / 4, B), 320>>>(A, Bt, out, N, H, W, 4)));
  RUN("row/block, 160 thr x4 px, b128 aligned, U=4", (row_pattern_x4<4><<<dim3(H, B), 192>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 160 thr x4 px, b128 aligned, U=8", (row_pattern_x4<8><<<dim3(H, B), 192>>>(A, Bt, out, N, H, W, 1)));
  RUN("row/block, 160 thr x4 px, b128 aligned, U=16", (row_pattern_x4<16><<<dim3(H, B), 192>>>(A, Bt, out, N, H, W, 1)));
  RUN("2 rows/block, 320 thr x4 px (2 rows side by side), U=8", (row_pattern_x4<8><<<dim3(H, B), 64>>>(A, Bt, out, N, H, W, 1)));
  return 0;
}
