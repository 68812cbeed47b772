// Probe for the P-pixels-per-lane sweep kernels: HBM throughput of the traffic shape without arithmetic.
// Per workgroup (one image row) and plane: every lane loads P+1 consecutive floats at a 4-byte-aligned shifted offset
// from two tensors (one 16-byte buffer load + a 4/8-byte one) and stores P consecutive floats at another shifted offset
// to two tensors (ring-wrapped rows like the gather-form adjoint).  P = 1 is the round-1 kernels' shape (8-byte loads,
// 4-byte stores).  hipcc --offload-arch=gfx950 -O3 scripts/probes/wide_probe.hip -o scripts/probes/wide_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t Rsrc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ Rsrc rsrc(const float* p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ float ld1(Rsrc r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); }
__device__ __forceinline__ v2f ld2(Rsrc r, unsigned off) { return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }
typedef float v3f __attribute__((ext_vector_type(3)));
__device__ __forceinline__ v3f ld3(Rsrc r, unsigned off) { return __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, 0, 0)); }
__device__ __forceinline__ void st2(Rsrc r, unsigned off, v2f v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), r, (int)off, 0, 0); }
__device__ __forceinline__ v4f ld4(Rsrc r, unsigned off) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); }
__device__ __forceinline__ void st1(Rsrc r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0); }
__device__ __forceinline__ void st4(Rsrc r, unsigned off, v4f v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), r, (int)off, 0, 0); }

template <int P> struct Run { float v[P + 1]; };

template <int P> __device__ __forceinline__ Run<P> load_run(Rsrc r, unsigned off) {
  Run<P> o;
  if (P == 1) { const v2f a = ld2(r, off); o.v[0] = a.x; o.v[1] = a.y; }
  if (P == 2) { const v3f a = ld3(r, off); o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; }
  if (P == 3) { const v4f a = ld4(r, off); o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w; }
  if (P == 4) { const v4f a = ld4(r, off); o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w; o.v[4] = ld1(r, off + 16); }
  if (P == 5) { const v4f a = ld4(r, off); const v2f b = ld2(r, off + 16); o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w; o.v[4] = b.x; o.v[5] = b.y; }
  return o;
}
template <int P> __device__ __forceinline__ void store_run(Rsrc r, unsigned off, const float* v) {
  if (P == 1) st1(r, off, v[0]);
  if (P == 2) st2(r, off, v2f{v[0], v[1]});
  if (P == 3) { st1(r, off, v[0]); st1(r, off + 4, v[1]); st1(r, off + 8, v[2]); }
  if (P >= 4) st4(r, off, v4f{v[0], v[1], v[2], v[3]});
  if (P == 5) st1(r, off + 16, v[4]);
}

// MODE bit 0: loads, bit 1: stores.  ROWS target rows per workgroup pass (waves stride over segments of 64*P pixels).
template <int P, int U, int MODE>
__global__ __launch_bounds__(512) void pattern(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ GA,
                                               float* __restrict__ GB, float* __restrict__ out, int N, int H, int W) {
  extern __shared__ float dummy[];
  const int b = blockIdx.y, y = blockIdx.x;
  const long HW = (long)H * W;
  float acc = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int seg_px = 64 * P, nseg = (W + seg_px - 1) / seg_px;
  for (int seg = wave; seg < nseg; seg += nwaves) {
    const int x = seg * seg_px + lane * P;
    Run<P> ca[U], cb[U], na[U], nb[U];
    auto issue = [&](Run<P>* va, Run<P>* vb, int n0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = min(n0 + u, N - 1);
        const unsigned off = (unsigned)(x + ((MODE & 8) ? n * 8 : n * 6 + 1)) << 2;
        if (MODE & 1) {
          va[u] = load_run<P>(rsrc(A + ((long)b * N + n) * HW + (long)y * W, W * 4), off);
          vb[u] = load_run<P>(rsrc(Bt + ((long)b * N + n) * HW + (long)y * W, W * 4), off);
        } else {
#pragma unroll
          for (int i = 0; i <= P; ++i) { va[u].v[i] = 1.f + i; vb[u].v[i] = 3.f + i; }
        }
      }
    };
    issue(ca, cb, 0);
    for (int n0 = 0; n0 < N; n0 += U) {
      issue(na, nb, min(n0 + U, N - 1));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = min(n0 + u, N - 1);
        float oa[P], ob[P];
#pragma unroll
        for (int i = 0; i < P; ++i) { oa[i] = ca[u].v[i] * 1.0001f + ca[u].v[i + 1]; ob[i] = cb[u].v[i] + cb[u].v[i + 1]; acc += oa[i] + ob[i]; }
        if (MODE & 2) {
          int xs = x + ((MODE & 4) ? n * 8 : n * 6 + 1);
          xs = xs < W ? xs : xs - W;   // ring of W slots (a lane whose run wraps is rare; the probe lets the hardware clip it)
          const unsigned off = (x < W) ? (unsigned)xs << 2 : 0xFFFFFFF0u;
          store_run<P>(rsrc(GA + ((long)b * N + n) * HW + (long)y * W, W * 4), off, oa);
          store_run<P>(rsrc(GB + ((long)b * N + n) * HW + (long)y * W, W * 4), off, ob);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { ca[u] = na[u]; cb[u] = nb[u]; }
    }
  }
  if (acc == 123.456f) out[0] = acc + dummy[threadIdx.x];
}

__global__ void fill(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
  }
}

template <class F> static double time_ms(F f, int iters = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

static float *A, *Bt, *GA, *GB, *out;
template <int P, int U, int MODE> static void run(int waves, int wg_per_cu, int B, int N, int H, int W) {
  const size_t lds = (size_t)(160 * 1024) / wg_per_cu - 1024;
  CK(hipFuncSetAttribute((const void*)pattern<P, U, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double ms = time_ms([&] { pattern<P, U, MODE><<<dim3(H, B), waves * 64, lds>>>(A, Bt, GA, GB, out, N, H, W); });
  const double n = (double)B * N * H * W * 4;
  const double gb = (((MODE & 1) ? 2 : 0) + ((MODE & 2) ? 2 : 0)) * n / 1e9;
  printf("W=%4d P=%d U=%d waves=%d wg/cu=%d %s%s%s%s  %7.3f ms  %7.1f GB/s\n", W, P, U, waves, wg_per_cu, (MODE & 1) ? "L" : "-",
         (MODE & 2) ? "S" : "-", (MODE & 4) ? " alignedS" : "", (MODE & 8) ? " alignedL" : "", ms, gb / (ms * 1e-3));
}

int main() {
  const int B = 8, N = 49, H = 192;
  const size_t n = (size_t)B * N * H * 1280 / 2;   // 640-wide and (B/2) 1280-wide fit the same buffers
  CK(hipMalloc(&A, n * 4 + 4096)); CK(hipMalloc(&Bt, n * 4 + 4096)); CK(hipMalloc(&GA, n * 4 + 4096)); CK(hipMalloc(&GB, n * 4 + 4096));
  CK(hipMalloc(&out, 64));
  fill<<<4096, 256>>>(A, n, 1u); fill<<<4096, 256>>>(Bt, n, 7u);   // random data: zero-filled inputs flatter the clocks
  CK(hipDeviceSynchronize());
  const int W = 640;
  run<1, 2, 1>(4, 3, B, N, H, W); run<1, 2, 3>(4, 3, B, N, H, W);
  // 2 pixels per lane: 12-byte loads, 8-byte stores, 5 waves per row
  run<2, 2, 1>(5, 3, B, N, H, W); run<2, 4, 1>(5, 3, B, N, H, W); run<2, 2, 2>(5, 3, B, N, H, W); run<2, 2, 3>(5, 3, B, N, H, W);
  run<2, 4, 3>(5, 3, B, N, H, W); run<2, 2, 3>(5, 2, B, N, H, W); run<2, 2, 7>(5, 3, B, N, H, W);
  // 4 pixels per lane for reference
  run<4, 2, 1>(3, 3, B, N, H, W); run<4, 2, 3>(3, 3, B, N, H, W);
  return 0;
}
