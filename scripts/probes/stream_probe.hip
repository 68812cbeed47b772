// Probe for the source-ordered ("row-stream") backward: one wave walks a whole source row of one plane segment by
// segment (no hand-over between waves), lanes own ALIGNED source slots, the per-target-pixel context comes out of LDS
// at the plane's shift.  Measures what the traffic shape + LDS reads + K dummy VALU instructions per pixel and plane
// cost, before the real kernel is written.
//   P  = slots per lane (1: dword stores, 8-byte loads; 2: 8-byte stores, 12-byte loads; 4: 16-byte stores, 16+4-byte loads)
//   D  = global-load prefetch depth in (plane, segment) iterations; the LDS reads run one iteration ahead
//   K  = dummy VALU instructions per pixel and plane
// hipcc --offload-arch=gfx950 -O3 scripts/probes/stream_probe.hip -o scripts/probes/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t Rsrc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ Rsrc rsrc(const float* p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ float ld1(Rsrc r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); }
__device__ __forceinline__ v2f ld2(Rsrc r, unsigned off) { return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }
__device__ __forceinline__ v3f ld3(Rsrc r, unsigned off) { return __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, (int)off, 0, 0)); }
__device__ __forceinline__ v4f ld4(Rsrc r, unsigned off) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); }
__device__ __forceinline__ void st1(Rsrc r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0); }
__device__ __forceinline__ void st2(Rsrc r, unsigned off, v2f v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)off, 0, 0); }
__device__ __forceinline__ void st4(Rsrc r, unsigned off, v4f v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, (int)off, 0, 0); }

template <int P> struct Glb { float a[P + 1], b[P + 1]; };
template <int P> struct Lds { v4f c0[P], c1[P]; v2f c2[P]; v4f col[P + 1]; };

template <int P> __device__ __forceinline__ void load_run(Rsrc r, unsigned off, float* v) {
  if (P == 1) { const v2f a = ld2(r, off); v[0] = a.x; v[1] = a.y; }
  if (P == 2) { const v3f a = ld3(r, off); v[0] = a.x; v[1] = a.y; v[2] = a.z; }
  if (P == 4) { const v4f a = ld4(r, off); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = ld1(r, off + 16); }
}
template <int P> __device__ __forceinline__ void store_run(Rsrc r, unsigned off, const float* v) {
  if (P == 1) st1(r, off, v[0]);
  if (P == 2) st2(r, off, v2f{v[0], v[1]});
  if (P == 4) st4(r, off, v4f{v[0], v[1], v[2], v[3]});
}

// K VALU instructions on the group's data (4 independent chains)
template <int K> __device__ __forceinline__ float burn(float x0, float x1, float x2, float x3) {
#pragma unroll
  for (int i = 0; i < K / 4; ++i) { x0 = fmaf(x0, 1.0001f, x1); x1 = fmaf(x1, 0.9999f, x2); x2 = fmaf(x2, 1.0002f, x3); x3 = fmaf(x3, 0.9998f, x0); }
  return x0 + x1 + x2 + x3;
}

template <int P, int D, int K, int MODE, int WAVES, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void stream(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ GA,
                                                 float* __restrict__ GB, const float* __restrict__ ctx_src, const int* __restrict__ kshift,
                                                 float* __restrict__ out, int N, int H, int W, int Bn) {
  extern __shared__ v4f lds[];
  const int RS = W + 4;                 // ctx rows with guard cells
  v4f* c0 = lds; v4f* c1 = lds + RS; v4f* col = lds + 2 * RS; v2f* c2 = reinterpret_cast<v2f*>(lds + 3 * RS);
  // MODE & 8 (VERDICT r3 #1c): a softmax / mixture state of six floats per TARGET cell and wave, kept in LDS and updated
  // by read-modify-write at the shifted cell while the wave walks plane rows in SOURCE order (merged once per row in a real kernel)
  v4f* st0 = lds + 4 * RS; v2f* st1 = reinterpret_cast<v2f*>(lds + 4 * RS + WAVES * RS);
  if (MODE & 8) {
    for (int x = threadIdx.x; x < WAVES * RS; x += blockDim.x) { st0[x] = v4f{0, 0, 0, 0}; st1[x] = v2f{0, 0}; }
  }
  const int id = blockIdx.x;            // row-major over images: rows of all images, image fastest
  const int b = id % Bn, y = id / Bn;
  const long HW = (long)H * W;
  // stage: 13 floats per pixel from global (target, stash, rgb_rec, g_rgb in the real kernel)
  for (int x = threadIdx.x; x < RS; x += blockDim.x) {
    const int xi = x - 2;
    v4f a = {0, 0, 0, 1e30f}, g = {0, 0, 0, 0}, cc = {0, 0, 0, 0}; v2f h = {0, 0};
    if (xi >= 0 && xi < W) {
      const float* p = ctx_src + ((long)b * 13) * HW + (long)y * W + xi;
      a = v4f{p[0], p[HW], p[2 * HW], p[3 * HW]}; g = v4f{p[4 * HW], p[5 * HW], p[6 * HW], p[7 * HW]};
      h = v2f{p[8 * HW], p[9 * HW]}; cc = v4f{p[10 * HW], p[11 * HW], p[12 * HW], 0};
    }
    c0[x] = a; c1[x] = g; c2[x] = h; col[x] = cc;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = WAVES;
  const int SEG = 64 * P, nseg = (W + SEG - 1) / SEG;
  // this wave's share of the (plane, segment) list: a contiguous range, balanced to one item
  const int items = N * nseg;
  const int i0 = __builtin_amdgcn_readfirstlane(items * wave / nwaves), i1 = __builtin_amdgcn_readfirstlane(items * (wave + 1) / nwaves);
  float acc = 0.f;
  const float* Ab = A + (long)b * N * HW + (long)y * W; const float* Bb = Bt + (long)b * N * HW + (long)y * W;
  float* GAb = GA + (long)b * N * HW + (long)y * W; float* GBb = GB + (long)b * N * HW + (long)y * W;
  auto issue_g = [&](Glb<P>& g, int it_raw) {
    const int it = min(it_raw, i1 - 1);
    const int n = it / nseg, seg = it - n * nseg;
    const int xs = seg * SEG + lane * P;
    if (MODE & 1) {
      load_run<P>(rsrc(Ab + (unsigned)(n * (int)HW), W * 4), xs * 4, g.a);
      load_run<P>(rsrc(Bb + (unsigned)(n * (int)HW), W * 4), xs * 4, g.b);
    } else {
#pragma unroll
      for (int i = 0; i <= P; ++i) { g.a[i] = 1.f + i; g.b[i] = 2.f + i; }
    }
  };
  auto issue_l = [&](Lds<P>& g, int it_raw) {
    const int it = min(it_raw, i1 - 1);
    const int n = it / nseg, seg = it - n * nseg;
    const int xs = seg * SEG + lane * P;
    const int k = __builtin_amdgcn_readfirstlane(kshift[b * N + n]);
    if (MODE & 4) {
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int xt = min(max(xs + i - k, -1), W) + 2;
        g.c0[i] = c0[xt]; g.c1[i] = c1[xt]; g.c2[i] = c2[xt];
      }
#pragma unroll
      for (int i = 0; i <= P; ++i) g.col[i] = col[min(xs + i, W) + 2];
    } else {
#pragma unroll
      for (int i = 0; i < P; ++i) { g.c0[i] = v4f{1, 2, 3, 4}; g.c1[i] = v4f{1, 2, 3, 4}; g.c2[i] = v2f{1, 2}; }
#pragma unroll
      for (int i = 0; i <= P; ++i) g.col[i] = v4f{1, 2, 3, 4};
    }
  };
  float carry_a = 0.f, carry_b = 0.f;
  auto compute = [&](const Glb<P>& g, const Lds<P>& c, int it) {
    const int n = it / nseg, seg = it - n * nseg;
    const int xs = seg * SEG + lane * P;
    if (seg == 0) carry_a = carry_b = 0.f;
    float oa[P], ob[P], la = 0.f, lb = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float l = g.a[i] * 0.25f + g.a[i + 1] * 0.75f, s = g.b[i] * 0.25f + g.b[i + 1] * 0.75f;
      const float cr = c.col[i].x * 0.25f + c.col[i + 1].x * 0.75f, cg = c.col[i].y * 0.25f + c.col[i + 1].y * 0.75f;
      const float r = burn<K>(l + c.c0[i].x + c.c1[i].x, s + c.c0[i].y + c.c1[i].y + c.c2[i].x, cr + c.c0[i].z + c.c1[i].z, cg + c.c0[i].w + c.c1[i].w + c.c2[i].y);
      const float c0v = r * 0.25f, c1v = r * 0.75f, d0v = r * 0.5f, d1v = r * 0.125f;
      if (i == 0) { oa[0] = c0v; ob[0] = d0v; } else { oa[i] = c0v + la; ob[i] = d0v + lb; }
      la = c1v; lb = d1v;
      acc += r;
    }
    if (MODE & 8) {
      const int k = __builtin_amdgcn_readfirstlane(kshift[b * N + n]);
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int xt = min(max(xs + i - k, -1), W) + 2 + wave * RS;
        v4f a0 = st0[xt]; v2f a1 = st1[xt];
        a0 += v4f{oa[i], ob[i], la, lb}; a1 += v2f{oa[i] * 0.5f, ob[i] * 0.5f};
        st0[xt] = a0; st1[xt] = a1;
      }
    }
    // lane's first slot also receives the previous lane's last right-hand contribution (lane 0: the carried one)
    const float pa = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry_a), __float_as_int(la), 0x138, 0xF, 0xF, false));
    const float pb = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry_b), __float_as_int(lb), 0x138, 0xF, 0xF, false));
    oa[0] += pa; ob[0] += pb;
    carry_a = __builtin_amdgcn_readlane(la, 63); carry_b = __builtin_amdgcn_readlane(lb, 63);
    if (MODE & 2) {
      store_run<P>(rsrc(GAb + (unsigned)(n * (int)HW), W * 4), xs * 4, oa);
      store_run<P>(rsrc(GBb + (unsigned)(n * (int)HW), W * 4), xs * 4, ob);
    }
  };
  // software pipeline: global loads D iterations ahead (ring of D+1 register groups, statically indexed), LDS reads 1 ahead
  Glb<P> g[D + 1];
  Lds<P> c[2];
#pragma unroll
  for (int j = 0; j < D; ++j) issue_g(g[j], i0 + j);
  issue_l(c[0], i0);
  int it = i0;
  constexpr int UN = (D + 1) % 2 ? 2 * (D + 1) : (D + 1);   // unroll so that both rings are statically indexed
  for (; it + UN <= i1; it += UN) {
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      issue_g(g[(j + D) % (D + 1)], it + j + D);
      issue_l(c[(j + 1) & 1], it + j + 1);
      compute(g[j % (D + 1)], c[j & 1], it + j);
    }
  }
  // tail (fewer than UN iterations): same schedule with guards
#pragma unroll
  for (int j = 0; j < UN; ++j) {
    if (it + j < i1) {
      issue_g(g[(j + D) % (D + 1)], it + j + D);
      issue_l(c[(j + 1) & 1], it + j + 1);
      compute(g[j % (D + 1)], c[j & 1], it + j);
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

__global__ void fill(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
  }
}
__global__ void fill_k(int* k, int n, int N, int W, int even) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { int v = (int)(300.0f * (W / 640.0f) * powf(2.0f / 300.0f, (float)(i % N) / (N - 1))); k[i] = even ? (v & ~1) : v; } }

template <class F> static double time_ms(F f, int iters = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

static float *A, *Bt, *GA, *GB, *out, *ctx; static int* ks;
template <int P, int D, int K, int MODE, int WAVES = 4, int OCC = 4> static void run(int B, int N, int H, int W) {
  const size_t ldsb = (size_t)(W + 4) * (3 * 16 + 8) + ((MODE & 8) ? (size_t)(W + 4) * 8 + (size_t)WAVES * (W + 4) * 24 : 0);
  if (ldsb > 160 * 1024) { printf("W=%4d P=%d D=%d K=%3d waves=%d: %zu bytes of LDS do not fit a CU — skipped\n", W, P, D, K, WAVES, ldsb); return; }
  CK(hipFuncSetAttribute((const void*)stream<P, D, K, MODE, WAVES, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  const double ms = time_ms([&] { stream<P, D, K, MODE, WAVES, OCC><<<dim3(H * B), WAVES * 64, ldsb>>>(A, Bt, GA, GB, ctx, ks, out, N, H, W, B); });
  CK(hipGetLastError());
  const double bytes = (double)B * H * W * 4 * (((MODE & 1) ? 2 * N : 0) + ((MODE & 2) ? 2 * N : 0) + 13);
  printf("W=%4d P=%d D=%d K=%3d waves=%d occ=%d %s%s%s%s lds=%zu  %7.3f ms  %7.1f GB/s\n", W, P, D, K, WAVES, OCC, (MODE & 1) ? "L" : "-", (MODE & 2) ? "S" : "-",
         (MODE & 4) ? "C" : "-", (MODE & 8) ? "R" : "-", ldsb, ms, bytes / 1e9 / (ms * 1e-3));
}


// ---- forward shape: TARGET-ordered (the softmax state lives with the target pixel), two target pixels per lane, the
// taps of plane n at xt + k_n: one 12-byte load per tensor at 4-byte alignment, colour taps out of LDS at the same
// shift, K VALU per pixel-plane.  A wave owns SEGS consecutive 128-pixel segments and a share of the planes; per plane it
// walks its segments in order (SEGS = 1: one segment, hopping from plane to plane).
template <int D, int K, int WAVES, int OCC, int SEGS>
__global__ __launch_bounds__(WAVES * 64, OCC) void fwdstream(const float* __restrict__ A, const float* __restrict__ Bt,
                                                             const float* __restrict__ ctx_src, const int* __restrict__ kshift,
                                                             float* __restrict__ outp, int N, int H, int W, int Bn) {
  extern __shared__ v4f lds[];
  const int RS = W + 8;
  v4f* col = lds;
  float* park = reinterpret_cast<float*>(lds + RS);
  const int id = blockIdx.x, b = id % Bn, y = id / Bn;
  const long HW = (long)H * W;
  for (int x = threadIdx.x; x < RS; x += blockDim.x) {
    const int xi = x - 4;
    v4f cc = {0, 0, 0, 0};
    if (xi >= 0 && xi < W) { const float* p = ctx_src + ((long)b * 13) * HW + (long)y * W + xi; cc = v4f{p[0], p[HW], p[2 * HW], 0}; }
    col[x] = cc;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (W + 127) / 128, ngrp = nseg / SEGS, split = WAVES / ngrp;   // waves per segment group
  const int grp = wave / split, part = wave - grp * split;
  const int n0 = N * part / split, n1 = N * (part + 1) / split;
  const int xt0 = grp * SEGS * 128 + lane * 2;
  const float* Ab = A + (long)b * N * HW + (long)y * W; const float* Bb = Bt + (long)b * N * HW + (long)y * W;
  float t[SEGS][6], acc[SEGS][16];
#pragma unroll
  for (int j = 0; j < SEGS; ++j) {
    const float* tp = ctx_src + ((long)b * 13 + 3) * HW + (long)y * W + min(xt0 + j * 128, W - 2);
    t[j][0] = tp[0]; t[j][1] = tp[1]; t[j][2] = tp[HW]; t[j][3] = tp[HW + 1]; t[j][4] = tp[2 * HW]; t[j][5] = tp[2 * HW + 1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  }
  struct G { float a[3], b[3]; };
  const int items = (n1 - n0) * SEGS;      // (plane, segment) pairs, segment innermost
  auto issue = [&](G& g, int it_raw) {
    const int it = min(it_raw, items - 1), n = n0 + it / SEGS, j = it % SEGS;
    const int k = __builtin_amdgcn_readfirstlane(kshift[b * N + n]);
    const unsigned off = (unsigned)(xt0 + j * 128 + k) * 4;
    const v3f va = ld3(rsrc(Ab + (unsigned)(n * (int)HW), W * 4), off), vb = ld3(rsrc(Bb + (unsigned)(n * (int)HW), W * 4), off);
    g.a[0] = va.x; g.a[1] = va.y; g.a[2] = va.z; g.b[0] = vb.x; g.b[1] = vb.y; g.b[2] = vb.z;
  };
  auto compute = [&](const G& g, int n, int j) {
    const int k = __builtin_amdgcn_readfirstlane(kshift[b * N + n]);
    const v4f* cp = col + min(max(xt0 + j * 128 + k, -4), W + 1) + 4;
    const v4f c0 = cp[0], c1 = cp[1], c2 = cp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float l = g.a[i] * 0.25f + g.a[i + 1] * 0.75f, s = g.b[i] * 0.25f + g.b[i + 1] * 0.75f;
      const v4f ca = i ? c1 : c0, cb = i ? c2 : c1;
      const float cr = ca.x * 0.25f + cb.x * 0.75f, cg = ca.y * 0.25f + cb.y * 0.75f, cbl = ca.z * 0.25f + cb.z * 0.75f;
      const float r = burn<K>(l + t[j][i], s + t[j][2 + i], cr + t[j][4 + i], cg + cbl);
      acc[j][i * 8 + 0] += r; acc[j][i * 8 + 1] += l; acc[j][i * 8 + 2] += s * r; acc[j][i * 8 + 3] += cr * r;
      acc[j][i * 8 + 4] += cg * r; acc[j][i * 8 + 5] += cbl * r; acc[j][i * 8 + 6] += r * l; acc[j][i * 8 + 7] += r * s;
    }
  };
  // software pipeline over the item list; the unroll factor is a multiple of SEGS so that the segment index is static
  constexpr int UN = (D + 1) * SEGS;
  G g[D + 1];
#pragma unroll
  for (int q = 0; q < D; ++q) issue(g[q], q);
  int it = 0;
  for (; it + UN <= items; it += UN) {
#pragma unroll
    for (int q = 0; q < UN; ++q) { issue(g[(q + D) % (D + 1)], it + q + D); compute(g[q % (D + 1)], n0 + (it + q) / SEGS, q % SEGS); }
  }
#pragma unroll
  for (int q = 0; q < UN; ++q) if (it + q < items) { issue(g[(q + D) % (D + 1)], it + q + D); compute(g[q % (D + 1)], n0 + (it + q) / SEGS, q % SEGS); }
  if (part > 0) {
#pragma unroll
    for (int j = 0; j < SEGS; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) park[(((grp * (split - 1) + part - 1) * SEGS + j) * 16 + i) * 64 + lane] = acc[j][i];
  }
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int j = 0; j < SEGS; ++j) {
      for (int p2 = 1; p2 < split; ++p2)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] += park[(((grp * (split - 1) + p2 - 1) * SEGS + j) * 16 + i) * 64 + lane];
      const int xt = xt0 + j * 128;
      if (xt < W) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          v2f v = {acc[j][c], acc[j][8 + c]};
          *reinterpret_cast<v2f*>(outp + ((long)b * 8 + c) * HW + (long)y * W + xt) = v;
        }
      }
    }
  }
}

// ---- forward shape, R rows per wave: a wave owns one 128-pixel segment of R CONSECUTIVE rows and all planes; per plane it
// loads its segment of the R rows back to back (R loads per tensor within 2.5 KB x R of one plane) before it hops to the
// next plane — does staying on a plane for R loads (DRAM page / TLB locality) buy what walking a row buys the backward?
template <int D, int K, int OCC, int R>
__global__ __launch_bounds__(5 * 64, OCC) void fwdrows(const float* __restrict__ A, const float* __restrict__ Bt,
                                                      const float* __restrict__ ctx_src, const int* __restrict__ kshift,
                                                      float* __restrict__ outp, int N, int H, int W, int Bn) {
  extern __shared__ v4f lds[];
  const int RS = W + 8;
  const int id = blockIdx.x, b = id % Bn, y0 = (id / Bn) * R;
  const long HW = (long)H * W;
  for (int x = threadIdx.x; x < RS * R; x += blockDim.x) {
    const int r = x / RS, xi = x - r * RS - 4;
    v4f cc = {0, 0, 0, 0};
    if (xi >= 0 && xi < W) { const float* p = ctx_src + ((long)b * 13) * HW + (long)(y0 + r) * W + xi; cc = v4f{p[0], p[HW], p[2 * HW], 0}; }
    lds[x] = cc;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xt0 = wave * 128 + lane * 2;
  const float* Ab = A + (long)b * N * HW + (long)y0 * W; const float* Bb = Bt + (long)b * N * HW + (long)y0 * W;
  float t[R][6], acc[R][16];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float* tp = ctx_src + ((long)b * 13 + 3) * HW + (long)(y0 + r) * W + min(xt0, W - 2);
    t[r][0] = tp[0]; t[r][1] = tp[1]; t[r][2] = tp[HW]; t[r][3] = tp[HW + 1]; t[r][4] = tp[2 * HW]; t[r][5] = tp[2 * HW + 1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
  }
  struct G { float a[R][3], b[R][3]; };
  auto issue = [&](G& g, int n_raw) {
    const int n = min(n_raw, N - 1);
    const int k = __builtin_amdgcn_readfirstlane(kshift[b * N + n]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned off = (unsigned)(r * W + xt0 + k) * 4;
      const v3f va = ld3(rsrc(Ab + (unsigned)(n * (int)HW), (R * W) * 4), off), vb = ld3(rsrc(Bb + (unsigned)(n * (int)HW), (R * W) * 4), off);
      g.a[r][0] = va.x; g.a[r][1] = va.y; g.a[r][2] = va.z; g.b[r][0] = vb.x; g.b[r][1] = vb.y; g.b[r][2] = vb.z;
    }
  };
  auto compute = [&](const G& g, int n) {
    const int k = __builtin_amdgcn_readfirstlane(kshift[b * N + n]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const v4f* cp = lds + r * RS + min(max(xt0 + k, -4), W + 1) + 4;
      const v4f c0 = cp[0], c1 = cp[1], c2 = cp[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float l = g.a[r][i] * 0.25f + g.a[r][i + 1] * 0.75f, s = g.b[r][i] * 0.25f + g.b[r][i + 1] * 0.75f;
        const v4f ca = i ? c1 : c0, cb = i ? c2 : c1;
        const float cr = ca.x * 0.25f + cb.x * 0.75f, cg = ca.y * 0.25f + cb.y * 0.75f, cbl = ca.z * 0.25f + cb.z * 0.75f;
        const float q = burn<K>(l + t[r][i], s + t[r][2 + i], cr + t[r][4 + i], cg + cbl);
        acc[r][i * 8 + 0] += q; acc[r][i * 8 + 1] += l; acc[r][i * 8 + 2] += s * q; acc[r][i * 8 + 3] += cr * q;
        acc[r][i * 8 + 4] += cg * q; acc[r][i * 8 + 5] += cbl * q; acc[r][i * 8 + 6] += q * l; acc[r][i * 8 + 7] += q * s;
      }
    }
  };
  G g[D + 1];
#pragma unroll
  for (int q = 0; q < D; ++q) issue(g[q], q);
  int n = 0;
  for (; n + (D + 1) <= N; n += D + 1) {
#pragma unroll
    for (int q = 0; q < D + 1; ++q) { issue(g[(q + D) % (D + 1)], n + q + D); compute(g[q], n + q); }
  }
#pragma unroll
  for (int q = 0; q < D + 1; ++q) if (n + q < N) { issue(g[(q + D) % (D + 1)], n + q + D); compute(g[q], n + q); }
  if (xt0 < W) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v2f v = {acc[r][c], acc[r][8 + c]};
        *reinterpret_cast<v2f*>(outp + ((long)b * 8 + c) * HW + (long)(y0 + r) * W + xt0) = v;
      }
  }
}

template <int D, int K, int OCC, int R> static void runr(int B, int N, int H, int W) {
  const size_t ldsb = (size_t)(W + 8) * 16 * R + 64;
  CK(hipFuncSetAttribute((const void*)fwdrows<D, K, OCC, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  const double ms = time_ms([&] { fwdrows<D, K, OCC, R><<<dim3(H / R * B), 5 * 64, ldsb>>>(A, Bt, ctx, ks, GA, N, H, W, B); });
  CK(hipGetLastError());
  const double bytes = (double)B * H * W * 4 * (2 * N + 9 + 8);
  printf("FWDROWS W=%4d D=%d K=%3d occ=%d rows=%d lds=%zu  %7.3f ms  %7.1f GB/s\n", W, D, K, OCC, R, ldsb, ms, bytes / 1e9 / (ms * 1e-3));
}

template <int D, int K, int WAVES, int OCC, int SEGS = 1> static void runf(int B, int N, int H, int W) {
  const int nseg = (W + 127) / 128, split = WAVES / (nseg / SEGS);
  const size_t ldsb = (size_t)(W + 8) * 16 + (size_t)nseg * (split - 1) * 16 * 64 * 4 + 64;
  CK(hipFuncSetAttribute((const void*)fwdstream<D, K, WAVES, OCC, SEGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  const double ms = time_ms([&] { fwdstream<D, K, WAVES, OCC, SEGS><<<dim3(H * B), WAVES * 64, ldsb>>>(A, Bt, ctx, ks, GA, N, H, W, B); });
  CK(hipGetLastError());
  const double bytes = (double)B * H * W * 4 * (2 * N + 9 + 8);
  printf("FWD W=%4d D=%d K=%3d waves=%d occ=%d segs=%d lds=%zu  %7.3f ms  %7.1f GB/s\n", W, D, K, WAVES, OCC, SEGS, ldsb, ms, bytes / 1e9 / (ms * 1e-3));
}

int main(int argc, char** argv) {
  const int B = 8, N = 49, H = 192, W = 640;
  const size_t n = (size_t)B * N * H * W;
  CK(hipMalloc(&A, n * 4 + 4096)); CK(hipMalloc(&Bt, n * 4 + 4096)); CK(hipMalloc(&GA, n * 4 + 4096)); CK(hipMalloc(&GB, n * 4 + 4096));
  CK(hipMalloc(&ctx, (size_t)B * 13 * H * W * 4)); CK(hipMalloc(&ks, B * N * 4)); CK(hipMalloc(&out, 64));
  fill<<<4096, 256>>>(A, n, 1u); fill<<<4096, 256>>>(Bt, n, 7u); fill<<<1024, 256>>>(ctx, (size_t)B * 13 * H * W, 3u);
  fill_k<<<(B * N + 255) / 256, 256>>>(ks, B * N, N, W, (argc > 2) ? 1 : 0);   // any second argument: even shifts only
  CK(hipDeviceSynchronize());
  if (argc > 1 && argv[1][0] == 'l') {   // loads alone / stores alone, by width, depth and workgroup size
    run<1, 2, 0, 1>(B, N, H, W); run<2, 2, 0, 1>(B, N, H, W); run<4, 2, 0, 1>(B, N, H, W);
    run<2, 4, 0, 1>(B, N, H, W); run<4, 4, 0, 1>(B, N, H, W); run<2, 2, 0, 1, 8, 4>(B, N, H, W); run<4, 2, 0, 1, 8, 4>(B, N, H, W);
    run<2, 4, 0, 1, 8, 4>(B, N, H, W); run<2, 6, 0, 1, 8, 4>(B, N, H, W);
    run<1, 2, 0, 2>(B, N, H, W); run<2, 2, 0, 2>(B, N, H, W); run<4, 2, 0, 2>(B, N, H, W); run<2, 2, 0, 2, 8, 4>(B, N, H, W);
    run<2, 2, 0, 3, 8, 4>(B, N, H, W); run<4, 2, 0, 3, 8, 4>(B, N, H, W); run<2, 4, 0, 3, 8, 4>(B, N, H, W);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 's') {   // VERDICT r3 #1c: source-ordered forward with the softmax state in LDS (R) against loads alone
    run<2, 3, 0, 1, 4, 4>(B, N, H, W); run<2, 3, 0, 9, 4, 4>(B, N, H, W); run<2, 3, 40, 9, 4, 4>(B, N, H, W); run<2, 3, 40, 13, 4, 4>(B, N, H, W);
    run<2, 3, 0, 1, 8, 4>(B, N, H, W); run<2, 3, 0, 9, 8, 4>(B, N, H, W); run<2, 3, 40, 9, 8, 4>(B, N, H, W); run<2, 3, 40, 13, 8, 4>(B, N, H, W);
    run<4, 2, 40, 9, 4, 4>(B, N, H, W); run<4, 2, 40, 9, 8, 4>(B, N, H, W); run<2, 3, 40, 9, 2, 4>(B, N, H, W); run<1, 4, 40, 9, 4, 4>(B, N, H, W);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'r') {   // forward shape, R rows per wave (five waves = one segment each; W = 640)
    runr<1, 0, 4, 1>(B, N, H, W); runr<2, 0, 4, 1>(B, N, H, W); runr<1, 0, 4, 2>(B, N, H, W); runr<2, 0, 4, 2>(B, N, H, W);
    runr<1, 0, 4, 4>(B, N, H, W); runr<1, 0, 3, 4>(B, N, H, W); runr<1, 0, 2, 8>(B, N, H, W);
    runr<1, 40, 4, 1>(B, N, H, W); runr<1, 40, 4, 2>(B, N, H, W); runr<2, 40, 4, 2>(B, N, H, W); runr<1, 40, 3, 4>(B, N, H, W);
    runr<1, 60, 4, 1>(B, N, H, W); runr<1, 60, 4, 2>(B, N, H, W); runr<1, 60, 3, 4>(B, N, H, W);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'f') {   // forward shape
    // W = 640: five segments.  One segment per wave (10 waves) against all five per wave (2 or 4 waves = plane halves / quarters)
    runf<2, 0, 10, 5>(B, N, H, W); runf<2, 40, 10, 5>(B, N, H, W); runf<1, 40, 10, 5>(B, N, H, W);
    runf<2, 0, 4, 4, 5>(B, N, H, W); runf<2, 40, 4, 4, 5>(B, N, H, W); runf<1, 40, 4, 4, 5>(B, N, H, W); runf<2, 60, 4, 4, 5>(B, N, H, W);
    runf<2, 40, 2, 4, 5>(B, N, H, W); runf<2, 40, 4, 3, 5>(B, N, H, W); runf<3, 40, 4, 3, 5>(B, N, H, W);
    // W = 512: four segments, two per wave (8 waves) against one per wave
    runf<2, 40, 8, 4>(B, N, H, 512); runf<2, 40, 4, 4, 2>(B, N, H, 512); runf<2, 40, 8, 4, 2>(B, N, H, 512); runf<2, 40, 2, 4, 4>(B, N, H, 512);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'F') {   // calibration of FETCH_SIZE on the segment-stream FORWARD's access shape (round 4): one wave per
    // 128-pixel segment, 12-byte loads at 4-byte alignment (xt + k), all planes; known bytes printed below
    runr<2, 0, 4, 1>(B, N, H, W);
    printf("known bytes per launch: tap loads %.0f, colour staging + target loads %.0f, stores %.0f\n", 2.0 * n * 4, 6.0 * B * H * W * 4, 8.0 * B * H * W * 4);
    return 0;
  }
  if (argc > 1) {   // calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (scripts/gpu_r3_profile.sh): the row-stream
    // backward's access shape (12-byte aligned loads, 8-byte aligned stores), loads alone and loads + stores
    run<2, 2, 0, 1>(B, N, H, W); run<2, 2, 0, 3>(B, N, H, W);
    printf("known bytes per launch: tap loads %.0f, context staging %.0f, stores %.0f\n", 2.0 * n * 4, 13.0 * B * H * W * 4, 2.0 * n * 4);
    return 0;
  }
  printf("-- memory shape alone, by prefetch depth\n");
  run<1, 1, 0, 3>(B, N, H, W); run<1, 2, 0, 3>(B, N, H, W); run<1, 4, 0, 3>(B, N, H, W);
  run<2, 1, 0, 3>(B, N, H, W); run<2, 2, 0, 3>(B, N, H, W); run<2, 4, 0, 3>(B, N, H, W);
  run<4, 1, 0, 3>(B, N, H, W); run<4, 2, 0, 3>(B, N, H, W); run<4, 4, 0, 3>(B, N, H, W);
  printf("-- + LDS context + 80 VALU per pixel-plane, by prefetch depth\n");
  run<1, 1, 80, 7>(B, N, H, W); run<1, 2, 80, 7>(B, N, H, W); run<1, 3, 80, 7>(B, N, H, W); run<1, 4, 80, 7>(B, N, H, W); run<1, 6, 80, 7>(B, N, H, W);
  run<2, 1, 80, 7>(B, N, H, W); run<2, 2, 80, 7>(B, N, H, W); run<2, 3, 80, 7>(B, N, H, W); run<2, 4, 80, 7>(B, N, H, W);
  run<4, 1, 80, 7>(B, N, H, W); run<4, 2, 80, 7>(B, N, H, W); run<4, 3, 80, 7>(B, N, H, W);
  printf("-- 100 / 120 VALU\n");
  run<1, 4, 100, 7>(B, N, H, W); run<1, 4, 120, 7>(B, N, H, W); run<2, 3, 100, 7>(B, N, H, W); run<2, 3, 120, 7>(B, N, H, W); run<4, 2, 100, 7>(B, N, H, W); run<4, 2, 120, 7>(B, N, H, W);
  printf("-- occupancy / workgroup size\n");
  run<1, 4, 100, 7, 4, 3>(B, N, H, W); run<2, 3, 100, 7, 4, 3>(B, N, H, W); run<2, 3, 100, 7, 4, 2>(B, N, H, W); run<4, 2, 100, 7, 4, 2>(B, N, H, W);
  run<1, 4, 100, 7, 8, 4>(B, N, H, W); run<2, 3, 100, 7, 8, 4>(B, N, H, W); run<2, 3, 100, 7, 2, 4>(B, N, H, W);
  printf("-- arithmetic + LDS alone\n");
  run<1, 1, 80, 4>(B, N, H, W); run<1, 1, 100, 4>(B, N, H, W); run<1, 1, 120, 4>(B, N, H, W); run<2, 1, 100, 4>(B, N, H, W);
  return 0;
}
