// The O(B*N) 3x3 algebra in front of the homography sweep (reference layers.py:206-219, 223-225):
//   M = R + t n^T / d ;  H_s2t = K M K^-1 ;  H_t2s = inverse(H_s2t) ;  R n  (for the facing test)
// and its adjoint, as ONE launch each instead of the ~12 torch operators + the rocSOLVER inverse (which also cannot be
// captured in a HIP graph).  B*N is a few hundred matrices: the cost is launch latency, not arithmetic, so everything is
// evaluated in fp64 and rounded once to fp32 — closer to the exact value than either fp32 evaluation (torch.inverse is
// LAPACK on the reference's CPU runs, cuSOLVER / rocSOLVER on GPUs: they differ in the last bits among themselves,
// tests/test_gpu_parity.py three-way bounds).
//
// Modes (include/planedepth_hip.h):
//   PD_HMAT_PLANES   one homography per (image, plane):            H_t2s [B,N,3,3], Rn [B,N,3]
//   PD_HMAT_UNIFORM  zero-translation poses (Trainer.predict_poses without COLMAP, trainer.py:386-400): slice 0 is the
//                    image's homography (plane 0, translation detached), slices 1..3 the virtual planes n/d = e_j that
//                    carry the translation's gradient:             H_t2s [B,4,3,3], Rn [B,N,3]
//   PD_HMAT_STEREO_ROWS  identity rotation + x translation + normals without an x component: the warp is the shift
//                    h01*y + h02 per (plane, row), the facing test is constant along x:
//                                                                  shift [B,N,rows], mask [B,N,rows]  (H_t2s optional)
#include "pd_common.h"

namespace pd {

struct M3 {
  double m[3][3];
};

__device__ __forceinline__ M3 mul3(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return c;
}

__device__ __forceinline__ M3 transpose3(const M3& a) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[i][j] = a.m[j][i];
  return c;
}

// adjugate / determinant in fp64: cond(H) ~ 1e3..1e4 here, far inside what 53 bits carry.  A singular matrix yields
// inf / nan entries (torch.inverse raises instead; the sweep then masks nothing sensible either way).
__device__ __forceinline__ M3 inverse3(const M3& a) {
  M3 c;
  const double c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  const double c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  const double c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  const double det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
  const double r = 1.0 / det;
  c.m[0][0] = c00 * r;
  c.m[1][0] = c01 * r;
  c.m[2][0] = c02 * r;
  c.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * r;
  c.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * r;
  c.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * r;
  c.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * r;
  c.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * r;
  c.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * r;
  return c;
}

__device__ __forceinline__ M3 load3_from44(const float* __restrict__ p) {   // the upper-left 3x3 of a [4,4] matrix
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[i][j] = (double)p[i * 4 + j];
  return c;
}

struct PlaneOf {   // (n, d) of homography slot j in the given mode
  double n[3], d;
};

__device__ __forceinline__ PlaneOf plane_of(int mode, int b, int j, int N, const float* __restrict__ distance,
                                            const float* __restrict__ norm) {
  PlaneOf p;
  if (mode == PD_HMAT_UNIFORM && j > 0) {
    p.n[0] = j == 1, p.n[1] = j == 2, p.n[2] = j == 3, p.d = 1.0;
    return p;
  }
  const long k = (long)b * N + (mode == PD_HMAT_UNIFORM ? 0 : j);
  p.n[0] = norm[k * 3 + 0], p.n[1] = norm[k * 3 + 1], p.n[2] = norm[k * 3 + 2], p.d = distance[k];
  return p;
}

struct Chain {
  M3 K, Ki, H;      // intrinsics, their inverse as handed over, H_t2s
  double t[3];
};

__device__ __forceinline__ Chain homography_chain(const float* __restrict__ T, const float* __restrict__ K,
                                                  const float* __restrict__ inv_K, int b, const PlaneOf& p) {
  Chain c;
  c.K = load3_from44(K + (long)b * 16);
  c.Ki = load3_from44(inv_K + (long)b * 16);
  M3 M = load3_from44(T + (long)b * 16);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c.t[i] = (double)T[(long)b * 16 + i * 4 + 3];
#pragma unroll
    for (int j = 0; j < 3; ++j) M.m[i][j] += c.t[i] * p.n[j] / p.d;   // layers.py:216
  }
  c.H = inverse3(mul3(c.K, mul3(M, c.Ki)));                            // layers.py:218-219
  return c;
}

__global__ __launch_bounds__(kWave) void homography_matrices_fwd_kernel(
    int B, int N, int NH, int mode, int rows, const float* __restrict__ distance, const float* __restrict__ norm,
    const float* __restrict__ T, const float* __restrict__ K, const float* __restrict__ inv_K, float* __restrict__ H_t2s,
    float* __restrict__ Rn, float* __restrict__ shift, float* __restrict__ mask) {
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < max(N, NH); j += kWave) {
    float rn[3] = {0.0f, 0.0f, 0.0f};
    if (j < N) {   // R n of the real plane j (layers.py:223)
      const long k = (long)b * N + j;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) s += (double)T[(long)b * 16 + i * 4 + q] * (double)norm[k * 3 + q];
        rn[i] = (float)s;
        if (Rn) Rn[k * 3 + i] = rn[i];
      }
    }
    if (j >= NH) continue;
    const Chain c = homography_chain(T, K, inv_K, b, plane_of(mode, b, j, N, distance, norm));
    float h[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        h[i][q] = (float)c.H.m[i][q];
        if (H_t2s) H_t2s[((long)b * NH + j) * 9 + i * 3 + q] = h[i][q];
      }
  }
}

// PD_HMAT_STEREO_ROWS: one wave per (plane, image); the 3x3 chain is evaluated by every lane (a few hundred flops) and the
// lanes stride over the rows.  (As a serial loop over the rows inside the per-image kernel above these took 13 + 22 us
// per step at 8 x 49 x 192.)
__global__ __launch_bounds__(kWave) void stereo_rows_fwd_kernel(
    int N, int rows, const float* __restrict__ distance, const float* __restrict__ norm, const float* __restrict__ T,
    const float* __restrict__ K, const float* __restrict__ inv_K, float* __restrict__ H_t2s, float* __restrict__ Rn,
    float* __restrict__ shift, float* __restrict__ mask) {
  const int j = blockIdx.x, b = blockIdx.y;
  const long k = (long)b * N + j;
  float rn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {   // R n (layers.py:223)
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) s += (double)T[(long)b * 16 + i * 4 + q] * (double)norm[k * 3 + q];
    rn[i] = (float)s;
    if (Rn && threadIdx.x == 0) Rn[k * 3 + i] = rn[i];
  }
  const Chain c = homography_chain(T, K, inv_K, b, plane_of(PD_HMAT_STEREO_ROWS, b, j, N, distance, norm));
  if (H_t2s && threadIdx.x < 9) H_t2s[k * 9 + threadIdx.x] = (float)c.H.m[threadIdx.x / 3][threadIdx.x % 3];
  // per-row shift and mask (layers.py:219-229 at x = 0: h00 = 1 and the x terms of the facing test vanish)
  const float ik01 = inv_K[(long)b * 16 + 1], ik02 = inv_K[(long)b * 16 + 2], ik11 = inv_K[(long)b * 16 + 5],
              ik12 = inv_K[(long)b * 16 + 6], ik21 = inv_K[(long)b * 16 + 9], ik22 = inv_K[(long)b * 16 + 10];
  const float h21 = (float)c.H.m[2][1], h22 = (float)c.H.m[2][2];
  for (int y = threadIdx.x; y < rows; y += kWave) {
#pragma clang fp contract(off)
    const float fy = (float)y;
    const float facing = (ik01 * fy + ik02) * rn[0] + (ik11 * fy + ik12) * rn[1] + (ik21 * fy + ik22) * rn[2];
    const float z = h21 * fy + h22;
    shift[k * rows + y] = (float)(c.H.m[0][1] * (double)y + c.H.m[0][2]);   // one rounding
    mask[k * rows + y] = (facing > 0.0f && z > 1e-7f) ? 1.0f : 0.0f;
  }
}

// shift = h01 * y + h02  ->  g_h01 = sum_y y g, g_h02 = sum_y g  ->  (adjoint of the chain)  ->  g_distance
__global__ __launch_bounds__(kWave) void stereo_rows_bwd_kernel(
    int N, int rows, const float* __restrict__ distance, const float* __restrict__ norm, const float* __restrict__ T,
    const float* __restrict__ K, const float* __restrict__ inv_K, const float* __restrict__ g_H,
    const float* __restrict__ g_shift, float* __restrict__ g_distance) {
  const int j = blockIdx.x, b = blockIdx.y;
  const long k = (long)b * N + j;
  double s1 = 0.0, s0 = 0.0;
  if (g_shift)
    for (int y = threadIdx.x; y < rows; y += kWave) {
      const double g = (double)g_shift[k * rows + y];
      s1 += g * (double)y;
      s0 += g;
    }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, kWave); s0 += __shfl_xor(s0, off, kWave); }
  if (threadIdx.x != 0 || !g_distance) return;
  const PlaneOf p = plane_of(PD_HMAT_STEREO_ROWS, b, j, N, distance, norm);
  const Chain c = homography_chain(T, K, inv_K, b, p);
  M3 gH;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int q = 0; q < 3; ++q) gH.m[i][q] = g_H ? (double)g_H[k * 9 + i * 3 + q] : 0.0;
  gH.m[0][1] += s1;
  gH.m[0][2] += s0;
  const M3 Ht = transpose3(c.H);
  const M3 gA = mul3(Ht, mul3(gH, Ht));            // (sign applied below)
  const M3 gM = mul3(transpose3(c.K), mul3(gA, transpose3(c.Ki)));
  double tgMn = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) tgMn += c.t[i] * (gM.m[i][0] * p.n[0] + gM.m[i][1] * p.n[1] + gM.m[i][2] * p.n[2]);
  g_distance[k] = (float)(tgMn / (p.d * p.d));     // -(-t^T gM n) / d^2
}

// Adjoint.  dH_t2s = -H dA H  ->  gA = -H^T gH H^T ;  A = K M K^-1  ->  gM = K^T gA K^-T ;
// M = R + t n^T / d  ->  gR = gM, gt = gM n / d, gn = gM^T t / d, gd = -(t^T gM n) / d^2.
// One wave per image: lanes stride over the planes, the [3,4] pose gradient is reduced across the wave (fixed order).
__global__ __launch_bounds__(kWave) void homography_matrices_bwd_kernel(
    int B, int N, int NH, int mode, int rows, const float* __restrict__ distance, const float* __restrict__ norm,
    const float* __restrict__ T, const float* __restrict__ K, const float* __restrict__ inv_K,
    const float* __restrict__ g_H, const float* __restrict__ g_shift, float* __restrict__ g_distance,
    float* __restrict__ g_norm, float* __restrict__ g_T) {
  const int b = blockIdx.x;
  double gT[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) gT[i] = 0.0;
  for (int j = threadIdx.x; j < NH; j += kWave) {
    const PlaneOf p = plane_of(mode, b, j, N, distance, norm);
    const Chain c = homography_chain(T, K, inv_K, b, p);
    M3 gH;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) gH.m[i][q] = g_H ? (double)g_H[((long)b * NH + j) * 9 + i * 3 + q] : 0.0;
    const M3 Ht = transpose3(c.H);
    M3 gA = mul3(Ht, mul3(gH, Ht));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) gA.m[i][q] = -gA.m[i][q];
    const M3 gM = mul3(transpose3(c.K), mul3(gA, transpose3(c.Ki)));
    double gMn[3], tgM[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      gMn[i] = gM.m[i][0] * p.n[0] + gM.m[i][1] * p.n[1] + gM.m[i][2] * p.n[2];
      tgM[i] = c.t[0] * gM.m[0][i] + c.t[1] * gM.m[1][i] + c.t[2] * gM.m[2][i];
    }
    if (mode == PD_HMAT_UNIFORM) {
      // slice 0 owns the rotation (its translation is detached), slices 1..3 the translation (their rotation is detached)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (j == 0) {
#pragma unroll
          for (int q = 0; q < 3; ++q) gT[i * 4 + q] += gM.m[i][q];
        } else {
          gT[i * 4 + 3] += gMn[i];   // n = e_(j-1), d = 1
        }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int q = 0; q < 3; ++q) gT[i * 4 + q] += gM.m[i][q];
      gT[i * 4 + 3] += gMn[i] / p.d;
    }
    const long k = (long)b * N + j;
    if (g_distance) g_distance[k] = (float)(-(c.t[0] * gMn[0] + c.t[1] * gMn[1] + c.t[2] * gMn[2]) / (p.d * p.d));
    if (g_norm) {
#pragma unroll
      for (int i = 0; i < 3; ++i) g_norm[k * 3 + i] = (float)(tgM[i] / p.d);
    }
  }
  if (mode == PD_HMAT_UNIFORM) {   // no plane carries a distance / normal gradient there (t = 0 in the real chain)
    for (int j = threadIdx.x; j < N; j += kWave) {
      if (g_distance) g_distance[(long)b * N + j] = 0.0f;
      if (g_norm) g_norm[((long)b * N + j) * 3 + 0] = g_norm[((long)b * N + j) * 3 + 1] = g_norm[((long)b * N + j) * 3 + 2] = 0.0f;
    }
  }
  if (!g_T) return;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    double v = gT[i];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    gT[i] = v;
  }
  if (threadIdx.x < 16) {
    const int i = threadIdx.x;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 12; ++q) v = (i == q) ? gT[q] : v;
    g_T[(long)b * 16 + i] = i < 12 ? (float)v : 0.0f;   // [3,4] block; the last row of the [4,4] pose is never read
  }
}

}  // namespace pd

using namespace pd;

static int homography_slots(int mode, int N) { return mode == PD_HMAT_UNIFORM ? 4 : N; }

extern "C" int pd_homography_matrices_fwd(int B, int N, int mode, int rows, const float* distance, const float* norm,
                                          const float* T, const float* K, const float* inv_K, float* H_t2s, float* Rn,
                                          float* shift, float* mask, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && N > 0, "bad shape");
  PD_REQUIRE(mode == PD_HMAT_PLANES || mode == PD_HMAT_UNIFORM || mode == PD_HMAT_STEREO_ROWS, "bad mode");
  PD_REQUIRE(distance && norm && T && K && inv_K, "null input");
  PD_REQUIRE(mode == PD_HMAT_STEREO_ROWS ? (rows > 0 && shift && mask) : H_t2s != nullptr, "null output");
  if (mode == PD_HMAT_STEREO_ROWS) {
    PD_REQUIRE(B <= 65535, "bad shape");
    stereo_rows_fwd_kernel<<<dim3(N, B), kWave, 0, (hipStream_t)stream>>>(N, rows, distance, norm, T, K, inv_K, H_t2s, Rn,
                                                                            shift, mask);
    return check_launch("stereo_rows_fwd_kernel");
  }
  homography_matrices_fwd_kernel<<<B, kWave, 0, (hipStream_t)stream>>>(B, N, homography_slots(mode, N), mode, rows,
                                                                        distance, norm, T, K, inv_K, H_t2s, Rn, shift, mask);
  return check_launch("homography_matrices_fwd_kernel");
}

extern "C" int pd_homography_matrices_bwd(int B, int N, int mode, int rows, const float* distance, const float* norm,
                                          const float* T, const float* K, const float* inv_K, const float* g_H,
                                          const float* g_shift, float* g_distance, float* g_norm, float* g_T,
                                          pd_stream_t stream) {
  PD_REQUIRE(B > 0 && N > 0, "bad shape");
  PD_REQUIRE(mode == PD_HMAT_PLANES || mode == PD_HMAT_UNIFORM || mode == PD_HMAT_STEREO_ROWS, "bad mode");
  PD_REQUIRE(distance && norm && T && K && inv_K, "null input");
  PD_REQUIRE(g_H || (mode == PD_HMAT_STEREO_ROWS && g_shift && rows > 0), "no upstream gradient");
  if (mode == PD_HMAT_STEREO_ROWS) {   // g_distance only (see the header): the pose and the normals are constants there
    PD_REQUIRE(B <= 65535 && !g_norm && !g_T, "PD_HMAT_STEREO_ROWS carries the gradient of distance only");
    stereo_rows_bwd_kernel<<<dim3(N, B), kWave, 0, (hipStream_t)stream>>>(N, rows, distance, norm, T, K, inv_K, g_H, g_shift,
                                                                            g_distance);
    return check_launch("stereo_rows_bwd_kernel");
  }
  homography_matrices_bwd_kernel<<<B, kWave, 0, (hipStream_t)stream>>>(B, N, homography_slots(mode, N), mode, rows,
                                                                        distance, norm, T, K, inv_K, g_H, g_shift,
                                                                        g_distance, g_norm, g_T);
  return check_launch("homography_matrices_bwd_kernel");
}
