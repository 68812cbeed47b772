// Geometry modules of the reference as standalone kernels:
//   BackprojectDepth.forward  layers.py:150-156     cam = depth * (K^-1 [x,y,1]) , homogeneous
//   Project3D.forward         layers.py:169-182     grid = normalise( (P cam).xy / ((P cam).z + eps) )
//   HomographyWarp.forward    layers.py:221-233     per-pixel part: p = H_t2s [x,y,1], mask, clamp z, divide, normalise
// The [*,3,3] / [*,4,4] matrix algebra in front of them (K@T, R + t n^T/d, torch.inverse) is O(B*N) work and stays in
// torch on the host side of the boundary (planedepth_amd/layers.py) — see DESIGN.md.
#include "pd_common.h"

namespace pd {

__global__ __launch_bounds__(kBlock) void backproject_kernel(int HW, int W, const float* __restrict__ depth,
                                                             const float* __restrict__ inv_K, float* __restrict__ cam) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= HW) return;
  const int b = blockIdx.y;
  const float* Ki = inv_K + (long)b * 16;  // [4,4] row-major; uses [:3,:3]
  const float y = (float)(pix / W), x = (float)(pix % W);
  const float d = depth[(long)b * HW + pix];
  float* o = cam + (long)b * 4 * HW + pix;
  o[0] = d * hrow_dot(Ki[0], Ki[1], Ki[2], x, y);      // (layers.py:152 torch.matmul's rounding order: pd_common.h)
  o[HW] = d * hrow_dot(Ki[4], Ki[5], Ki[6], x, y);
  o[2 * HW] = d * hrow_dot(Ki[8], Ki[9], Ki[10], x, y);
  o[3 * HW] = 1.0f;
}

__global__ __launch_bounds__(kBlock) void backproject_bwd_kernel(int HW, int W, const float* __restrict__ inv_K,
                                                                 const float* __restrict__ g_cam,
                                                                 float* __restrict__ g_depth) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= HW) return;
  const int b = blockIdx.y;
  const float* Ki = inv_K + (long)b * 16;
  const float y = (float)(pix / W), x = (float)(pix % W);
  const float* g = g_cam + (long)b * 4 * HW + pix;
  g_depth[(long)b * HW + pix] = g[0] * (Ki[0] * x + Ki[1] * y + Ki[2]) + g[HW] * (Ki[4] * x + Ki[5] * y + Ki[6]) +
                                g[2 * HW] * (Ki[8] * x + Ki[9] * y + Ki[10]);
}

__global__ __launch_bounds__(kBlock) void project3d_kernel(int H, int W, float eps, const float* __restrict__ cam,
                                                           const float* __restrict__ P, float* __restrict__ grid) {
  const int HW = H * W;
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= HW) return;
  const int b = blockIdx.y;
  const float* Pm = P + (long)b * 12;  // [3,4]
  const float* c = cam + (long)b * 4 * HW + pix;
  const float X = c[0], Y = c[HW], Z = c[2 * HW], Wh = c[3 * HW];
  const float p0 = hrow_dot4(Pm[0], Pm[1], Pm[2], Pm[3], X, Y, Z, Wh);   // (layers.py:174 torch.matmul's rounding order)
  const float p1 = hrow_dot4(Pm[4], Pm[5], Pm[6], Pm[7], X, Y, Z, Wh);
  const float z = hrow_dot4(Pm[8], Pm[9], Pm[10], Pm[11], X, Y, Z, Wh) + eps;
  float2 o;
  o.x = normalise(p0 / z, (float)(W - 1));
  o.y = normalise(p1 / z, (float)(H - 1));
  reinterpret_cast<float2*>(grid)[(long)b * HW + pix] = o;
}

// g_cam[b,:,pix] and per-block partial sums of g_P[b,3,4]
__global__ __launch_bounds__(kBlock) void project3d_bwd_kernel(int H, int W, float eps, const float* __restrict__ cam,
                                                               const float* __restrict__ P,
                                                               const float* __restrict__ g_grid,
                                                               float* __restrict__ g_cam, float* __restrict__ partials) {
  __shared__ float red[12];
  const int HW = H * W;
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  const int b = blockIdx.y;
  if (threadIdx.x < 12) red[threadIdx.x] = 0.0f;
  __syncthreads();
  float gk[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) gk[k] = 0.0f;
  if (pix < HW) {
    const float* Pm = P + (long)b * 12;
    const float* c = cam + (long)b * 4 * HW + pix;
    const float v[4] = {c[0], c[HW], c[2 * HW], c[3 * HW]};
    const float p0 = Pm[0] * v[0] + Pm[1] * v[1] + Pm[2] * v[2] + Pm[3] * v[3];
    const float p1 = Pm[4] * v[0] + Pm[5] * v[1] + Pm[6] * v[2] + Pm[7] * v[3];
    const float z = Pm[8] * v[0] + Pm[9] * v[1] + Pm[10] * v[2] + Pm[11] * v[3] + eps;
    const float2 g = reinterpret_cast<const float2*>(g_grid)[(long)b * HW + pix];
    const float gpx = g.x * 2.0f / (float)(W - 1), gpy = g.y * 2.0f / (float)(H - 1);
    const float g0 = gpx / z, g1 = gpy / z, g2 = -(gpx * p0 + gpy * p1) / (z * z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gk[j] = g0 * v[j];
      gk[4 + j] = g1 * v[j];
      gk[8 + j] = g2 * v[j];
    }
    if (g_cam) {
      float* gc = g_cam + (long)b * 4 * HW + pix;
#pragma unroll
      for (int j = 0; j < 4; ++j) gc[(long)j * HW] = g0 * Pm[j] + g1 * Pm[4 + j] + g2 * Pm[8 + j];
    }
  }
  if (partials) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float s = wave_sum(gk[k]);
      if ((threadIdx.x & (kWave - 1)) == 0) lds_add(&red[k], s);
    }
    __syncthreads();
    if (threadIdx.x < 12) partials[((long)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = red[threadIdx.x];
  }
}

__global__ __launch_bounds__(kBlock) void homography_grid_kernel(int H, int W, const float* __restrict__ Ht2s,
                                                                 const float* __restrict__ Rn,
                                                                 const float* __restrict__ invK3,
                                                                 float* __restrict__ grid, uint8_t* __restrict__ mask) {
  const int HW = H * W;
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= HW) return;
  const int m = blockIdx.y;
  const float* Hm = Ht2s + (long)m * 9;
  const float* Ki = invK3 + (long)m * 9;
  const float* rn = Rn + (long)m * 3;
  const float fy = (float)(pix / W), fx = (float)(pix % W);
  const float p0 = hrow_dot(Hm[0], Hm[1], Hm[2], fx, fy);
  const float p1 = hrow_dot(Hm[3], Hm[4], Hm[5], fx, fy);
  const float z = hrow_dot(Hm[6], Hm[7], Hm[8], fx, fy);
  const float facing = facing_dot(hrow_dot(Ki[0], Ki[1], Ki[2], fx, fy), hrow_dot(Ki[3], Ki[4], Ki[5], fx, fy),
                                  hrow_dot(Ki[6], Ki[7], Ki[8], fx, fy), rn[0], rn[1], rn[2]);
  const float zc = (z < 1e-7f) ? 1e-7f : z;
  float2 o;
  o.x = normalise(p0 / zc, (float)(W - 1));
  o.y = normalise(p1 / zc, (float)(H - 1));
  reinterpret_cast<float2*>(grid)[(long)m * HW + pix] = o;
  if (mask) mask[(long)m * HW + pix] = (facing > 0.0f && z > 1e-7f) ? 1 : 0;
}

// d loss / d H_t2s [M,3,3] from the grid gradient: per plane nine sums over all pixels.  A workgroup covers
// kHgPix * 256 pixels (strided, coalesced), every thread accumulates its pixels' nine terms, then ONE set of wave
// reductions per workgroup (DPP, two components per reduction) — the first version reduced after every pixel through
// ds_bpermute shuffles and divided five times per pixel: 0.50 ms for 392 planes of 192x640, bound by neither memory nor
// arithmetic; this form reads the 385 MB of g_grid at memory speed.
constexpr int kHgPix = 8;

__global__ __launch_bounds__(kBlock) void homography_grid_bwd_kernel(int H, int W, const float* __restrict__ Ht2s,
                                                                     const float* __restrict__ g_grid,
                                                                     float* __restrict__ partials) {
  __shared__ float red[9];
  const int HW = H * W;
  const int m = blockIdx.y;
  if (threadIdx.x < 9) red[threadIdx.x] = 0.0f;
  __syncthreads();
  float gk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) gk[k] = 0.0f;
  const float* Hm = Ht2s + (long)m * 9;
  const float sx = 2.0f / (float)(W - 1), sy = 2.0f / (float)(H - 1);
  const int base = blockIdx.x * (kBlock * kHgPix) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kHgPix; ++i) {
    const int pix = base + i * kBlock;
    if (pix < HW) {
      const int iy = pix / W;
      const float fy = (float)iy, fx = (float)(pix - iy * W);
      const float p0 = hrow_dot(Hm[0], Hm[1], Hm[2], fx, fy);
      const float p1 = hrow_dot(Hm[3], Hm[4], Hm[5], fx, fy);
      const float z = hrow_dot(Hm[6], Hm[7], Hm[8], fx, fy);
      const bool clamped = z < 1e-7f;
      const float zc = clamped ? 1e-7f : z;
      const float2 g = reinterpret_cast<const float2*>(g_grid)[(long)m * HW + pix];
      float iz = fast_rcp(zc);
      iz = fmaf(fmaf(-zc, iz, 1.0f), iz, iz);   // refined reciprocal (gradient side: no bit-exactness at stake)
      const float g0 = g.x * sx * iz, g1 = g.y * sy * iz;
      const float g2 = clamped ? 0.0f : -(g0 * p0 + g1 * p1) * iz;
      gk[0] += g0 * fx; gk[1] += g0 * fy; gk[2] += g0;
      gk[3] += g1 * fx; gk[4] += g1 * fy; gk[5] += g1;
      gk[6] += g2 * fx; gk[7] += g2 * fy; gk[8] += g2;
    }
  }
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const float v = half_wave_sums_hi(gk[k], gk[k + 1]);   // totals in lanes 31 and 63
    if ((lane & 31) == 31) lds_add(&red[k + (lane >> 5)], v);
  }
  {
    const float v = wave_sum_hi(gk[8]);
    if (lane == kWave - 1) lds_add(&red[8], v);
  }
  __syncthreads();
  if (threadIdx.x < 9) partials[((long)m * gridDim.x + blockIdx.x) * 9 + threadIdx.x] = red[threadIdx.x];
}

// partials [Bo][nblk][M] -> out [Bo][M] in a fixed order: one wave per output element, lanes stride over the blocks
__global__ void reduce_small_kernel(const float* __restrict__ partials, float* __restrict__ out, int nblk, int M) {
  const int j = blockIdx.x, b = blockIdx.y;
  const float* p = partials + (long)b * nblk * M + j;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < nblk; i += kWave) acc += p[(long)i * M];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * M + j] = acc;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_backproject(int B, int H, int W, const float* depth, const float* inv_K, float* cam,
                              pd_stream_t stream) {
  PD_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, "bad shape");
  PD_REQUIRE(depth && inv_K && cam, "NULL pointer");
  backproject_kernel<<<dim3(ceil_div(H * W, kBlock), B), kBlock, 0, (hipStream_t)stream>>>(H * W, W, depth, inv_K, cam);
  return check_launch("backproject_kernel");
}

extern "C" int pd_backproject_bwd(int B, int H, int W, const float* inv_K, const float* g_cam, float* g_depth,
                                  pd_stream_t stream) {
  PD_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, "bad shape");
  PD_REQUIRE(inv_K && g_cam && g_depth, "NULL pointer");
  backproject_bwd_kernel<<<dim3(ceil_div(H * W, kBlock), B), kBlock, 0, (hipStream_t)stream>>>(H * W, W, inv_K, g_cam,
                                                                                                g_depth);
  return check_launch("backproject_bwd_kernel");
}

extern "C" int pd_project3d(int B, int H, int W, float eps, const float* cam, const float* P, float* grid,
                            pd_stream_t stream) {
  PD_REQUIRE(B > 0 && B <= 65535 && H > 1 && W > 1, "bad shape");
  PD_REQUIRE(cam && P && grid, "NULL pointer");
  project3d_kernel<<<dim3(ceil_div(H * W, kBlock), B), kBlock, 0, (hipStream_t)stream>>>(H, W, eps, cam, P, grid);
  return check_launch("project3d_kernel");
}

extern "C" int pd_project3d_bwd(int B, int H, int W, float eps, const float* cam, const float* P, const float* g_grid,
                                float* g_cam, float* g_P, float* workspace, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && B <= 65535 && H > 1 && W > 1, "bad shape");
  PD_REQUIRE(cam && P && g_grid, "NULL pointer");
  PD_REQUIRE(!g_P || workspace, "g_P needs workspace");
  const int nblk = ceil_div(H * W, kBlock);
  project3d_bwd_kernel<<<dim3(nblk, B), kBlock, 0, (hipStream_t)stream>>>(H, W, eps, cam, P, g_grid, g_cam,
                                                                          g_P ? workspace : nullptr);
  int rc = check_launch("project3d_bwd_kernel");
  if (rc || !g_P) return rc;
  reduce_small_kernel<<<dim3(12, B), kWave, 0, (hipStream_t)stream>>>(workspace, g_P, nblk, 12);
  return check_launch("reduce_small_kernel");
}

extern "C" int pd_homography_grid(int M, int H, int W, const float* H_t2s, const float* Rn, const float* inv_K3,
                                  float* grid, uint8_t* mask, pd_stream_t stream) {
  PD_REQUIRE(M > 0 && M <= 65535 && H > 1 && W > 1, "bad shape");
  PD_REQUIRE(H_t2s && Rn && inv_K3 && grid, "NULL pointer");
  homography_grid_kernel<<<dim3(ceil_div(H * W, kBlock), M), kBlock, 0, (hipStream_t)stream>>>(H, W, H_t2s, Rn, inv_K3,
                                                                                                grid, mask);
  return check_launch("homography_grid_kernel");
}

extern "C" int pd_homography_grid_bwd(int M, int H, int W, const float* H_t2s, const float* g_grid, float* g_H,
                                      float* workspace, pd_stream_t stream) {
  PD_REQUIRE(M > 0 && M <= 65535 && H > 1 && W > 1, "bad shape");
  PD_REQUIRE(H_t2s && g_grid && g_H && workspace, "NULL pointer");
  const int nblk = ceil_div(H * W, kBlock * kHgPix);   // <= the documented workspace of 9 * M * ceil(H*W/256) floats
  homography_grid_bwd_kernel<<<dim3(nblk, M), kBlock, 0, (hipStream_t)stream>>>(H, W, H_t2s, g_grid, workspace);
  int rc = check_launch("homography_grid_bwd_kernel");
  if (rc) return rc;
  reduce_small_kernel<<<dim3(9, M), kWave, 0, (hipStream_t)stream>>>(workspace, g_H, nblk, 9);
  return check_launch("reduce_small_kernel");
}
