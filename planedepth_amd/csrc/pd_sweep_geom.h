// Per-(pixel, plane) sampling geometry shared by the general sweep kernels (pd_plane_sweep.hip, pd_plane_sweep_tile.hip).
#pragma once
#include "pd_sweep.h"

namespace pd {

// ---------------------------------------------------------------------------------------------------------------
// Sampling position of target pixel (x,y) on plane n of image b, plus the padding mask.
// DISP:        trainer.py:540-554   (x + sign*d, y) normalised by (W-1, H-1)
// HOMOGRAPHY:  layers.py:221-233    p = H_t2s [x,y,1]; mask = ((K^-1 p_t).(R n) > 0) & (z > 1e-7); z<1e-7 -> 1e-7
// ---------------------------------------------------------------------------------------------------------------
struct PlaneGeom {   // per (pixel, plane) state needed again for the grid gradient
  float ix, iy;
  float p0, p1, zc;  // homography only
  bool z_clamped;
};

// Sizes the coordinates are normalised by, with their refined reciprocals (uniform; computed once per thread).
struct CoordNorm {
  float Wm1, Hm1, rcpW, rcpH;
  bool fast;  // sizes >= 2: the reciprocal form is exact (a size of 1 divides by zero; keep IEEE semantics there)
};
__device__ __forceinline__ CoordNorm make_coord_norm(int W, int H) {
  CoordNorm c;
  c.Wm1 = (float)(W - 1); c.Hm1 = (float)(H - 1);
  c.fast = (W >= 2) && (H >= 2);
  c.rcpW = c.fast ? refined_rcp(c.Wm1) : 0.0f;
  c.rcpH = c.fast ? refined_rcp(c.Hm1) : 0.0f;
  return c;
}

template <int MODE>
__device__ __forceinline__ PlaneGeom plane_coords(const SweepArgs& a, const CoordNorm& cn, int b, int n, int x, int y,
                                                  float iy_disp, bool& mask) {
  PlaneGeom g;
  if (MODE == PD_WARP_DISP) {
    float d;
    if (a.flags & PD_DISP_DENSE)
      d = a.plane[(((long)b * a.N + n) * a.H + y) * a.W + x];
    else
      d = a.plane[b * a.N + n];
    g.ix = normalise_roundtrip((float)x + a.sign * d, (float)(a.W - 1));
    g.iy = iy_disp;
    g.p0 = g.p1 = g.zc = 0.0f;
    g.z_clamped = false;
    mask = true;  // caller applies the padding_mask tensor
  } else {
    const float* Hm = a.plane + ((long)b * a.N + n) * 9;
    const float* Rn = a.plane_aux + ((long)b * a.N + n) * 3;
    const float* Ki = a.inv_K3 + (long)b * 9;
    const float fx = (float)x, fy = (float)y;
    g.p0 = hrow_dot(Hm[0], Hm[1], Hm[2], fx, fy);
    g.p1 = hrow_dot(Hm[3], Hm[4], Hm[5], fx, fy);
    const float z = hrow_dot(Hm[6], Hm[7], Hm[8], fx, fy);
    const float r0 = hrow_dot(Ki[0], Ki[1], Ki[2], fx, fy);
    const float r1 = hrow_dot(Ki[3], Ki[4], Ki[5], fx, fy);
    const float r2 = hrow_dot(Ki[6], Ki[7], Ki[8], fx, fy);
    const float facing = facing_dot(r0, r1, r2, Rn[0], Rn[1], Rn[2]);
    mask = (facing > 0.0f) && (z > kZMin);
    g.z_clamped = (z < kZMin);
    g.zc = g.z_clamped ? kZMin : z;
    if (cn.fast) {  // uniform
      g.ix = normalise_roundtrip_rcp(g.p0 / g.zc, cn.Wm1, cn.rcpW);
      g.iy = normalise_roundtrip_rcp(g.p1 / g.zc, cn.Hm1, cn.rcpH);
    } else {
      g.ix = normalise_roundtrip(g.p0 / g.zc, cn.Wm1);
      g.iy = normalise_roundtrip(g.p1 / g.zc, cn.Hm1);
    }
  }
  return g;
}

// bilinear weight with which target sample (ix, iy) reaches source pixel (sx, sy): torch's own expressions
// (x1 - ix) / (ix - x0) for the tap that IS (sx, sy), zero when neither tap column / row is
__device__ __forceinline__ float tap_weight_on(float ix, float iy, int sx, int sy) {
  const float xf = floorf(ix), yf = floorf(iy);
  const float fsx = (float)sx, fsy = (float)sy;
  float wx = 0.0f, wy = 0.0f;
  if (xf == fsx) wx = (xf + 1.0f) - ix; else if (xf + 1.0f == fsx) wx = ix - xf;
  if (yf == fsy) wy = (yf + 1.0f) - iy; else if (yf + 1.0f == fsy) wy = iy - yf;
  return wx * wy;
}

__device__ __forceinline__ bool read_mask(const SweepArgs& a, int b, int n, int x, int y) {
  return a.padding_mask[(((long)b * a.N + n) * a.H + y) * a.W + x] != 0.0f;
}

}  // namespace pd
