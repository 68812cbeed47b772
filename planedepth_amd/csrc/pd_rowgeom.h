// Row geometry shared by the row-organised sweep kernels (pd_plane_sweep_rowshift.hip, pd_plane_sweep_rowstage.hip):
// the vertical footprint of a target row and the bit-exact horizontal sampling position of the reference.
#pragma once
#include "pd_sweep.h"

namespace pd {

// Vertical footprint of target row y (workgroup-uniform): up to two live source rows with their weights.
struct RowSel {
  int nrows;      // 0, 1 or 2 live rows
  int yA, yB;     // source rows (yB only if nrows == 2)
  float wA, wB;   // their bilinear weights
  float wy_main;  // weight of the workgroup's own row y (vertical adjoint)
};

__device__ __forceinline__ RowSel make_row_sel(int y, int H) {
  RowSel r;
  const float iy = normalise_roundtrip((float)y, (float)(H - 1));
  const float yf = floorf(iy);
  const float wy0 = (yf + 1.0f) - iy, wy1 = iy - yf;
  const int y0 = (int)yf;
  const bool use0 = (yf >= 0.0f) && (yf <= (float)(H - 1)) && (wy0 != 0.0f);
  const bool use1 = (yf + 1.0f >= 0.0f) && (yf + 1.0f <= (float)(H - 1)) && (wy1 != 0.0f);
  r.nrows = (int)use0 + (int)use1;
  r.yA = use0 ? y0 : y0 + 1;
  r.wA = use0 ? wy0 : (use1 ? wy1 : 0.0f);
  r.yB = y0 + 1;
  r.wB = wy1;
  if (r.nrows == 0) r.yA = min(max(y0, 0), H - 1);
  r.wy_main = (y0 == y) ? wy0 : ((y0 + 1 == y) ? wy1 : 0.0f);
  return r;
}

struct ColTap {    // horizontal footprint of one target pixel on one plane
  int x0;          // floor(ix)
  float w0, w1;    // torch's weights (x1 - ix), (ix - x0)
};

// ix = unnormalise(normalise(px)) of the reference, bit for bit, in 7 operations:
//   reference:  q = px/(W-1);  g = (q - 0.5)*2;            [trainer.py:550-552]
//               ix = ((g + 1)/2) * (W-1)                    [grid_sample, align_corners=True]
//   (g + 1)/2 = fl(2h + 1)/2 with h = fl(q - 0.5); scaling by 2 commutes with rounding, so it equals fl(h + 0.5).
// |px| <= 2W+2 by construction (the per-plane shift is clamped to +-(W+2) when it is staged), so floor(ix) converts
// to int without saturating and x0*4 cannot alias into the row.
__device__ __forceinline__ ColTap make_col_tap(float px, float Wm1, float rcpWm1) {
  ColTap t;
  float ix;
  {
#pragma clang fp contract(off)
    const float q = div_by(px, Wm1, rcpWm1);
    const float h = q - 0.5f;
    const float hh = h + 0.5f;
    ix = hh * Wm1;
  }
  const float xf = floorf(ix);
  t.w0 = (xf + 1.0f) - ix;
  t.w1 = ix - xf;
  t.x0 = (int)xf;
  return t;
}

}  // namespace pd
