// Row geometry shared by the row-organised sweep kernels (pd_plane_sweep_rowshift.hip, pd_plane_sweep_rowstage.hip):
// the vertical footprint of a target row and the bit-exact horizontal sampling position of the reference.
#pragma once
#include <math.h>

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

// ---- row pairs (forward: pd_rowshift_fwd.h has the full story; backward: pd_plane_sweep_rowstream.hip) -------------------
// A row y whose vertical round trip is inexact samples (1 - eps) * row y + eps * row p with p = y +- 1 ("leans" on p).  When p
// is exact, or leans back on y, one workgroup serves BOTH target rows from the two source rows it loads anyway, and the
// workgroup of p retires at once.  The rule is local (rows y-1 .. y+1), so every workgroup decides its own role without a table:
//   * y leans on p, p leans back on y  -> the lower of the two leads;
//   * y leans on p, p exact            -> y leads unless p-1 also leans on p and y = p+1 (the upper neighbour wins);
//   * y leans on p, p leans elsewhere  -> y stays a single two-source-row row (a chain; rare).
// At H = 192: 48 inexact rows -> 26 pairs, 10 left alone; H = 384: 94 -> 60 + 14.
enum PairRole { kSingle = 0, kLeader = 1, kAbsorbed = 2 };

__device__ __forceinline__ int row_lean(int y, int H) {  // 0: exact; +-1: direction of the second source row
  if (y < 0 || y >= H) return 0;
  const RowSel r = make_row_sel(y, H);
  if (r.nrows != 2) return 0;
  return (r.yA == y) ? +1 : -1;  // rows (y, y+1) or (y-1, y)
}
__device__ __forceinline__ bool leads(int y, int H) {  // y is inexact and takes its partner along
  const int l = row_lean(y, H);
  if (l == 0) return false;
  const int p = y + l;
  const int lp = row_lean(p, H);
  if (lp == -l) return y < p;                       // mutual
  if (lp != 0) return false;                        // chain
  if (l == -1) return row_lean(p - 1, H) != +1;     // p = y-1 is exact: its lower neighbour has the first call on it
  return true;
}
__device__ __forceinline__ PairRole pair_role(int y, int H, int& partner) {
  partner = y;
  const int l = row_lean(y, H);
  if (l != 0) {
    partner = y + l;
    if (leads(y, H)) return kLeader;
    return (row_lean(partner, H) == -l && leads(partner, H)) ? kAbsorbed : kSingle;  // mutual: the other one leads
  }
  if (row_lean(y - 1, H) == +1 && leads(y - 1, H)) { partner = y - 1; return kAbsorbed; }
  if (row_lean(y + 1, H) == -1 && leads(y + 1, H)) { partner = y + 1; return kAbsorbed; }
  return kSingle;
}

// Weights of the pair (leader y, partner p) on the two source rows: target y = a0*R_y + b0*R_p, target p = a1*R_p + b1*R_y
struct PairW {
  float a0, b0, a1, b1;
};
__device__ __forceinline__ PairW pair_weights(int y, int p, int H) {
  PairW w;
  const RowSel ry = make_row_sel(y, H), rp = make_row_sel(p, H);
  w.a0 = (ry.yA == y) ? ry.wA : ry.wB;
  w.b0 = (ry.yA == y) ? ry.wB : ry.wA;
  if (rp.nrows == 2) {  // mutual lean
    w.a1 = (rp.yA == p) ? rp.wA : rp.wB;
    w.b1 = (rp.yA == p) ? rp.wB : rp.wA;
  } else {
    w.a1 = rp.wA;  // exact row: 1
    w.b1 = 0.0f;
  }
  return w;
}

// ---- host mirrors (launch planning only: which rows go together, in which order) ---------------------------------------
// make_row_sel / row_lean / pair_role on the host, operation by operation in fp32 (volatile: no contraction, no excess
// precision).  The kernels never TRUST these for correctness: a unit table built from them says which rows a workgroup serves,
// and the device re-derives every footprint itself (a pair the device's own arithmetic does not confirm is served row by row).
struct HostRowSel { int nrows, yA, yB; float wA, wB, wy_main; };
inline HostRowSel host_row_sel(int y, int H) {
  volatile float hm1 = (float)(H - 1);
  volatile float q = (float)y / hm1;
  volatile float h = q - 0.5f;
  volatile float g = h * 2.0f;
  volatile float sv = g + 1.0f;
  volatile float hh = sv * 0.5f;
  volatile float iy = hh * hm1;
  const float yf = floorf(iy);
  volatile float yf1 = yf + 1.0f;
  volatile float wy0 = yf1 - iy, wy1 = iy - yf;
  const bool use0 = yf >= 0.0f && yf <= hm1 && wy0 != 0.0f, use1 = yf1 >= 0.0f && yf1 <= hm1 && wy1 != 0.0f;
  const int y0 = (int)yf;
  HostRowSel r;
  r.nrows = (int)use0 + (int)use1;
  r.yA = use0 ? y0 : y0 + 1;
  r.wA = use0 ? wy0 : (use1 ? wy1 : 0.0f);
  r.yB = y0 + 1;
  r.wB = wy1;
  if (r.nrows == 0) r.yA = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
  r.wy_main = (y0 == y) ? wy0 : ((y0 + 1 == y) ? wy1 : 0.0f);
  return r;
}
inline int host_row_lean(int y, int H) {
  if (y < 0 || y >= H) return 0;
  const HostRowSel r = host_row_sel(y, H);
  if (r.nrows != 2) return 0;
  return (r.yA == y) ? +1 : -1;
}
inline bool host_leads(int y, int H) {
  const int l = host_row_lean(y, H);
  if (l == 0) return false;
  const int p = y + l;
  const int lp = host_row_lean(p, H);
  if (lp == -l) return y < p;
  if (lp != 0) return false;
  if (l == -1) return host_row_lean(p - 1, H) != +1;
  return true;
}
// 0 single, 1 leader (partner = the row it takes along), 2 absorbed (partner = its leader)
inline int host_pair_role(int y, int H, int& partner) {
  partner = y;
  const int l = host_row_lean(y, H);
  if (l != 0) {
    partner = y + l;
    if (host_leads(y, H)) return 1;
    return (host_row_lean(partner, H) == -l && host_leads(partner, H)) ? 2 : 0;
  }
  if (host_row_lean(y - 1, H) == +1 && host_leads(y - 1, H)) { partner = y - 1; return 2; }
  if (host_row_lean(y + 1, H) == -1 && host_leads(y + 1, H)) { partner = y + 1; return 2; }
  return 0;
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
