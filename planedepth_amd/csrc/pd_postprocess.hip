// Warps of the self-distillation post-process, Trainer.generate_post_process_disp (reference trainer.py:421-466;
// SURVEY.md §8f rank 2).  The reference builds two [B*N,H,W,2] grids, calls F.grid_sample five times on [B*N,1,H,W]
// tensors, two softmaxes over the planes and three sum/clamp passes.  Both shapes it needs are fused here:
//   pd_warp_softmax   out[b,n,y,x] = softmax_n( planes[b,n] sampled at (x + s*d[b,n], y) )       (trainer.py:443-446, 451-453)
//   pd_warp_sum       out[b,0,y,x] = min(cap, sum_n planes[b,n] sampled at (x + s*d[b,n], y) )   (:447-449, 454-456, 463-465)
// with the reference's pixel -> [-1,1] -> pixel round trip (trainer.py:427-441 + grid_sample, align_corners=True,
// zeros padding) reproduced through pd_common.h's normalise_roundtrip, and PD_PP_FLIP_SRC reading the source planes
// mirrored along x (the `.flip(-1)` of trainer.py:451) without materialising the flipped tensor.  Forward only: the
// reference runs this under the fixed (no-grad) networks and detaches the result (:466).
#include <math.h>
#include <stdlib.h>
#include <utility>
#include "pd_common.h"
#include "pd_rowshift_common.h"

namespace pd {

struct WarpArgs {
  int N, H, W;
  int dense, flip, rows;   // rows: disp is [B,N,H] — one disparity per (plane, row): xz planes (PD_PP_DISP_ROWS)
  int dimg, dplane, ymask; // disparity of (b, n, y) when !dense: disp[b * dimg + n * dplane + (y & ymask)] — (N, 1, 0) per plane,
                           // (N H, H, ~0) per row: 32-bit, branch-free (a select between two 64-bit addresses per plane cost the
                           // unrolled segment kernels 90 scalar instructions a plane)
  float sign;
  float Wm1, rcpWm1;   // W - 1 and its correctly rounded reciprocal (host-computed: 1/(W-1) in double, rounded once)
  const float* planes;
  const float* disp;
};

typedef float v2f_u4 __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte load at 4-byte alignment

// These kernels are latency-bound gathers (no reuse, one thread per pixel), so the loop is shaped for loads in flight:
// everything that depends on the row only is hoisted (the vertical taps are the same for all planes), the two column
// taps of a row come as ONE 8-byte load from a clamped, always-valid position with the out-of-image cases folded into
// the weights (no branches around loads), planes go in groups of kGroup with all loads issued before the first use,
// and a wave whose pixels all have a zero second vertical weight (three rows of four: the reference's
// y -> [-1,1] -> y round trip is exact there) runs a one-row loop that never reads the second row.  (With finite
// inputs skipping a zero-weighted row gives the same number.)
constexpr int kGroup = 4;

struct RowTaps {        // per pixel, plane-independent
  int y;                // the target row itself
  int ra, rb;           // clamped source rows
  float wa, wb;         // their weights (0 when the row is outside the image)
};
__device__ __forceinline__ RowTaps row_taps(int y, int H) {
  const float iy = normalise_roundtrip((float)y, (float)(H - 1));
  const float yf = floorf(iy), yf1 = yf + 1.0f;
  const bool va = (yf >= 0.0f) && (yf <= (float)(H - 1)), vb = (yf1 >= 0.0f) && (yf1 <= (float)(H - 1));
  const int y0 = (int)fminf(fmaxf(yf, -2.0f), (float)H);
  RowTaps r;
  r.y = y;
  r.ra = min(max(y0, 0), H - 1);
  r.rb = min(max(y0 + 1, 0), H - 1);
  r.wa = va ? yf1 - iy : 0.0f;
  r.wb = vb ? iy - yf : 0.0f;
  return r;
}

struct ColPair {        // per pixel and plane: where to load the column pair and how to weight its two values
  int off;              // first column of the pair, in [0, W-2]
  float e0, e1;
};
template <bool FLIP>
__device__ __forceinline__ ColPair col_pair(float ix, int W) {
  const float xf = floorf(ix), xf1 = xf + 1.0f;
  const float wx0 = xf1 - ix, wx1 = ix - xf;
  const bool v0 = (xf >= 0.0f) && (xf <= (float)(W - 1)), v1 = (xf1 >= 0.0f) && (xf1 <= (float)(W - 1));
  const int x0 = (int)fminf(fmaxf(xf, -2.0f), (float)W);
  const int c = min(max(x0, 0), W - 2);
  const float g0 = v0 ? wx0 : 0.0f, g1 = v1 ? wx1 : 0.0f;   // weights of columns x0 and x0+1
  // the loaded pair is columns (c, c+1); x0 == c in the interior, c-1 at the left border, c+1 at the right one
  const float e0 = (x0 == c) ? g0 : ((x0 == c - 1) ? g1 : 0.0f);
  const float e1 = (x0 == c) ? g1 : ((x0 == c + 1) ? g0 : 0.0f);
  ColPair p;
  if (FLIP) { p.off = W - 2 - c; p.e0 = e1; p.e1 = e0; }    // mirrored columns: the pair is read in reverse order
  else      { p.off = c;         p.e0 = e0; p.e1 = e1; }
  return p;
}

__device__ __forceinline__ float plane_ix(const WarpArgs& a, int b, int n, int x, int y) {
  const float d = a.dense ? a.disp[(((long)b * a.N + n) * a.H + y) * a.W + x]
                          : a.disp[b * a.dimg + n * a.dplane + (y & a.ymask)];
  // the division by W-1 through its refined reciprocal: the same bits as IEEE division (pd_common.h), a third of the cost
  return normalise_roundtrip_rcp((float)x + a.sign * d, a.Wm1, a.rcpWm1);
}

// Samples of planes n0 .. n0+U-1 at this pixel (all loads first).
template <bool FLIP, int NR, int U>
__device__ __forceinline__ void sample_group(const WarpArgs& a, const float* __restrict__ pb, const RowTaps& r, int b,
                                             int n0, int x, int y, int HW, float (&out)[U]) {
  ColPair cp[U];
  v2f_u4 va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    cp[u] = col_pair<FLIP>(plane_ix(a, b, n0 + u, x, y), a.W);
    const float* pl = pb + (long)(n0 + u) * HW;
    va[u] = *reinterpret_cast<const v2f_u4*>(pl + (long)r.ra * a.W + cp[u].off);
    if (NR == 2) vb[u] = *reinterpret_cast<const v2f_u4*>(pl + (long)r.rb * a.W + cp[u].off);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float v = (va[u].x * cp[u].e0 + va[u].y * cp[u].e1) * r.wa;
    if (NR == 2) v += (vb[u].x * cp[u].e0 + vb[u].y * cp[u].e1) * r.wb;
    out[u] = v;
  }
}

// f(n, value) for every plane of this pixel, in plane order
template <bool FLIP, int NR, typename F>
__device__ __forceinline__ void for_each_sample(const WarpArgs& a, const float* __restrict__ pb, const RowTaps& r, int b,
                                                int x, int y, int HW, F f) {
  int n = 0;
  for (; n + kGroup <= a.N; n += kGroup) {
    float v[kGroup];
    sample_group<FLIP, NR, kGroup>(a, pb, r, b, n, x, y, HW, v);
#pragma unroll
    for (int u = 0; u < kGroup; ++u) f(n + u, v[u]);
  }
  for (; n < a.N; ++n) {
    float v[1];
    sample_group<FLIP, NR, 1>(a, pb, r, b, n, x, y, HW, v);
    f(n, v[0]);
  }
}

// Softmax over the planes of the warped logits.  Pass 1 finds max and sum, pass 2 samples again and writes the
// probabilities: N writes and (cache permitting) N reads per pixel.  Measured at 4x49x192x640: 92-100 us; parking the
// sampled logits in `out` between the passes (2N writes + 2N reads) 118 us; parking them in LDS (128-thread workgroups,
// 25 KB each) 120 us — the kernel is VALU-bound (coordinate chain + tap set-up per plane), LDS costs it occupancy.
template <bool FLIP>
__global__ __launch_bounds__(kBlock) void warp_softmax_kernel(WarpArgs a, float* __restrict__ out) {
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  const bool active = pix < HW;
  const int y = active ? pix / a.W : 0, x = active ? pix - y * a.W : 0;
  const RowTaps r = row_taps(y, a.H);
  const bool two_rows = __any(active && r.wb != 0.0f);   // wave-uniform
  if (!active) return;
  const float* pb = a.planes + (long)b * a.N * HW;
  float* ob = out + (long)b * a.N * HW + pix;
  float m = -INFINITY, Z = 0.0f;
  auto stat = [&](int, float l) {
    if (l > m) { Z *= __expf(m - l); m = l; }
    Z += __expf(l - m);
  };
  if (two_rows) for_each_sample<FLIP, 2>(a, pb, r, b, x, y, HW, stat);
  else          for_each_sample<FLIP, 1>(a, pb, r, b, x, y, HW, stat);
  const float invZ = 1.0f / Z;
  auto emit = [&](int n, float l) { ob[(long)n * HW] = __expf(l - m) * invZ; };
  if (two_rows) for_each_sample<FLIP, 2>(a, pb, r, b, x, y, HW, emit);
  else          for_each_sample<FLIP, 1>(a, pb, r, b, x, y, HW, emit);
}

template <bool FLIP>
__global__ __launch_bounds__(kBlock) void warp_sum_kernel(WarpArgs a, float cap, float* __restrict__ out) {
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  const bool active = pix < HW;
  const int y = active ? pix / a.W : 0, x = active ? pix - y * a.W : 0;
  const RowTaps r = row_taps(y, a.H);
  const bool two_rows = __any(active && r.wb != 0.0f);
  if (!active) return;
  const float* pb = a.planes + (long)b * a.N * HW;
  float acc = 0.0f;
  auto add = [&](int, float v) { acc += v; };
  if (two_rows) for_each_sample<FLIP, 2>(a, pb, r, b, x, y, HW, add);
  else          for_each_sample<FLIP, 1>(a, pb, r, b, x, y, HW, add);
  out[(long)b * HW + pix] = fminf(acc, cap);   // o[o > 1] = 1
}

// ---------------------------------------------------------------------------------------------------------------
// Per-plane scalar disparities (the decoder's expanded [B,N,1,1] levels, the usual case): along a target row every pixel
// samples plane n at x + s*d_n, so the row-shift machinery of the sweep applies — one wave per 64-pixel segment of a
// row, the vertical taps are wave-uniform scalars, each (plane, row) gets a buffer descriptor whose hardware range
// check IS padding_mode="zeros", the two column taps come in one 8-byte load (column -1 fixed up through the
// weights), and the sampling position costs make_col_tap's 7 operations instead of the general chain — half the
// instructions of the gather kernels above in a VALU-bound loop.  FLIP reads the pair at the mirrored columns.
// ---------------------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t PRsrc;
typedef float v2f_b __attribute__((ext_vector_type(2)));
__device__ __forceinline__ PRsrc pp_row_rsrc(const float* row, int W) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, W * 4, 0x00020000);
}
__device__ __forceinline__ v2f_b pp_load2(PRsrc r, unsigned byte_off) {
  return __builtin_bit_cast(v2f_b, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}

struct RowPair {   // where the pair load goes and how its two dwords are weighted
  unsigned off;
  float e0, e1;
};
template <bool FLIP>
__device__ __forceinline__ RowPair row_pair(const ColTap& t, int W) {
  // columns x0 / x0+1 carry weights w0 / w1; mirrored, they are columns W-1-x0 / W-2-x0, read as the pair at W-2-x0
  const int c = FLIP ? W - 2 - t.x0 : t.x0;
  const bool edge = (c == -1);   // the pair starts one column left of the row: load at column 0, its first dword is the second tap
  RowPair p;
  p.off = edge ? 0u : ((unsigned)c << 2);   // any other out-of-row position is range-checked to zeros by the hardware
  const float first = FLIP ? t.w1 : t.w0, second = FLIP ? t.w0 : t.w1;
  p.e0 = edge ? second : first;
  p.e1 = edge ? 0.0f : second;
  return p;
}

// the shift s*d_n, clamped like the sweep's staged shifts (beyond +-(W+1) nothing is in view; keeps x0*4 from aliasing)
__device__ __forceinline__ float plane_shift(const WarpArgs& a, int b, int n, int y) {
  const float sd = a.sign * a.disp[b * a.dimg + n * a.dplane + (y & a.ymask)], lim = (float)(a.W + 2);
  return (sd >= -lim && sd <= lim) ? sd : ((sd < 0.0f) ? -lim : lim);   // NaN -> +lim
}

template <bool FLIP, int NR, int U>
__device__ __forceinline__ void rows_group(const WarpArgs& a, const float* __restrict__ pb, const RowTaps& r, int b,
                                           int n0, int x, int HW, float (&out)[U]) {
  RowPair rp[U];
  v2f_b va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    rp[u] = row_pair<FLIP>(make_col_tap((float)x + plane_shift(a, b, n0 + u, r.y), a.Wm1, a.rcpWm1), a.W);
    const float* pl = pb + (long)(n0 + u) * HW;   // wave-uniform
    va[u] = pp_load2(pp_row_rsrc(pl + (long)r.ra * a.W, a.W), rp[u].off);
    if (NR == 2) vb[u] = pp_load2(pp_row_rsrc(pl + (long)r.rb * a.W, a.W), rp[u].off);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float v = (va[u].x * rp[u].e0 + va[u].y * rp[u].e1) * r.wa;
    if (NR == 2) v += (vb[u].x * rp[u].e0 + vb[u].y * rp[u].e1) * r.wb;
    out[u] = v;
  }
}

template <bool FLIP, int NR, typename F>
__device__ __forceinline__ void rows_for_each(const WarpArgs& a, const float* __restrict__ pb, const RowTaps& r, int b,
                                              int x, int HW, F f) {
  int n = 0;
  for (; n + kGroup <= a.N; n += kGroup) {
    float v[kGroup];
    rows_group<FLIP, NR, kGroup>(a, pb, r, b, n, x, HW, v);
#pragma unroll
    for (int u = 0; u < kGroup; ++u) f(n + u, v[u]);
  }
  for (; n < a.N; ++n) {
    float v[1];
    rows_group<FLIP, NR, 1>(a, pb, r, b, n, x, HW, v);
    f(n, v[0]);
  }
}

// grid (segments of 64 pixels, H, B), one wave per workgroup
template <bool FLIP>
__global__ __launch_bounds__(kWave) void warp_softmax_rows_kernel(WarpArgs a, float* __restrict__ out) {
  const int HW = a.H * a.W, y = blockIdx.y, b = blockIdx.z;
  const int x = blockIdx.x * kWave + threadIdx.x;
  const RowTaps r = row_taps(y, a.H);   // wave-uniform
  const float* pb = a.planes + (long)b * a.N * HW;
  const bool active = x < a.W;
  float* ob = out + (long)b * a.N * HW + (long)y * a.W + (active ? x : 0);
  float m = -INFINITY, Z = 0.0f;
  auto stat = [&](int, float l) {
    if (l > m) { Z *= __expf(m - l); m = l; }
    Z += __expf(l - m);
  };
  if (r.wb != 0.0f) rows_for_each<FLIP, 2>(a, pb, r, b, x, HW, stat);
  else              rows_for_each<FLIP, 1>(a, pb, r, b, x, HW, stat);
  const float invZ = 1.0f / Z;
  auto emit = [&](int n, float l) { if (active) ob[(long)n * HW] = __expf(l - m) * invZ; };
  if (r.wb != 0.0f) rows_for_each<FLIP, 2>(a, pb, r, b, x, HW, emit);
  else              rows_for_each<FLIP, 1>(a, pb, r, b, x, HW, emit);
}

template <bool FLIP>
__global__ __launch_bounds__(kWave) void warp_sum_rows_kernel(WarpArgs a, float cap, float* __restrict__ out) {
  const int HW = a.H * a.W, y = blockIdx.y, b = blockIdx.z;
  const int x = blockIdx.x * kWave + threadIdx.x;
  const RowTaps r = row_taps(y, a.H);
  const float* pb = a.planes + (long)b * a.N * HW;
  float acc = 0.0f;
  auto add = [&](int, float v) { acc += v; };
  if (r.wb != 0.0f) rows_for_each<FLIP, 2>(a, pb, r, b, x, HW, add);
  else              rows_for_each<FLIP, 1>(a, pb, r, b, x, HW, add);
  if (x < a.W) out[(long)b * HW + (long)y * a.W + x] = fminf(acc, cap);
}

// ---------------------------------------------------------------------------------------------------------------
// Segment form of the two (round 6): the segment-stream forward's access shape (pd_plane_sweep_fwdstream.hip).  One wave owns a
// 128-pixel segment of a target row, a lane two adjacent pixels; their taps on plane n are the three source values at
// x + k .. x + k + 2, k = floor(s d_n): ONE 12-byte load per live source row (half the memory instructions per pixel of the
// 8-byte form above), coordinates through stream_ix, and — what the softmax gains most from — the sampled logits of ALL planes
// stay in registers (2 x N <= 128 VGPRs), so the planes are sampled ONCE: the kernels above sample twice (statistics, then
// probabilities).  FLIP reads the same three columns mirrored: the taps of pixels x, x+1 at mirrored columns c, c+1, c+2 are the
// actual columns W-1-c .. W-3-c, one 12-byte load at W-3-c in reverse order.  Exactness as in the forward: planes whose
// frac(s d) is within irregular_tol of an integer take the per-pixel path of the row kernels above (exact floor(ix), 8-byte
// pairs).  A lane whose load would START at column -2 or -1 (such a load reads as zeros as a whole although its last columns are
// inside the row; at -3 its last dword passes the 32-bit range check and reads the neighbouring row) aims it at column 0 and
// shifts the three values into place (seg_load) — no per-segment fallback for negative shifts.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSegPix = 2 * kWave;   // pixels per wave
constexpr int kSegWaves = 4;         // waves per workgroup (independent items)
typedef float v3f_pp __attribute__((ext_vector_type(3)));
__device__ __forceinline__ v3f_pp pp_load3(PRsrc r, unsigned byte_off) {
  return __builtin_bit_cast(v3f_pp, __builtin_amdgcn_raw_buffer_load_b96(r, (int)byte_off, 0, 0));
}

struct SegPlane {   // wave-uniform: one plane's shift for this segment
  float sd;
  int k;
  bool general;
};
// The shifts of planes nbase .. nbase + 63, one per lane (plane_shift's value: sign, clamp): ONE vector load per wave and 64 planes,
// whatever the disparities' layout; a plane's shift is then a v_readlane away.  (Scalar loads per plane were fine while the
// per-plane layout let the compiler fold n into the load's immediate and fetch sixteen planes at a time; with a run-time stride
// every plane waited for its own s_load: +30-50 % on the chain kernels.)
__device__ __forceinline__ float seg_shifts(const WarpArgs& a, int b, int y, int nbase) {
  const int n = min(nbase + (int)(threadIdx.x & (kWave - 1)), a.N - 1);
  return plane_shift(a, b, n, y);
}
__device__ __forceinline__ float shift_of(float shifts, int j) {   // j: wave-uniform lane index
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(shifts), j));
}
template <bool FLIP>
__device__ __forceinline__ SegPlane seg_plane(const WarpArgs& a, float sd) {
  SegPlane p;
  p.sd = sd;
  const float fl = floorf(p.sd), fr = p.sd - fl;
  p.k = (int)fl;
  const float tol = irregular_tol(a.W);
  const bool inview = fabsf(p.sd) < (float)(a.W + 1);
  const bool irr = inview && (fr < tol || fr > 1.0f - tol);
  p.general = irr;   // (a load that would start left of the row is re-aimed per lane: seg_load)
  return p;
}

// Where a lane's 12-byte load goes when its three columns start at `start` (may be negative): columns left of the row are zero
// padding, but the hardware drops a load that STARTS there as a whole (and lets the dword at byte offset -4 through).  start in
// {-2, -1}: load columns 0 .. 2 and shift them right by adj = -start when the values are used; start <= -3: nothing of it is
// inside the row — an offset beyond every row (the range check returns zeros).
struct SegAim { unsigned off; int adj; };
__device__ __forceinline__ SegAim seg_aim(int start) {
  SegAim m;
  m.adj = (start < 0 && start >= -2) ? -start : 0;
  m.off = (start <= -3) ? 0x7FFFFFF0u : ((unsigned)(start + m.adj) << 2);
  return m;
}
__device__ __forceinline__ v3f_pp seg_place(const v3f_pp& t, int adj) {
  v3f_pp o;
  o.x = (adj == 0) ? t.x : 0.0f;
  o.y = (adj == 0) ? t.y : ((adj == 1) ? t.x : 0.0f);
  o.z = (adj == 0) ? t.z : ((adj == 1) ? t.y : t.x);
  return o;
}

// the two samples of a lane on one plane from its loaded taps (A = first live row, B = second)
template <bool FLIP, int NR>
__device__ __forceinline__ void seg_values(const WarpArgs& a, const RowTaps& r, const SegPlane& p, float x0f, const v3f_pp& ta,
                                           const v3f_pp& tb, float& v0, float& v1) {
  const float kf = (float)p.k;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float xtf = x0f + (float)i, xsf = xtf + kf;          // integers below 2^24: exact
    const float ix = stream_ix(xtf, p.sd, a.Wm1, a.rcpWm1);
    const float w1 = ix - xsf, w0 = (xsf + 1.0f) - ix;         // torch's (ix - x0), (x1 - ix) with x0 = x + k
    // plain: pixel i reads taps i (w0), i + 1 (w1); mirrored: the three columns arrive reversed
    const float a0 = FLIP ? (i == 0 ? ta.z : ta.y) : (i == 0 ? ta.x : ta.y), a1 = FLIP ? (i == 0 ? ta.y : ta.x) : (i == 0 ? ta.y : ta.z);
    float v = (a0 * w0 + a1 * w1) * r.wa;
    if (NR == 2) {
      const float b0 = FLIP ? (i == 0 ? tb.z : tb.y) : (i == 0 ? tb.x : tb.y), b1 = FLIP ? (i == 0 ? tb.y : tb.x) : (i == 0 ? tb.y : tb.z);
      v += (b0 * w0 + b1 * w1) * r.wb;
    }
    if (i == 0) v0 = v; else v1 = v;
  }
}

// f(n, v0, v1) for every plane of the lane's two pixels, in plane order; loads of kGroup planes in flight
template <bool FLIP, int NR, int G, typename F>
__device__ __forceinline__ void seg_group(const WarpArgs& a, const float* __restrict__ pb, const RowTaps& r, int b, int xseg,
                                          int x0, int HW, int n0, F& f, float shifts, int sbase, int nbase = 0, int nend = 1 << 30) {
  // planes nbase + n0 .. nbase + n0 + G - 1 (those below min(nend, N)); the callback sees the index RELATIVE to nbase, which is a
  // compile-time constant where the caller's loop is unrolled
  const float x0f = (float)x0;
  const int nlim = min(nend, a.N);
  {
    SegPlane sp[G];
    v3f_pp ta[G], tb[G];
    int adj[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int n = min(nbase + n0 + u, a.N - 1);
      sp[u] = seg_plane<FLIP>(a, shift_of(shifts, n - sbase));
      const float* pl = pb + (long)n * HW;
      const SegAim aim = seg_aim(FLIP ? a.W - 3 - (x0 + sp[u].k) : x0 + sp[u].k);
      adj[u] = aim.adj;
      ta[u] = pp_load3(pp_row_rsrc(pl + (long)r.ra * a.W, a.W), aim.off);
      if (NR == 2) tb[u] = pp_load3(pp_row_rsrc(pl + (long)r.rb * a.W, a.W), aim.off);
      else tb[u] = ta[u];
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int n = nbase + n0 + u;
      if (n < nlim) {   // (uniform)
      float v0, v1;
      if (sp[u].general) {   // (uniform) exact per-pixel path
        float o0[1], o1[1];
        rows_group<FLIP, NR, 1>(a, pb, r, b, n, x0, HW, o0);
        rows_group<FLIP, NR, 1>(a, pb, r, b, n, x0 + 1, HW, o1);
        v0 = o0[0]; v1 = o1[0];
      } else {
        seg_values<FLIP, NR>(a, r, sp[u], x0f, seg_place(ta[u], adj[u]), seg_place(tb[u], adj[u]), v0, v1);
      }
      f(n0 + u, v0, v1);
      }
    }
  }
}
// NMAX > 0: the plane loop fully unrolled over NMAX >= N planes, group by group through an index pack (the callback then sees
// compile-time plane indices and its per-plane state stays in registers; a `break` in a pragma-unrolled loop left it in scratch);
// NMAX == 0: a run-time loop
template <bool FLIP, int NR, int G, typename F, int... I>
__device__ __forceinline__ void seg_groups(const WarpArgs& a, const float* __restrict__ pb, const RowTaps& r, int b, int xseg,
                                           int x0, int HW, F& f, std::integer_sequence<int, I...>, int nbase, int nend, float shifts) {
  ((nbase + I * G < min(nend, a.N) ? seg_group<FLIP, NR, G>(a, pb, r, b, xseg, x0, HW, I * G, f, shifts, 0, nbase, nend) : (void)0), ...);
}
template <bool FLIP, int NR, int NMAX, typename F, int G = kGroup>   // G planes' loads in flight per wave
__device__ __forceinline__ void seg_for_each(const WarpArgs& a, const float* __restrict__ pb, const RowTaps& r, int b, int xseg,
                                             int x0, int HW, F f, int nbase = 0, int nend = 1 << 30) {
  if constexpr (NMAX > 0) {   // (at most 64 planes: one register of shifts)
    const float shifts = seg_shifts(a, b, r.y, 0);
    seg_groups<FLIP, NR, G>(a, pb, r, b, xseg, x0, HW, f, std::make_integer_sequence<int, (NMAX + G - 1) / G>{}, nbase, nend, shifts);
  } else {
    float shifts = 0.0f;
    for (int n0 = 0; n0 < a.N; n0 += G) {
      if ((n0 & (kWave - 1)) == 0) shifts = seg_shifts(a, b, r.y, n0);   // (G divides 64)
      seg_group<FLIP, NR, G>(a, pb, r, b, xseg, x0, HW, n0, f, shifts, n0 & ~(kWave - 1));
    }
  }
}

// which (image, row, segment) a wave serves: items row-major, kSegWaves consecutive ones per workgroup
struct SegItem { int b, y, xseg; bool on; };
__device__ __forceinline__ SegItem seg_item(const WarpArgs& a, int B) {
  const int nseg = (a.W + kSegPix - 1) / kSegPix;
  const long item = (long)blockIdx.x * kSegWaves + (threadIdx.x >> 6);
  SegItem s;
  s.on = item < (long)B * a.H * nseg;
  const long it = s.on ? item : 0;
  s.xseg = (int)(it % nseg) * kSegPix;
  s.y = (int)((it / nseg) % a.H);
  s.b = (int)(it / ((long)nseg * a.H));
  s.xseg = __builtin_amdgcn_readfirstlane(s.xseg); s.y = __builtin_amdgcn_readfirstlane(s.y); s.b = __builtin_amdgcn_readfirstlane(s.b);
  return s;
}

template <bool FLIP, int NR, int NMAX>
__device__ __forceinline__ void seg_softmax_body(const WarpArgs& a, const SegItem& it, const RowTaps& r, float* __restrict__ out) {
  const int HW = a.H * a.W, lane = threadIdx.x & (kWave - 1);
  const int x0 = it.xseg + 2 * lane;
  const float* pb = a.planes + (long)it.b * a.N * HW;
  float l0[NMAX], l1[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) l0[n] = l1[n] = -INFINITY;
  float m0 = -INFINITY, m1 = -INFINITY;
  seg_for_each<FLIP, NR, NMAX>(a, pb, r, it.b, it.xseg, x0, HW, [&](int n, float v0, float v1) {
#pragma unroll
    for (int q = 0; q < NMAX; ++q) if (q == n) { l0[q] = v0; l1[q] = v1; }   // (n is a compile-time constant once unrolled)
    m0 = fmaxf(m0, v0); m1 = fmaxf(m1, v1);
  });
  float Z0 = 0.0f, Z1 = 0.0f;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {   // (planes beyond N hold -inf: they add exp(-inf) = 0)
    l0[n] = __expf(l0[n] - m0); l1[n] = __expf(l1[n] - m1);
    Z0 += l0[n]; Z1 += l1[n];
  }
  if (x0 >= a.W) return;   // (W is even: the lane's two pixels are inside or outside together)
  const float i0 = 1.0f / Z0, i1 = 1.0f / Z1;
  float* ob = out + (long)it.b * a.N * HW + (long)it.y * a.W + x0;
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
    if (n < a.N) *reinterpret_cast<float2*>(ob + (long)n * HW) = make_float2(l0[n] * i0, l1[n] * i1);
}

// (NMAX = 64: 128 registers of samples; bounded to three waves per SIMD — unbounded the allocator takes 206 VGPRs = two waves,
// and the kernel is no faster than the row form)
template <bool FLIP, int NMAX>
__global__ __launch_bounds__(kSegWaves* kWave, NMAX > 32 ? 3 : 4) void warp_softmax_seg_kernel(WarpArgs a, int B, float* __restrict__ out) {
  const SegItem it = seg_item(a, B);
  if (!it.on) return;
  const RowTaps r = row_taps(it.y, a.H);   // wave-uniform
  if (r.wb != 0.0f) seg_softmax_body<FLIP, 2, NMAX>(a, it, r, out);
  else              seg_softmax_body<FLIP, 1, NMAX>(a, it, r, out);
}

template <bool FLIP>
__global__ __launch_bounds__(kSegWaves* kWave) void warp_sum_seg_kernel(WarpArgs a, int B, float cap, float* __restrict__ out) {
  const SegItem it = seg_item(a, B);
  if (!it.on) return;
  const RowTaps r = row_taps(it.y, a.H);
  const int HW = a.H * a.W, lane = threadIdx.x & (kWave - 1);
  const int x0 = it.xseg + 2 * lane;
  const float* pb = a.planes + (long)it.b * a.N * HW;
  float acc0 = 0.0f, acc1 = 0.0f;
  auto add = [&](int, float v0, float v1) { acc0 += v0; acc1 += v1; };
  if (r.wb != 0.0f) seg_for_each<FLIP, 2, 0>(a, pb, r, it.b, it.xseg, x0, HW, add);
  else              seg_for_each<FLIP, 1, 0>(a, pb, r, it.b, it.xseg, x0, HW, add);
  if (x0 < a.W) *reinterpret_cast<float2*>(out + (long)it.b * HW + (long)it.y * a.W + x0) = make_float2(fminf(acc0, cap), fminf(acc1, cap));
}

// ---------------------------------------------------------------------------------------------------------------
// Row chains (round 6): softmax(warp(logits)) never leaves the CU.  trainer.py:443-449 / 451-456 warp the logits, take the softmax
// over the planes, warp THAT again and sum over the planes; with per-plane disparities both warps are horizontal, so the second
// one samples, for a target row, the softmax's rows ya / yb only — and its horizontal part does not depend on the target row:
//   S_r(x) = sum_n hinterp(softmax_row_r[n], x + s2 d2_n),        o(y, x) = min(1, wa_y S_ya(x) + wb_y S_yb(x))
// (the vertical weights factor out of the plane sum; fp32 reassociation only).  One workgroup per (chain, image, row r): the
// segment form samples the row's logits (all planes of a pixel pair in registers), writes the normalised probabilities into an
// LDS row buffer [N][W + 4] (125 KB at 49 x 640: one workgroup per CU) and then takes S_r from LDS with the exact per-pixel taps
// (make_col_tap; zero guard cells are the padding).  The [B,N,H,W] intermediate of the two-kernel form — written and read back,
// twice per call: 4 N of the path's 7 N planes of traffic — is gone; what remains are S rows, and pp_rows_finish_kernel applies the
// vertical weights, the clamp and the disp_pp blend.
// ---------------------------------------------------------------------------------------------------------------
#ifndef PD_PP_CHAIN_G
#define PD_PP_CHAIN_G 4
#endif
#ifndef PD_PP_CHAIN_SPLIT
#define PD_PP_CHAIN_SPLIT 1   // 0: one wave per segment (A/B)
#endif
struct ChainArgs {
  WarpArgs w[2];          // first warp of chain 0 (image, plain) / chain 1 (mirrored image, PD_PP_FLIP_SRC)
  const float* disp2[2];  // the second warp's disparities [B,N] ([B,N,H] with rows2) and sign
  float sign2[2];
  float* S[2];            // [3][B,1,H,W] each: S_r with the shifts of target row r, r - 1, r + 1 (the last two with rows2 only,
  int B, rows2;           // and only where that neighbour blends row r in)
};
// Which targets need S of source row r: r itself always; with per-row shifts also the neighbours t = r -+ 1 whose vertical taps
// reach r (their shift differs from row r's) — the inexact rows of the y round trip, a quarter of them at H = 192.
__device__ __forceinline__ bool chain_needs(int r, int t, int H) {
  if (t < 0 || t >= H) return false;
  const RowTaps q = row_taps(t, H);
  return (q.ra == r && q.wa != 0.0f) || (q.rb == r && q.wb != 0.0f);
}
// the second warp's shifts of planes 0 .. 63 under target row yt, one per lane (plane_shift's clamp)
__device__ __forceinline__ float chain_shifts(const float* __restrict__ disp2, float sign2, int rows2, int b, int N, int H, int yt,
                                              float lim) {
  const int n = min((int)(threadIdx.x & (kWave - 1)), N - 1);
  const float sdr = sign2 * disp2[rows2 ? (b * N + n) * H + yt : b * N + n];
  return (sdr >= -lim && sdr <= lim) ? sdr : ((sdr < 0.0f) ? -lim : lim);   // NaN -> +lim
}

template <bool FLIP, int NR, int NMAX, bool ROWS2>
__device__ __forceinline__ void chain_row(const WarpArgs& a, const float* __restrict__ disp2, float sign2, int rows2_, int B, int b,
                                          int y, const RowTaps& r, float* __restrict__ lds, float* __restrict__ S) {
  const int W = a.W, N = a.N, HW = a.H * a.W, RS = W + 4;
  constexpr int rows2 = ROWS2 ? 1 : 0;   // (a template flag: as a run-time one it cost the per-plane case 30-50 % — NOTEBOOK 11.4)
  (void)rows2_;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* pb = a.planes + (long)b * N * HW;
  for (int i = threadIdx.x; i < 4 * N; i += blockDim.x) {   // zero guard cells: columns -2, -1, W, W + 1 of every plane
    const int n = i >> 2, g = i & 3;
    lds[n * RS + (g < 2 ? g : W + g)] = 0.0f;
  }
  // (one segment per wave, no loop around the unrolled planes: inside a loop every plane's shift and descriptors are
  // loop-invariant, get hoisted, and 5 500 spilled SGPRs later the kernel lives in scratch)
  const int seg = wave;
  {
    const int xseg = seg * kSegPix, x0 = xseg + 2 * lane;
    float l0[NMAX], l1[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) l0[n] = l1[n] = -INFINITY;
    float m0 = -INFINITY, m1 = -INFINITY;
    auto keep = [&](int n, float v0, float v1) {
#pragma unroll
      for (int q = 0; q < NMAX; ++q) if (q == n) { l0[q] = v0; l1[q] = v1; }
      m0 = fmaxf(m0, v0); m1 = fmaxf(m1, v1);
    };
    // (one workgroup of nseg waves per CU: the loads in flight have to come from each wave — PD_PP_CHAIN_G planes at a time)
    seg_for_each<FLIP, NR, NMAX, decltype(keep), PD_PP_CHAIN_G>(a, pb, r, b, xseg, x0, HW, keep);
    float Z0 = 0.0f, Z1 = 0.0f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      l0[n] = __expf(l0[n] - m0); l1[n] = __expf(l1[n] - m1);
      Z0 += l0[n]; Z1 += l1[n];
    }
    if (x0 < W) {
      const float i0 = 1.0f / Z0, i1 = 1.0f / Z1;
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) *reinterpret_cast<float2*>(lds + n * RS + 2 + x0) = make_float2(l0[n] * i0, l1[n] * i1);
    }
  }
  __syncthreads();
  const float lim = (float)(W + 2);
  for (int v = 0; v < (rows2 ? 3 : 1); ++v) {   // S_r under the shifts of target row y, y - 1, y + 1 (workgroup-uniform)
    const int yt = (v == 0) ? y : ((v == 1) ? y - 1 : y + 1);
    if (v > 0 && !chain_needs(y, yt, a.H)) continue;
    const int x0 = seg * kSegPix + 2 * lane;
    float acc0 = 0.0f, acc1 = 0.0f;
    const float sh2 = chain_shifts(disp2, sign2, rows2, b, N, a.H, yt, lim);
    for (int n = 0; n < N; ++n) {
      const float sd = shift_of(sh2, n);
      const float* row = lds + n * RS;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const ColTap t = make_col_tap((float)(x0 + i) + sd, a.Wm1, a.rcpWm1);
        const int cell = min(max(t.x0, -2), W) + 2;
        const float val = row[cell] * t.w0 + row[cell + 1] * t.w1;
        if (i == 0) acc0 += val; else acc1 += val;
      }
    }
    if (x0 < W) *reinterpret_cast<float2*>(S + ((long)v * B + b) * HW + (long)y * W + x0) = make_float2(acc0, acc1);
  }
}

// The same with the planes of a segment dealt to P waves (P x nseg waves per workgroup: 15 at W = 640).  One workgroup per CU is
// all the row buffer allows, so the loads in flight and the issue slots of four SIMDs have to come from its own waves: with one
// wave per segment (5 per CU) the kernel ran at 1.2 TB/s.  Wave (seg, part) samples planes [part * NPART, (part + 1) * NPART),
// leaves (max, sum of exp relative to it) per pixel in LDS, takes the segment's max / sum from the P parts after a barrier, writes
// its planes' probabilities, and after the next barrier sums ITS planes' taps of the second warp; part 0 adds the partial sums.
template <bool FLIP, int NR, int NPART, int P, bool ROWS2, bool ALIAS>
__device__ __forceinline__ void chain_row_split(const WarpArgs& a, const float* __restrict__ disp2, float sign2, int rows2_, int B,
                                                int b, int y, const RowTaps& r, float* __restrict__ lds, float* __restrict__ S) {
  const int W = a.W, N = a.N, HW = a.H * a.W, RS = W + 4;
  constexpr int rows2 = ROWS2 ? 1 : 0;   // (a template flag: as a run-time one it cost the per-plane case 30-50 % — NOTEBOOK 11.4)
  (void)rows2_;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (W + kSegPix - 1) / kSegPix;
  const int seg = wave % nseg, part = wave / nseg;
  const int nb = part * NPART, ne = min(nb + NPART, N);
  const int SW = nseg * kSegPix;                                 // stats row: one float2 per pixel of the padded row and part
  // ALIAS (the row buffer alone fills the CU's LDS: 63 planes x 640): the parts' statistics and, later, their partial sums live
  // INSIDE the row buffer — while no probabilities are there yet, and after the last one has been read; two more barriers
  float2* stats = ALIAS ? reinterpret_cast<float2*>(lds) : reinterpret_cast<float2*>(lds + (size_t)N * RS);
  const float* pb = a.planes + (long)b * N * HW;
  auto zero_guards = [&]() {
    for (int i = threadIdx.x; i < 4 * N; i += blockDim.x) {   // zero guard cells: columns -2, -1, W, W + 1 of every plane
      const int n = i >> 2, g = i & 3;
      lds[n * RS + (g < 2 ? g : W + g)] = 0.0f;
    }
  };
  if (!ALIAS) zero_guards();
  const int xseg = seg * kSegPix, x0 = xseg + 2 * lane;
  float l0[NPART], l1[NPART];
#pragma unroll
  for (int j = 0; j < NPART; ++j) l0[j] = l1[j] = -INFINITY;
  float m0 = -INFINITY, m1 = -INFINITY;
  auto keep = [&](int j, float v0, float v1) {
#pragma unroll
    for (int q = 0; q < NPART; ++q) if (q == j) { l0[q] = v0; l1[q] = v1; }
    m0 = fmaxf(m0, v0); m1 = fmaxf(m1, v1);
  };
  seg_for_each<FLIP, NR, NPART, decltype(keep), kGroup>(a, pb, r, b, xseg, x0, HW, keep, nb, ne);
  float Z0 = 0.0f, Z1 = 0.0f;
#pragma unroll
  for (int j = 0; j < NPART; ++j) {   // (slots beyond the part's planes hold -inf: exp gives 0; a part without planes: m = -inf, Z = 0)
    l0[j] = (m0 == -INFINITY) ? 0.0f : __expf(l0[j] - m0);
    l1[j] = (m1 == -INFINITY) ? 0.0f : __expf(l1[j] - m1);
    Z0 += l0[j]; Z1 += l1[j];
  }
  stats[part * SW + x0] = make_float2(m0, Z0);
  stats[part * SW + x0 + 1] = make_float2(m1, Z1);
  __syncthreads();
  float M0 = -INFINITY, M1 = -INFINITY;
#pragma unroll
  for (int q = 0; q < P; ++q) { M0 = fmaxf(M0, stats[q * SW + x0].x); M1 = fmaxf(M1, stats[q * SW + x0 + 1].x); }
  float T0 = 0.0f, T1 = 0.0f;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const float2 s0 = stats[q * SW + x0], s1 = stats[q * SW + x0 + 1];
    T0 += (s0.x == -INFINITY) ? 0.0f : s0.y * __expf(s0.x - M0);
    T1 += (s1.x == -INFINITY) ? 0.0f : s1.y * __expf(s1.x - M1);
  }
  const float c0 = (m0 == -INFINITY) ? 0.0f : __expf(m0 - M0) / T0, c1 = (m1 == -INFINITY) ? 0.0f : __expf(m1 - M1) / T1;
  if (ALIAS) {   // every wave has its statistics in registers before the first probability overwrites them
    __syncthreads();
    zero_guards();
  }
  if (x0 < W) {
#pragma unroll
    for (int j = 0; j < NPART; ++j)
      if (nb + j < ne) *reinterpret_cast<float2*>(lds + (nb + j) * RS + 2 + x0) = make_float2(l0[j] * c0, l1[j] * c1);
  }
  __syncthreads();   // (every wave has read the stats by now: the buffer is free for the partial sums)
  const float lim = (float)(W + 2);
  constexpr int NV = ROWS2 ? 3 : 1;
  float accv[NV][2];
  bool need[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {   // S_r under the shifts of target row y, y - 1, y + 1 (workgroup-uniform)
    const int yt = (v == 0) ? y : ((v == 1) ? y - 1 : y + 1);
    need[v] = (v == 0) || chain_needs(y, yt, a.H);
    accv[v][0] = accv[v][1] = 0.0f;
    if (!need[v]) continue;
    float acc0 = 0.0f, acc1 = 0.0f;
    const float sh2 = chain_shifts(disp2, sign2, rows2, b, N, a.H, yt, lim);
    for (int n = nb; n < ne; ++n) {
      const float sd = shift_of(sh2, n);
      const float* row = lds + n * RS;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const ColTap t = make_col_tap((float)(x0 + i) + sd, a.Wm1, a.rcpWm1);
        const int cell = min(max(t.x0, -2), W) + 2;
        const float val = row[cell] * t.w0 + row[cell + 1] * t.w1;
        if (i == 0) acc0 += val; else acc1 += val;
      }
    }
    accv[v][0] = acc0; accv[v][1] = acc1;
    if (!ALIAS) {   // the partial sums go through the statistics' buffer, variant by variant
      if (part > 0) stats[part * SW + x0] = make_float2(acc0, acc1);
      __syncthreads();
      if (part == 0 && x0 < W) {
#pragma unroll
        for (int q = 1; q < P; ++q) { const float2 o = stats[q * SW + x0]; acc0 += o.x; acc1 += o.y; }
        *reinterpret_cast<float2*>(S + ((long)v * B + b) * HW + (long)y * W + x0) = make_float2(acc0, acc1);
      }
      if (ROWS2) __syncthreads();   // (the partial sums of the next variant go into the same buffer)
    }
  }
  if (ALIAS) {   // the probabilities have been read for the last time: the partial sums of every variant go where they were
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (need[v] && part > 0) stats[(v * P + part) * SW + x0] = make_float2(accv[v][0], accv[v][1]);
    __syncthreads();
    if (part == 0 && x0 < W) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (!need[v]) continue;
        float acc0 = accv[v][0], acc1 = accv[v][1];
#pragma unroll
        for (int q = 1; q < P; ++q) { const float2 o = stats[(v * P + q) * SW + x0]; acc0 += o.x; acc1 += o.y; }
        *reinterpret_cast<float2*>(S + ((long)v * B + b) * HW + (long)y * W + x0) = make_float2(acc0, acc1);
      }
    }
  }
}

template <int NPART, int P, bool ROWS2, bool ALIAS>
__global__ __launch_bounds__(1024) void pp_chain_split_kernel(ChainArgs c) {
  extern __shared__ float chain_lds[];
  const int H = c.w[0].H;
  const int item = blockIdx.x;
  const int y = item % H, b = (item / H) % c.B, chain = item / (H * c.B);
  const RowTaps r = row_taps(y, H);
  if (chain == 0) {
    if (r.wb != 0.0f) chain_row_split<false, 2, NPART, P, ROWS2, ALIAS>(c.w[0], c.disp2[0], c.sign2[0], c.rows2, c.B, b, y, r, chain_lds, c.S[0]);
    else              chain_row_split<false, 1, NPART, P, ROWS2, ALIAS>(c.w[0], c.disp2[0], c.sign2[0], c.rows2, c.B, b, y, r, chain_lds, c.S[0]);
  } else {
    if (r.wb != 0.0f) chain_row_split<true, 2, NPART, P, ROWS2, ALIAS>(c.w[1], c.disp2[1], c.sign2[1], c.rows2, c.B, b, y, r, chain_lds, c.S[1]);
    else              chain_row_split<true, 1, NPART, P, ROWS2, ALIAS>(c.w[1], c.disp2[1], c.sign2[1], c.rows2, c.B, b, y, r, chain_lds, c.S[1]);
  }
}

template <int NMAX, bool ROWS2>
__global__ __launch_bounds__(512) void pp_chain_kernel(ChainArgs c) {   // (at most 8 waves: 256 VGPRs for the sample registers)
  extern __shared__ float chain_lds[];
  const int H = c.w[0].H;
  const int item = blockIdx.x;                       // (chain, image, row): the rows of one image side by side
  const int y = item % H, b = (item / H) % c.B, chain = item / (H * c.B);
  const RowTaps r = row_taps(y, H);
  if (chain == 0) {
    if (r.wb != 0.0f) chain_row<false, 2, NMAX, ROWS2>(c.w[0], c.disp2[0], c.sign2[0], c.rows2, c.B, b, y, r, chain_lds, c.S[0]);
    else              chain_row<false, 1, NMAX, ROWS2>(c.w[0], c.disp2[0], c.sign2[0], c.rows2, c.B, b, y, r, chain_lds, c.S[0]);
  } else {
    if (r.wb != 0.0f) chain_row<true, 2, NMAX, ROWS2>(c.w[1], c.disp2[1], c.sign2[1], c.rows2, c.B, b, y, r, chain_lds, c.S[1]);
    else              chain_row<true, 1, NMAX, ROWS2>(c.w[1], c.disp2[1], c.sign2[1], c.rows2, c.B, b, y, r, chain_lds, c.S[1]);
  }
}

// o_l / o_fr from the S rows (vertical weights of the second warp, the clamp of trainer.py:449 / 456) and the disp_pp blend.
// S is [3][B,1,H,W]: source row rho contributes to target y through the variant computed with y's shifts — 0 for rho == y,
// 1 for rho == y + 1, 2 for rho == y - 1 (`rows2`; with per-plane shifts variant 0 serves everyone).  A zero weight reads nothing.
__device__ __forceinline__ float chain_o(const float* __restrict__ S, int B, int b, long HW, int W, int x, int y, const RowTaps& r,
                                         int rows2) {
#pragma clang fp contract(off)
  float o = 0.0f;
  if (r.wa != 0.0f) o = S[((long)((!rows2 || r.ra == y) ? 0 : ((r.ra == y + 1) ? 1 : 2)) * B + b) * HW + (long)r.ra * W + x] * r.wa;
  if (r.wb != 0.0f) o = o + S[((long)((!rows2 || r.rb == y) ? 0 : ((r.rb == y + 1) ? 1 : 2)) * B + b) * HW + (long)r.rb * W + x] * r.wb;
  return fminf(o, 1.0f);
}
__global__ __launch_bounds__(kBlock) void pp_rows_finish_kernel(int B, int H, int W, int rows2, const float* __restrict__ S_l,
                                                                const float* __restrict__ S_fr, const float* __restrict__ disp,
                                                                float* __restrict__ out) {
#pragma clang fp contract(off)
  const long HW = (long)H * W, i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW);
  const long p = i - (long)b * HW;
  const int y = (int)(p / W), x = (int)(p - (long)y * W);
  const RowTaps r = row_taps(y, H);
  const float ol = chain_o(S_l, B, b, HW, W, x, y, r, rows2), ofr = chain_o(S_fr, B, b, HW, W, x, y, r, rows2);
  const float d0 = disp[i], df = disp[((long)B + b) * HW + (long)y * W + (W - 1 - x)];
  const float mean = d0 * 0.5f + df * 0.5f;
  float pp = mean * ofr + d0 * (1.0f - ofr);
  pp = pp * ol + df * (1.0f - ol);
  out[i] = pp;
}

// disp_pp of trainer.py:458-461 from the two occlusion masks: one pass instead of eight elementwise launches.
//   mean = disp[b] * 0.5 + flip(disp[B + b]) * 0.5;  pp = mean * o_fr + disp[b] * (1 - o_fr);  pp = pp * o_l + flip(disp[B + b]) * (1 - o_l)
// in torch's operation order (every product and sum rounded on its own).
__global__ __launch_bounds__(kBlock) void pp_combine_kernel(int B, int H, int W, const float* __restrict__ disp,
                                                            const float* __restrict__ o_fr, const float* __restrict__ o_l,
                                                            float* __restrict__ out) {
#pragma clang fp contract(off)
  const long HW = (long)H * W, i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW);
  const long p = i - (long)b * HW;
  const int y = (int)(p / W), x = (int)(p - (long)y * W);
  const float d0 = disp[i], df = disp[((long)B + b) * HW + (long)y * W + (W - 1 - x)];
  const float ofr = o_fr[i], ol = o_l[i];
  const float mean = d0 * 0.5f + df * 0.5f;
  float pp = mean * ofr + d0 * (1.0f - ofr);
  pp = pp * ol + df * (1.0f - ol);
  out[i] = pp;
}

// pixel pairs: even width, 8-byte aligned output rows; the softmax holds 2 N samples in registers
static bool seg_applicable(const WarpArgs& a, int B, const float* out, int nmax) {
  return !switches().pp_seg_off && !a.dense && (a.W % 2 == 0) && a.N <= nmax && (reinterpret_cast<uintptr_t>(out) & 7) == 0 &&
         (long)B * a.H * ((a.W + kSegPix - 1) / kSegPix) < (1L << 31) * kSegWaves && a.W <= (1 << 24);
}

// the row kernels need 32-bit byte offsets inside a row and grid dimensions within the launch limits
static bool rows_applicable(const WarpArgs& a, int B) {
  return !switches().pp_rows_off && !a.dense && a.H <= 65535 && B <= 65535 && a.W <= (1 << 24);
}

static int warp_args(WarpArgs& a, int B, int N, int H, int W, float sign, int flags, const float* planes,
                     const float* disp) {
  PD_REQUIRE(B > 0 && B <= 65535 && N > 0 && H > 0 && W > 1, "bad shape");
  PD_REQUIRE((long)H * W < (1L << 31), "image too large");
  PD_REQUIRE((flags & ~(PD_PP_DISP_DENSE | PD_PP_FLIP_SRC | PD_PP_DISP_ROWS)) == 0, "unknown flags");
  PD_REQUIRE(!((flags & PD_PP_DISP_DENSE) && (flags & PD_PP_DISP_ROWS)), "PD_PP_DISP_DENSE and PD_PP_DISP_ROWS exclude each other");
  PD_REQUIRE(planes && disp, "NULL pointer");
  a.N = N; a.H = H; a.W = W;
  a.dense = (flags & PD_PP_DISP_DENSE) != 0;
  a.rows = (flags & PD_PP_DISP_ROWS) != 0;
  PD_REQUIRE(!a.rows || (long)B * N * H < (1L << 31), "too many (image, plane, row) disparities for 32-bit indices");
  a.dimg = a.rows ? N * H : N; a.dplane = a.rows ? H : 1; a.ymask = a.rows ? ~0 : 0;
  a.flip = (flags & PD_PP_FLIP_SRC) != 0;
  a.sign = sign; a.planes = planes; a.disp = disp;
  a.Wm1 = (float)(W - 1);
  {  // the correctly rounded fp32 reciprocal of W-1: candidates around the double quotient, |r * d - 1| exact in double
    const double dd = (double)(W - 1);
    float best = (float)(1.0 / dd);
    const float cand[2] = {nextafterf(best, 0.0f), nextafterf(best, 2.0f)};
    for (float c : cand)
      if (fabs((double)c * dd - 1.0) < fabs((double)best * dd - 1.0)) best = c;
    a.rcpWm1 = best;
  }
  return 0;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_warp_softmax(int B, int N, int H, int W, float sign, int flags, const float* planes,
                               const float* disp, float* out, pd_stream_t stream) {
  WarpArgs a;
  if (int rc = warp_args(a, B, N, H, W, sign, flags, planes, disp)) return rc;
  PD_REQUIRE(out && out != planes, "out must be a distinct buffer");
  if (seg_applicable(a, B, out, 64)) {
    const int nseg = ceil_div(W, kSegPix);
    const dim3 g((unsigned)(((long)B * H * nseg + kSegWaves - 1) / kSegWaves));
    if (N <= 32) { if (a.flip) warp_softmax_seg_kernel<true, 32><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, out);
                   else        warp_softmax_seg_kernel<false, 32><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, out); }
    else if (N <= 52) { if (a.flip) warp_softmax_seg_kernel<true, 52><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, out);   // (49 planes)
                        else        warp_softmax_seg_kernel<false, 52><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, out); }
    else         { if (a.flip) warp_softmax_seg_kernel<true, 64><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, out);
                   else        warp_softmax_seg_kernel<false, 64><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, out); }
    return check_launch("warp_softmax_seg_kernel");
  }
  if (rows_applicable(a, B)) {
    dim3 g(ceil_div(W, kWave), H, B);
    if (a.flip) warp_softmax_rows_kernel<true><<<g, kWave, 0, (hipStream_t)stream>>>(a, out);
    else        warp_softmax_rows_kernel<false><<<g, kWave, 0, (hipStream_t)stream>>>(a, out);
    return check_launch("warp_softmax_rows_kernel");
  }
  dim3 grid(ceil_div(H * W, kBlock), B);
  if (a.flip) warp_softmax_kernel<true><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, out);
  else        warp_softmax_kernel<false><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, out);
  return check_launch("warp_softmax_kernel");
}

extern "C" int pd_warp_sum(int B, int N, int H, int W, float sign, int flags, const float* planes, const float* disp,
                           float cap, float* out, pd_stream_t stream) {
  WarpArgs a;
  if (int rc = warp_args(a, B, N, H, W, sign, flags, planes, disp)) return rc;
  PD_REQUIRE(out, "NULL output");
  if (seg_applicable(a, B, out, 1 << 20)) {
    const int nseg = ceil_div(W, kSegPix);
    const dim3 g((unsigned)(((long)B * H * nseg + kSegWaves - 1) / kSegWaves));
    if (a.flip) warp_sum_seg_kernel<true><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, cap, out);
    else        warp_sum_seg_kernel<false><<<g, kSegWaves * kWave, 0, (hipStream_t)stream>>>(a, B, cap, out);
    return check_launch("warp_sum_seg_kernel");
  }
  if (rows_applicable(a, B)) {
    dim3 g(ceil_div(W, kWave), H, B);
    if (a.flip) warp_sum_rows_kernel<true><<<g, kWave, 0, (hipStream_t)stream>>>(a, cap, out);
    else        warp_sum_rows_kernel<false><<<g, kWave, 0, (hipStream_t)stream>>>(a, cap, out);
    return check_launch("warp_sum_rows_kernel");
  }
  dim3 grid(ceil_div(H * W, kBlock), B);
  if (a.flip) warp_sum_kernel<true><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, cap, out);
  else        warp_sum_kernel<false><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, cap, out);
  return check_launch("warp_sum_kernel");
}

extern "C" int pd_pp_combine(int B, int H, int W, const float* disp, const float* o_fr, const float* o_l, float* disp_pp,
                             pd_stream_t stream) {
  PD_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31) * kBlock, "bad shape");
  PD_REQUIRE(disp && o_fr && o_l && disp_pp, "NULL pointer");
  pp_combine_kernel<<<(unsigned)(((long)B * H * W + kBlock - 1) / kBlock), kBlock, 0, (hipStream_t)stream>>>(B, H, W, disp, o_fr, o_l, disp_pp);
  return check_launch("pp_combine_kernel");
}

// The whole of trainer.py:443-465 behind one call: six launches on `stream`, no host work in between (as separate operator
// calls the ~0.2 ms of kernels at 8 x 49 x 192 x 640 were paced by the host's per-call overhead).
extern "C" size_t pd_post_process_workspace_floats(int B, int N, int H, int W) {
  return (size_t)B * N * H * W + (size_t)6 * B * H * W;   // the single warps' [B,N,H,W] intermediate + o_l / o_fr, or the chains' 2 x 3 S maps
}

extern "C" int pd_post_process(int B, int N, int H, int W, int flags, const float* logits, const float* probability,
                               const float* disp, const float* disp_layered, float* workspace, float* disp_pp, float* mask_novel,
                               pd_stream_t stream) {
  PD_REQUIRE(B > 0 && N > 0 && H > 0 && W > 1, "bad shape");
  PD_REQUIRE((flags & ~(PD_PP_DISP_DENSE | PD_PP_DISP_ROWS)) == 0, "unknown flags");
  PD_REQUIRE(logits && probability && disp && disp_layered && workspace && disp_pp && mask_novel, "NULL pointer");
  const size_t P = (size_t)H * W, img = (size_t)N * P;
  const float* dl_r = disp_layered;                                                              // the image's planes
  const float* dl_l = disp_layered + ((flags & PD_PP_DISP_DENSE) ? (size_t)B * img                  // the mirrored image's
                                                                  : ((flags & PD_PP_DISP_ROWS) ? (size_t)B * N * H : (size_t)B * N));
  float* planes = workspace;
  float* o_l = workspace + (size_t)B * img;
  float* o_fr = o_l + (size_t)B * P;
  int rc;
  // Row chains where the softmax of a row fits the CU's LDS: per-plane disparities, pixel pairs, N <= 64
  const size_t chain_lds = (size_t)N * (W + 4) * sizeof(float);
  if (!(flags & PD_PP_DISP_DENSE) && !switches().pp_seg_off && !switches().pp_chain_off && (W % 2 == 0) && W <= 8 * kSegPix && N <= 64 && H <= 65535 &&
      chain_lds <= device_lds_bytes() && (long)2 * B * H < (1L << 31) && ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(disp_pp)) & 7) == 0) {
    ChainArgs c;
    const int rowsf = flags & PD_PP_DISP_ROWS;
    if ((rc = warp_args(c.w[0], B, N, H, W, +1.0f, rowsf, logits, dl_r))) return rc;
    if ((rc = warp_args(c.w[1], B, N, H, W, -1.0f, rowsf | PD_PP_FLIP_SRC, logits + (size_t)B * img, dl_l))) return rc;
    c.disp2[0] = dl_l; c.sign2[0] = -1.0f;
    c.disp2[1] = dl_r; c.sign2[1] = +1.0f;
    c.S[0] = workspace; c.S[1] = workspace + (size_t)3 * B * P;   // [3][B,1,H,W] each
    c.B = B; c.rows2 = rowsf ? 1 : 0;
    const int nseg = ceil_div(W, kSegPix);
    const dim3 grid((unsigned)(2 * B * H));
    // planes dealt to P = 3 waves per segment where the workgroup (3 nseg <= 16 waves) and its LDS (row buffer + one float2 per
    // pixel and part) fit; else one wave per segment
    const size_t side_lds = chain_lds + (size_t)3 * nseg * kSegPix * sizeof(float2);
    const bool alias = side_lds > device_lds_bytes();   // the parts' statistics / partial sums inside the row buffer (needs 9 maps of
    const size_t split_lds = alias ? chain_lds : side_lds;   // nseg x 128 float2 there: N >= 18 at W = 640)
    if (PD_PP_CHAIN_SPLIT && 3 * nseg <= 16 && (!alias || (size_t)9 * nseg * kSegPix * sizeof(float2) <= chain_lds)) {
      const dim3 block3(3 * nseg * kWave);
#define PD_PP_SPLIT_(NPART, R2, AL)                                                                                                 \
      do {                                                                                                                        \
        static LdsGrant granted;                                                                                                  \
        if ((rc = grant_dynamic_lds((const void*)pp_chain_split_kernel<NPART, 3, R2, AL>, split_lds, &granted, "pp_chain_split_kernel"))) return rc; \
        pp_chain_split_kernel<NPART, 3, R2, AL><<<grid, block3, split_lds, (hipStream_t)stream>>>(c);                             \
      } while (0)
#define PD_PP_SPLIT(NPART)                                                                              \
      do {                                                                                              \
        if (alias) { if (c.rows2) PD_PP_SPLIT_(NPART, true, true); else PD_PP_SPLIT_(NPART, false, true); } \
        else       { if (c.rows2) PD_PP_SPLIT_(NPART, true, false); else PD_PP_SPLIT_(NPART, false, false); } \
      } while (0)
      if (N <= 33) PD_PP_SPLIT(11); else if (N <= 54) PD_PP_SPLIT(18); else PD_PP_SPLIT(22);
#undef PD_PP_SPLIT_
#undef PD_PP_SPLIT
      if ((rc = check_launch("pp_chain_split_kernel"))) return rc;
      pp_rows_finish_kernel<<<(unsigned)(((long)B * P + kBlock - 1) / kBlock), kBlock, 0, (hipStream_t)stream>>>(B, H, W, c.rows2, c.S[0], c.S[1], disp, disp_pp);
      if ((rc = check_launch("pp_rows_finish_kernel"))) return rc;
      return pd_warp_sum(B, N, H, W, +1.0f, flags, probability, dl_r, 1.0f, mask_novel, stream);                         // :463-465
    }
    const dim3 block(nseg * kWave);   // one wave per segment
#define PD_PP_CHAIN_(NMAX, R2)                                                                                               \
    do {                                                                                                                    \
      static LdsGrant granted;                                                                                              \
      if ((rc = grant_dynamic_lds((const void*)pp_chain_kernel<NMAX, R2>, chain_lds, &granted, "pp_chain_kernel"))) return rc;  \
      pp_chain_kernel<NMAX, R2><<<grid, block, chain_lds, (hipStream_t)stream>>>(c);                                        \
    } while (0)
#define PD_PP_CHAIN(NMAX) do { if (c.rows2) PD_PP_CHAIN_(NMAX, true); else PD_PP_CHAIN_(NMAX, false); } while (0)
    if (N <= 32) PD_PP_CHAIN(32); else if (N <= 52) PD_PP_CHAIN(52); else PD_PP_CHAIN(64);
#undef PD_PP_CHAIN
#undef PD_PP_CHAIN_
    if ((rc = check_launch("pp_chain_kernel"))) return rc;
    pp_rows_finish_kernel<<<(unsigned)(((long)B * P + kBlock - 1) / kBlock), kBlock, 0, (hipStream_t)stream>>>(B, H, W, c.rows2, c.S[0], c.S[1], disp, disp_pp);
    if ((rc = check_launch("pp_rows_finish_kernel"))) return rc;
    return pd_warp_sum(B, N, H, W, +1.0f, flags, probability, dl_r, 1.0f, mask_novel, stream);                           // :463-465
  }
  if ((rc = pd_warp_softmax(B, N, H, W, +1.0f, flags, logits, dl_r, planes, stream))) return rc;                         // :443-446
  if ((rc = pd_warp_sum(B, N, H, W, -1.0f, flags, planes, dl_l, 1.0f, o_l, stream))) return rc;                         // :447-449
  if ((rc = pd_warp_softmax(B, N, H, W, -1.0f, flags | PD_PP_FLIP_SRC, logits + (size_t)B * img, dl_l, planes, stream))) return rc;   // :451-453
  if ((rc = pd_warp_sum(B, N, H, W, +1.0f, flags, planes, dl_r, 1.0f, o_fr, stream))) return rc;                        // :454-456
  if ((rc = pd_pp_combine(B, H, W, disp, o_fr, o_l, disp_pp, stream))) return rc;                                        // :458-461
  return pd_warp_sum(B, N, H, W, +1.0f, flags, probability, dl_r, 1.0f, mask_novel, stream);                             // :463-465
}
