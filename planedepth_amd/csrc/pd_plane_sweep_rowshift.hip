// Row-shift specialisation of the fused plane sweep: PD_WARP_DISP with one scalar disparity per (image, plane)
// (xy planes: the decoder's disp_layered is an expanded [B,N,1,1], reference networks/depth_decoder.py:153-156).
//
// Structure exploited: along one target row every pixel samples plane n at x + s*d_n, i.e. the warp is a shift that
// is uniform over the row.  Consequences used here:
//   * one workgroup owns one target row (b, y): the vertical taps/weights are workgroup-uniform, and rows whose
//     second vertical weight is exactly zero (3 of 4 rows at H=192: the normalise->unnormalise round trip returns y
//     exactly) touch ONE source row instead of two (kernels are specialised on the number of live rows);
//   * the source colour rows are staged once per workgroup in LDS as packed float4 (r,g,b,-): every plane then
//     reads its two colour taps with two ds_read_b128 instead of six global loads;
//   * the plane loop is processed in groups of U planes whose global loads are all issued before the first use, so
//     each wave keeps 4U..8U loads in flight (the loop is otherwise latency-bound: one dependent round trip per plane);
//   * the adjoint of the horizontal 2-tap gather is itself a 2-tap GATHER: source pixel xs receives from the targets
//     whose left tap is xs or xs-1, which are the lane itself and its left neighbour.  The backward therefore writes
//     g_logits / g_sigma with plain coalesced stores — no atomics, no zero-fill pass, each element written once.
//
// Exactness of the shifted ownership.  Slot i of the row (i = target index xt) owns source pixel (xt + k) mod W with
// k = floor(s*d) the nominal shift.  In exact arithmetic floor(ix(xt)) = xt + k; in fp32 the reference's
// normalise/un-normalise round trip can move ix across an integer when frac(s*d) is within ~1e-4 of 0 or 1, so
// delta = floor(ix) - xt - k may be -1, 0 or +1 for individual lanes.  The fast path (wave-uniform delta == 0) is one
// cross-lane shift; otherwise a general path routes every contribution to slot lane+delta(+1) with neighbour
// shuffles.  Contributions that leave a 64-lane segment are parked in LDS and added after the plane loop (the row
// is treated as a ring of W slots, which is also what gives wrapped slots their zero / boundary values).
//
// Vertical adjoint: a row whose round trip is inexact has weights (1-eps, eps) with eps <= ~8e-6 on two source rows.
// The forward and the per-pixel gradient use both rows exactly; the adjoint applies the (1-eps) weight to the
// workgroup's own row and drops the eps-weighted term of the neighbouring row (relative size <= 1e-5, an order of
// magnitude inside the 1e-4 parity budget; measured in tests/test_gpu_parity.py).  The general kernels keep it.
#include "pd_sweep.h"

namespace pd {

constexpr int kMaxRowThreads = 1024;

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
  int x0;           // floor(ix), clamped to [-2, W] (only meaningful when v0 || v1)
  unsigned i0, i1;  // x0 and x0+1 if inside the image, else 0: always safe to load
  float m0, m1;     // torch's weights (x1 - ix), (ix - x0) with out-of-image taps zeroed (padding_mode="zeros")
  bool v0, v1;      // tap inside the image
};

// Correctly rounded a / b from the correctly rounded reciprocal of b (Markstein's theorem; b = W-1 is an integer
// <= 2^24 and a is far from the over/underflow range, so no special cases arise).  Verified bit-for-bit against
// IEEE division over the whole coordinate range by tests/test_gpu_parity.py::test_fast_division_is_exact.
__device__ __forceinline__ float div_by(float a, float b, float rcp_b) {
  const float q0 = a * rcp_b;
  const float r = fmaf(-q0, b, a);
  return fmaf(r, rcp_b, q0);
}

__device__ __forceinline__ float refined_rcp(float b) {
  float y = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, y, 1.0f);
  return fmaf(e, y, y);
}

__device__ __forceinline__ ColTap make_col_tap(float px, float Wm1, float rcpWm1, int W) {
  ColTap t;
  float ix;
  {
#pragma clang fp contract(off)
    const float q = div_by(px, Wm1, rcpWm1);
    const float h = q - 0.5f;
    const float g = h * 2.0f;                               // trainer.py:550-552
    ix = unnormalise(g, Wm1);                               // grid_sample un-normalisation, align_corners=True
  }
  const float xf = floorf(ix);
  const float wx0 = (xf + 1.0f) - ix, wx1 = ix - xf;
  t.x0 = (int)fminf(fmaxf(xf, -2.0f), (float)W);            // NaN / huge coordinates end up outside
  t.v0 = (unsigned)t.x0 < (unsigned)W;
  t.v1 = (unsigned)(t.x0 + 1) < (unsigned)W;
  t.i0 = t.v0 ? (unsigned)t.x0 : 0u;
  t.i1 = t.v1 ? (unsigned)(t.x0 + 1) : 0u;
  t.m0 = t.v0 ? wx0 : 0.0f;
  t.m1 = t.v1 ? wx1 : 0.0f;
  return t;
}

// The (up to) four taps of one scalar plane, loaded up-front.
template <int NROWS>
struct Taps {
  float a0, a1, b0, b1;  // row A (x0, x0+1), row B (x0, x0+1)
};

// Loads are UNCONDITIONAL (clamped indices off a workgroup-uniform row pointer): no exec-mask branches, so the compiler
// can issue a whole group's loads back to back.  Out-of-image taps are cancelled by the zeroed weights m0/m1.
template <int NROWS>
__device__ __forceinline__ Taps<NROWS> load_taps(const float* __restrict__ rowA, const float* __restrict__ rowB,
                                                 const ColTap& c) {
  Taps<NROWS> t;
  t.a0 = rowA[c.i0];
  t.a1 = rowA[c.i1];
  t.b0 = t.b1 = 0.0f;
  if (NROWS == 2) {
    t.b0 = rowB[c.i0];
    t.b1 = rowB[c.i1];
  }
  return t;
}

template <int NROWS>
__device__ __forceinline__ float tap_value(const Taps<NROWS>& t, const RowSel& r, const ColTap& c) {
  float v = t.a0 * (c.m0 * r.wA) + t.a1 * (c.m1 * r.wA);
  if (NROWS == 2) v += t.b0 * (c.m0 * r.wB) + t.b1 * (c.m1 * r.wB);
  return v;
}

// d value / d ix: (ne - nw) * wy with out-of-image taps reading as zero
template <int NROWS>
__device__ __forceinline__ float tap_dx(const Taps<NROWS>& t, const RowSel& r, const ColTap& c) {
  float d = ((c.v1 ? t.a1 : 0.0f) - (c.v0 ? t.a0 : 0.0f)) * r.wA;
  if (NROWS == 2) d += ((c.v1 ? t.b1 : 0.0f) - (c.v0 ? t.b0 : 0.0f)) * r.wB;
  return d;
}

// Colour taps from the LDS copy of the source rows: [row][x] float4 (r, g, b, unused)
template <int NROWS>
__device__ __forceinline__ void colour_taps(const float4* __restrict__ lrgb, int W, const RowSel& r, const ColTap& c,
                                            float& c0, float& c1, float& c2) {
  const float4 nw = lrgb[c.i0], ne = lrgb[c.i1];
  const float w0 = c.m0 * r.wA, w1 = c.m1 * r.wA;
  c0 = nw.x * w0 + ne.x * w1;
  c1 = nw.y * w0 + ne.y * w1;
  c2 = nw.z * w0 + ne.z * w1;
  if (NROWS == 2) {
    const float4 sw = lrgb[W + c.i0], se = lrgb[W + c.i1];
    const float u0 = c.m0 * r.wB, u1 = c.m1 * r.wB;
    c0 += sw.x * u0 + se.x * u1;
    c1 += sw.y * u0 + se.y * u1;
    c2 += sw.z * u0 + se.z * u1;
  }
}

template <int NROWS>
__device__ __forceinline__ void colour_taps_dx(const float4* __restrict__ lrgb, int W, const RowSel& r,
                                               const ColTap& c, float& c0, float& c1, float& c2, float& d0, float& d1,
                                               float& d2) {
  const float4 nw = lrgb[c.i0], ne = lrgb[c.i1];
  const float w0 = c.m0 * r.wA, w1 = c.m1 * r.wA;
  const float z0 = c.v0 ? r.wA : 0.0f, z1 = c.v1 ? r.wA : 0.0f;
  c0 = nw.x * w0 + ne.x * w1;
  c1 = nw.y * w0 + ne.y * w1;
  c2 = nw.z * w0 + ne.z * w1;
  d0 = ne.x * z1 - nw.x * z0;
  d1 = ne.y * z1 - nw.y * z0;
  d2 = ne.z * z1 - nw.z * z0;
  if (NROWS == 2) {
    const float4 sw = lrgb[W + c.i0], se = lrgb[W + c.i1];
    const float u0 = c.m0 * r.wB, u1 = c.m1 * r.wB;
    const float y0 = c.v0 ? r.wB : 0.0f, y1 = c.v1 ? r.wB : 0.0f;
    c0 += sw.x * u0 + se.x * u1;
    c1 += sw.y * u0 + se.y * u1;
    c2 += sw.z * u0 + se.z * u1;
    d0 += se.x * y1 - sw.x * y0;
    d1 += se.y * y1 - sw.y * y0;
    d2 += se.z * y1 - sw.z * y0;
  }
}

// Stage the live source colour rows of image b into LDS as float4.
template <int NROWS>
__device__ __forceinline__ void stage_colour_rows(float4* __restrict__ lrgb, const float* __restrict__ srcb, int HW,
                                                  int W, const RowSel& r) {
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const float* p = srcb + (long)r.yA * W + x;
    lrgb[x] = make_float4(p[0], p[HW], p[2 * HW], 0.0f);
    if (NROWS == 2) {
      const float* q = srcb + (long)r.yB * W + x;
      lrgb[W + x] = make_float4(q[0], q[HW], q[2 * HW], 0.0f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------------------------
// LDS layout shared by both kernels: float4 colour[2*W] | float sdisp[N] (= sign * disparity of each plane) | ...
template <bool MIX, bool HASMASK, int NROWS, int U>
__device__ __forceinline__ void fwd_group(const SweepArgs& a, const RowSel& row, const float4* __restrict__ lrgb,
                                          const float* __restrict__ sdisp, int b, int n0, int x, int pix, int HW,
                                          float Wm1, float rcpWm1, float t0, float t1, float t2, float ea,
                                          bool automask, FwdAcc& acc, uint32_t& bits, float* __restrict__ stash) {
  ColTap ct[U];
  Taps<NROWS> tl[U], ts[U];
  float mval[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {  // issue every global load of the group before the first use
    const int n = n0 + u;
    const long pl = ((long)b * a.N + n) * HW;               // workgroup-uniform
    const float* lA = a.logits + pl + (long)row.yA * a.W;   // uniform row pointers: scalar base + 32-bit lane offset
    const float* lB = a.logits + pl + (long)row.yB * a.W;
    ct[u] = make_col_tap((float)x + sdisp[n], Wm1, rcpWm1, a.W);
    mval[u] = HASMASK ? (a.padding_mask + pl)[(unsigned)pix] : 1.0f;
    tl[u] = load_taps<NROWS>(lA, lB, ct[u]);
    if (MIX) ts[u] = load_taps<NROWS>(a.sigma + pl + (long)row.yA * a.W, a.sigma + pl + (long)row.yB * a.W, ct[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = n0 + u;
    if (HASMASK) {  // rec_features * padding_mask (trainer.py:580): a masked plane samples as all-zero features,
      const bool mk = mval[u] != 0.0f;  // i.e. every tap weight of the plane is zero
      if (mk) bits |= 1u << (n & 31);
      if ((n & 31) == 31 || n == a.N - 1) {
        stash[((long)b * a.stash_k + kStashBase + (n >> 5)) * HW + pix] = __uint_as_float(bits);
        bits = 0;
      }
      ct[u].m0 = mk ? ct[u].m0 : 0.0f;
      ct[u].m1 = mk ? ct[u].m1 : 0.0f;
    }
    const float l = tap_value<NROWS>(tl[u], row, ct[u]);
    const float s = MIX ? tap_value<NROWS>(ts[u], row, ct[u]) : 0.0f;
    float c0, c1, c2;
    colour_taps<NROWS>(lrgb, a.W, row, ct[u], c0, c1, c2);
    fwd_accumulate<MIX>(acc, l, s, c0, c1, c2, t0, t1, t2, ea, automask);
  }
}

template <bool MIX, bool HASMASK, int NROWS>
__device__ __forceinline__ void rowshift_fwd_body(const SweepArgs& a, const RowSel& row, float4* lrgb, float* sdisp,
                                                  float* __restrict__ rgb_rec, float* __restrict__ ph_map,
                                                  float* __restrict__ stash) {
  constexpr int U = (NROWS == 1) ? 4 : 2;
  const int y = blockIdx.x, b = blockIdx.y;
  const int HW = a.H * a.W;
  const bool automask = a.flags & PD_AUTOMASK;
  const float Wm1 = (float)(a.W - 1), rcpWm1 = refined_rcp(Wm1);
  const float* srcb = a.src + (long)b * 3 * HW;
  stage_colour_rows<NROWS>(lrgb, srcb, HW, a.W, row);
  for (int i = threadIdx.x; i < a.N; i += blockDim.x) sdisp[i] = a.sign * a.plane[(long)b * a.N + i];
  __syncthreads();
  for (int x = threadIdx.x; x < a.W; x += blockDim.x) {
    const int pix = y * a.W + x;
    const float t0 = a.tgt[((long)b * 3 + 0) * HW + pix];
    const float t1 = a.tgt[((long)b * 3 + 1) * HW + pix];
    const float t2 = a.tgt[((long)b * 3 + 2) * HW + pix];
    float ea = 0.0f;  // 3 x identity-reprojection error
    if (automask) ea = fabsf(srcb[pix] - t0) + fabsf(srcb[HW + pix] - t1) + fabsf(srcb[2 * HW + pix] - t2);
    FwdAcc acc;
    uint32_t bits = 0;
    int n = 0;
    for (; n + U <= a.N; n += U)
      fwd_group<MIX, HASMASK, NROWS, U>(a, row, lrgb, sdisp, b, n, x, pix, HW, Wm1, rcpWm1, t0, t1, t2, ea, automask,
                                         acc, bits, stash);
    for (; n < a.N; ++n)
      fwd_group<MIX, HASMASK, NROWS, 1>(a, row, lrgb, sdisp, b, n, x, pix, HW, Wm1, rcpWm1, t0, t1, t2, ea, automask,
                                         acc, bits, stash);
    const FwdResult r = fwd_finish<MIX>(acc, t0, t1, t2, ea, automask);
    float* st = stash + (long)b * a.stash_k * HW + pix;
    st[0] = r.lse2;
    st[HW] = r.Sn;
    st[2 * HW] = r.mx;
    st[3 * HW] = r.sel;
    rgb_rec[((long)b * 3 + 0) * HW + pix] = r.r0;
    rgb_rec[((long)b * 3 + 1) * HW + pix] = r.r1;
    rgb_rec[((long)b * 3 + 2) * HW + pix] = r.r2;
    ph_map[(long)b * HW + pix] = r.ph;
  }
}

template <bool MIX, bool HASMASK>
__global__ __launch_bounds__(kMaxRowThreads) void rowshift_fwd_kernel(SweepArgs a, float* __restrict__ rgb_rec,
                                                                     float* __restrict__ ph_map,
                                                                     float* __restrict__ stash) {
  extern __shared__ float4 lds4[];
  float* sdisp = reinterpret_cast<float*>(lds4 + 2 * a.W);
  const RowSel row = make_row_sel(blockIdx.x, a.H);
  if (row.nrows == 2) rowshift_fwd_body<MIX, HASMASK, 2>(a, row, lds4, sdisp, rgb_rec, ph_map, stash);
  else                rowshift_fwd_body<MIX, HASMASK, 1>(a, row, lds4, sdisp, rgb_rec, ph_map, stash);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward: one workgroup per target row; 64-lane segments of the row; gather-form adjoint.
// ---------------------------------------------------------------------------------------------------------------
// Segment-boundary records live in LDS, zero-filled at kernel start: bnd[(seg*N + n)*6 + tensor*3 + j] with
// j = 0: to global slot T0-1, j = 1: to slot Tlast+1, j = 2: to slot Tlast+2 (the row is a ring of W slots).
//
// Route lane contributions (c0 -> slot lane+dl, c1 -> slot lane+dl+1) to their slots inside the wave and return this
// lane's slot total.  `last` = last active lane of the segment; `bp` = this (segment, plane, tensor)'s 3 records.
__device__ __forceinline__ float route(float c0, float c1, int dl, bool regular, int lane, int last,
                                       float* __restrict__ bp) {
  if (regular) {  // every lane's left tap is exactly its own slot: one wave-wide shift
    if (lane == last) bp[1] = c1;  // leaves the segment on the right
    return c0 + wave_shift_up1(c1);
  }
  float out = 0.0f;
#pragma unroll
  for (int r = -2; r <= 1; ++r) {
    const int srcl = lane + r;
    const int sl = min(max(srcl, 0), kWave - 1);
    const float v0 = __shfl(c0, sl, kWave), v1 = __shfl(c1, sl, kWave);
    const int dd = __shfl(dl, sl, kWave);
    const bool in = (srcl >= 0) && (srcl <= last);
    if (in && dd == -r) out += v0;       // c0 of lane+r lands on slot lane+r+dd == lane
    if (in && dd == -r - 1) out += v1;   // c1 of lane+r lands on slot lane+r+dd+1 == lane
  }
  const float c0_first = __shfl(c0, 0, kWave);
  const int d_first = __shfl(dl, 0, kWave);
  const float c0_last = __shfl(c0, last, kWave), c1_last = __shfl(c1, last, kWave);
  const int d_last = __shfl(dl, last, kWave);
  const int lp = max(last - 1, 0);
  const float c1_prev = __shfl(c1, lp, kWave);
  const int d_prev = __shfl(dl, lp, kWave);
  if (lane == 0) {
    bp[0] = (d_first == -1) ? c0_first : 0.0f;                                      // slot -1
    bp[1] = ((d_last == 1) ? c0_last : 0.0f) + ((d_last == 0) ? c1_last : 0.0f) +
            ((last >= 1 && d_prev == 1) ? c1_prev : 0.0f);                          // slot last+1
    bp[2] = (d_last == 1) ? c1_last : 0.0f;                                         // slot last+2
  }
  return out;
}

struct SegCtx {
  int seg, T0, xt, last, lane, pix;
  bool active;
};

template <bool MIX, bool HASMASK, int NROWS, int U>
__device__ __forceinline__ void bwd_group(const SweepArgs& a, const BwdOut& o, const RowSel& row,
                                          const float4* __restrict__ lrgb, const float* __restrict__ sdisp,
                                          const int* __restrict__ kshift, float* __restrict__ red,
                                          float* __restrict__ bnd, int b, int y, int n0,
                                          const SegCtx& sc, const PixelCtx& c, int HW, float Wm1, float rcpWm1,
                                          float gix_scale, bool want_plane, uint32_t& bits) {
  const int W = a.W, N = a.N;
  ColTap ct[U];
  Taps<NROWS> tl[U], ts[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = n0 + u;
    const long pl = ((long)b * N + n) * HW;
    ct[u] = make_col_tap((float)sc.xt + sdisp[n], Wm1, rcpWm1, W);
    tl[u] = load_taps<NROWS>(a.logits + pl + (long)row.yA * W, a.logits + pl + (long)row.yB * W, ct[u]);
    if (MIX) ts[u] = load_taps<NROWS>(a.sigma + pl + (long)row.yA * W, a.sigma + pl + (long)row.yB * W, ct[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = n0 + u;
    const long pl = ((long)b * N + n) * HW;
    const int k = kshift[n];  // nominal shift floor(s*d), |k| <= W
    bool mk = sc.active;
    if (HASMASK) {
      if ((n & 31) == 0)
        bits = __float_as_uint((o.stash + ((long)b * a.stash_k + kStashBase + (n >> 5)) * HW)[(unsigned)sc.pix]);
      mk = sc.active && ((bits >> (n & 31)) & 1u);
    }
    const ColTap& t = ct[u];
    float c0, c1, c2, d0x, d1x, d2x;
    colour_taps_dx<NROWS>(lrgb, W, row, t, c0, c1, c2, d0x, d1x, d2x);
    float l = tap_value<NROWS>(tl[u], row, t);
    float s = MIX ? tap_value<NROWS>(ts[u], row, t) : 0.0f;
    const PlaneGrad pg = plane_grad<MIX>(c, l, s, c0, c1, c2);
    const float live = mk ? row.wy_main : 0.0f;   // padding mask x vertical adjoint weight of the own row
    const int dl = (mk && (t.v0 || t.v1)) ? t.x0 - sc.xt - k : 0;
    const float gl = pg.g_l * live, gs = pg.g_s * live;
    const float cl0 = gl * t.m0, cl1 = gl * t.m1, cs0 = gs * t.m0, cs1 = gs * t.m1;
    float gd = 0.0f;
    if (want_plane) {
      const float dlx = tap_dx<NROWS>(tl[u], row, t);
      const float dsx = MIX ? tap_dx<NROWS>(ts[u], row, t) : 0.0f;
      gd = (pg.g_l * dlx + pg.g_s * dsx + pg.gc0 * d0x + pg.gc1 * d1x + pg.gc2 * d2x) * gix_scale;
      gd = mk ? gd : 0.0f;
    }
    const bool regular = __all(dl == 0);
    int xs = sc.xt + k;                       // the source pixel this slot owns (ring of W slots)
    xs = (xs >= W) ? xs - W : ((xs < 0) ? xs + W : xs);
    float* bp = bnd + ((long)sc.seg * N + n) * 6;
    const float out_l = route(cl0, cl1, dl, regular, sc.lane, sc.last, bp);
    if (sc.active && o.g_logits) (o.g_logits + pl + (long)y * W)[(unsigned)xs] = out_l;
    if (MIX) {
      const float out_s = route(cs0, cs1, dl, regular, sc.lane, sc.last, bp + 3);
      if (sc.active && o.g_sigma) (o.g_sigma + pl + (long)y * W)[(unsigned)xs] = out_s;
    }
    if (want_plane) {
      const float v = wave_sum_hi(gd);
      if (sc.lane == kWave - 1) atomicAdd(&red[n], v);
    }
  }
}

template <bool MIX, bool HASMASK, int NROWS>
__device__ __forceinline__ void rowshift_bwd_body(const SweepArgs& a, const BwdOut& o, const RowSel& row,
                                                  float* sdisp, int* kshift, float* red, float* bnd, float4* lrgb) {
  constexpr int U = (NROWS == 1) ? 2 : 1;
  const int y = blockIdx.x, b = blockIdx.y;
  const int HW = a.H * a.W, W = a.W, N = a.N;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int nseg = (W + kWave - 1) / kWave;
  const bool want_plane = (o.g_plane != nullptr);
  const float* srcb = a.src + (long)b * 3 * HW;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    red[i] = 0.0f;
    const float sd = a.sign * a.plane[(long)b * N + i];
    sdisp[i] = sd;
    kshift[i] = (int)fminf(fmaxf(floorf(sd), -(float)W), (float)W);
  }
  for (int i = threadIdx.x; i < nseg * N * 6; i += blockDim.x) bnd[i] = 0.0f;
  stage_colour_rows<NROWS>(lrgb, srcb, HW, W, row);
  __syncthreads();
  const float Wm1 = (float)(W - 1), rcpWm1 = refined_rcp(Wm1);
  const float gix_scale = (Wm1 / 2) * 2.0f / Wm1 * a.sign;  // d ix / d disp through un-normalise, *2, /(W-1)

  for (int seg = wave; seg < nseg; seg += nwaves) {
    SegCtx sc;
    sc.seg = seg;
    sc.T0 = seg * kWave;
    sc.xt = sc.T0 + lane;
    sc.lane = lane;
    sc.active = sc.xt < W;
    sc.last = min(kWave - 1, W - 1 - sc.T0);
    sc.pix = y * W + (sc.active ? sc.xt : 0);
    const PixelCtx c = sc.active ? make_pixel_ctx<MIX>(a, o, b, sc.pix, HW) : zero_pixel_ctx();
    uint32_t bits = 0;
    int n = 0;
    for (; n + U <= N; n += U)
      bwd_group<MIX, HASMASK, NROWS, U>(a, o, row, lrgb, sdisp, kshift, red, bnd, b, y, n, sc, c, HW, Wm1, rcpWm1, gix_scale,
                                         want_plane, bits);
    for (; n < N; ++n)
      bwd_group<MIX, HASMASK, NROWS, 1>(a, o, row, lrgb, sdisp, kshift, red, bnd, b, y, n, sc, c, HW, Wm1, rcpWm1, gix_scale,
                                         want_plane, bits);
  }
  __syncthreads();
  // Deferred segment-boundary contributions: record (seg, n, j) targets global slot g (ring), i.e. source (g+k) mod W.
  const int ntens = MIX ? 2 : 1;
  const int nrec = nseg * N * 3 * ntens;
  for (int i = threadIdx.x; i < nrec; i += blockDim.x) {
    const int j = i % 3, tns = (i / 3) % ntens, n = (i / (3 * ntens)) % N, seg = i / (3 * ntens * N);
    const float v = bnd[((long)seg * N + n) * 6 + tns * 3 + j];
    if (v == 0.0f) continue;
    float* dst = (tns == 0) ? o.g_logits : o.g_sigma;
    if (!dst) continue;
    const int T0 = seg * kWave, last = min(kWave - 1, W - 1 - T0);
    int g = (j == 0) ? T0 - 1 : T0 + last + j;   // j=1 -> last+1, j=2 -> last+2
    g = ((g % W) + W) % W;
    const int k = kshift[n];
    int xs = g + k;
    xs = (xs >= W) ? xs - W : ((xs < 0) ? xs + W : xs);
    unsafeAtomicAdd(dst + ((long)b * N + n) * HW + (long)y * W + xs, v);
  }
  if (want_plane) {
    float* dstp = o.partials + ((long)b * a.H + y) * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) dstp[i] = red[i];
  }
}

template <bool MIX, bool HASMASK>
__global__ __launch_bounds__(kMaxRowThreads) void rowshift_bwd_kernel(SweepArgs a, BwdOut o) {
  extern __shared__ float4 lds4[];
  // LDS: colour rows float4[2*W] | sdisp[N] | kshift[N] | red[N] | bnd[nseg][N][6]  (6 = 3 records x {logits, sigma})
  float* sdisp = reinterpret_cast<float*>(lds4 + 2 * a.W);
  int* kshift = reinterpret_cast<int*>(sdisp + a.N);
  float* red = sdisp + 2 * a.N;
  float* bnd = red + a.N;
  const RowSel row = make_row_sel(blockIdx.x, a.H);
  if (row.nrows == 2) rowshift_bwd_body<MIX, HASMASK, 2>(a, o, row, sdisp, kshift, red, bnd, lds4);
  else                rowshift_bwd_body<MIX, HASMASK, 1>(a, o, row, sdisp, kshift, red, bnd, lds4);
}

// partials [B][R][M] -> out [B][M]; one wave per (b, j): lanes stride over R, then wave-reduce.  Deterministic.
__global__ void reduce_rows_kernel(const float* __restrict__ partials, float* __restrict__ out, int R, int M) {
  const int j = blockIdx.x, b = blockIdx.y;
  const float* p = partials + (long)b * R * M + j;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < R; i += kWave) acc += p[(long)i * M];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * M + j] = acc;
}

// Device self-check used by the tests: div_by(refined reciprocal) == IEEE division, bit for bit.
__global__ void div_check_kernel(float Wm1, int count, float lo, float step, int* __restrict__ mismatches) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float a = lo + step * (float)i;
  const float q_fast = div_by(a, Wm1, refined_rcp(Wm1));
  const float q_ieee = __fdiv_rn(a, Wm1);
  if (__float_as_uint(q_fast) != __float_as_uint(q_ieee)) atomicAdd(mismatches, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// Waves per row-workgroup: a divisor of the number of 64-lane segments (equal work per wave), at most 8, so that
// several workgroups fit a CU at the kernels' register footprint (e.g. W=640: 10 segments -> 5 waves x 2 segments).
static int row_threads(int W) {
  const int nseg = ceil_div(W, kWave);
  int waves = 1;
  for (int w = 1; w <= 8 && w <= nseg; ++w)
    if (nseg % w == 0) waves = w;
  if (waves < 4 && nseg > 8) waves = 8;  // awkward segment counts (primes): accept a ragged last pass
  return waves * kWave;
}

bool rowshift_applicable(const pd_sweep_desc* d) {
  return d->mode == PD_WARP_DISP && !(d->flags & PD_DISP_DENSE) && !(d->flags & PD_RENDER_PROB) && d->H <= 65535 &&
         (size_t)d->W * 32 + ((size_t)3 * d->N + (size_t)ceil_div(d->W, kWave) * d->N * 6) * 4 <= 160 * 1024;
}

size_t rowshift_bwd_workspace_floats(const pd_sweep_desc* d) { return (size_t)d->B * d->H * d->N; }

template <typename K>
static void allow_lds(K kernel, size_t shmem) {
  if (shmem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}

#define PD_ROW_DISPATCH(KERNEL, mix, hasmask, grid, block, shmem, stream, ...)                 \
  do {                                                                                          \
    if (mix) {                                                                                  \
      if (hasmask) { allow_lds(KERNEL<true, true>, shmem);  KERNEL<true, true><<<grid, block, shmem, stream>>>(__VA_ARGS__); }   \
      else         { allow_lds(KERNEL<true, false>, shmem); KERNEL<true, false><<<grid, block, shmem, stream>>>(__VA_ARGS__); }  \
    } else {                                                                                    \
      if (hasmask) { allow_lds(KERNEL<false, true>, shmem);  KERNEL<false, true><<<grid, block, shmem, stream>>>(__VA_ARGS__); } \
      else         { allow_lds(KERNEL<false, false>, shmem); KERNEL<false, false><<<grid, block, shmem, stream>>>(__VA_ARGS__); }\
    }                                                                                           \
  } while (0)

int rowshift_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash,
                 hipStream_t stream) {
  dim3 grid(d->H, d->B), block(row_threads(d->W));
  const size_t shmem = (size_t)d->W * 2 * sizeof(float4) + (size_t)d->N * sizeof(float);
  PD_ROW_DISPATCH(rowshift_fwd_kernel, (d->flags & PD_MIXTURE) != 0, a.has_mask != 0, grid, block, shmem, stream, a,
                  rgb_rec, ph_map, stash);
  return check_launch("rowshift_fwd_kernel");
}

int rowshift_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, hipStream_t stream) {
  dim3 grid(d->H, d->B), block(row_threads(d->W));
  const int nseg = ceil_div(d->W, kWave);
  const size_t shmem = (size_t)d->W * 2 * sizeof(float4) + ((size_t)3 * d->N + (size_t)nseg * d->N * 6) * sizeof(float);
  PD_ROW_DISPATCH(rowshift_bwd_kernel, (d->flags & PD_MIXTURE) != 0, a.has_mask != 0, grid, block, shmem, stream, a, o);
  int rc = check_launch("rowshift_bwd_kernel");
  if (rc || !o.g_plane) return rc;
  reduce_rows_kernel<<<dim3(d->N, d->B), kWave, 0, stream>>>(o.partials, o.g_plane, d->H, d->N);
  return check_launch("reduce_rows_kernel");
}

}  // namespace pd

// Test hook (not part of the public header): counts fast-vs-IEEE division mismatches over `count` samples.
extern "C" int pd_selftest_division(float Wm1, int count, float lo, float step, int* d_mismatches, void* stream) {
  pd::div_check_kernel<<<pd::ceil_div(count, 256), 256, 0, (hipStream_t)stream>>>(Wm1, count, lo, step, d_mismatches);
  return pd::check_launch("div_check_kernel");
}
