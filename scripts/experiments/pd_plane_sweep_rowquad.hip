// Row-shift sweep with FOUR target pixels per lane ("row-quad" kernels): the headline path of BASELINE configs[1]/[2]/[5]
// (disp_warp, per-plane or per-row disparities; reference trainer.py:540-554 + 567-603 + 728-742 and their autograd).
//
// Same algorithm, stash and results as pd_plane_sweep_rowshift.hip (one workgroup per target row, planes streamed with
// an online softmax, gather-form adjoint without atomics or zero-fill) — what changes is the shape of the memory
// traffic.  The one-pixel-per-lane kernels issue an 8-byte buffer load per pixel, plane and tensor and a 4-byte store
// per gradient element; measured on MI355X (scripts/probes/wide_probe.hip, random data, 8x49x192x640):
//     8-byte shifted loads   4.5 TB/s      16+4-byte loads of 5 consecutive floats per lane   7.6 TB/s
//     4-byte ring stores     3.6 TB/s      16-byte stores of 4 consecutive floats per lane    3.9-4.4 TB/s
//     both together          4.0 TB/s      both together                                      5.2-5.9 TB/s
// i.e. the vector memory pipeline is paced by instructions, not bytes, and the round-1 kernels sat at its 8-byte
// ceiling.  Here a lane owns 4 consecutive target pixels: their 5 source taps are ONE 16-byte + ONE 4-byte load per
// tensor row (both taps of all four pixels; out-of-row parts read as zero = padding_mode "zeros"), the colour taps five
// ds_read_b128 from the LDS row, and the four gradients of the lane's slots one 16-byte store.
//
// Exactness.  fp32 noise of the reference's coordinate chain can move floor(ix) by one for single pixels when
// frac(sign * d) is within ~1e-4 of an integer, so "pixel i of the lane taps column x0 + i" is CHECKED per wave and plane
// (a vote); waves where it fails (and the one wave per plane that straddles the left image border when sign < 0) run
// that plane through per-pixel loads and an LDS routing buffer — the general path, rare by construction.
#ifdef PD_EXPERIMENTS   // measured slower than the default kernels: built with -DPD_EXPERIMENTS only (scripts/build_variants.sh)
#include "pd_rowshift_common.h"

namespace pd {

// Q = target pixels per lane (template parameter of everything below: 2 or 4); a wave covers a segment of 64 * Q pixels
constexpr int seg_px(int Q) { return kWave * Q; }
constexpr int kQGuardR = 6;           // zero guard cells right of the colour row (a run of 5 cells may start at W + 1)
constexpr int kQRowCells(int W) { return W + 2 + kQGuardR; }
constexpr int kQMaxWaves = 8;
#ifndef PD_QFWD_Q
#define PD_QFWD_Q 2     // pixels per lane, forward (12-byte loads: 8.1 TB/s in the probe)
#endif
#ifndef PD_QBWD_Q
#define PD_QBWD_Q 2     // pixels per lane, backward
#endif

#ifndef PD_QFWD_U
#define PD_QFWD_U 2     // planes per prefetch group (two-row footprint; one-row rows take twice as many), forward
#endif
#ifndef PD_QBWD_U
#define PD_QBWD_U 2     // same, backward
#endif
#ifndef PD_QSPLIT
#define PD_QSPLIT 0     // 1: one kernel per vertical footprint (launched back to back); 0: both bodies in one kernel
#endif
#ifndef PD_QFWD_OCC1
#define PD_QFWD_OCC1 3  // minimum waves per SIMD the register allocator must leave room for: one-row / two-row kernels
#endif
#ifndef PD_QFWD_OCC2
#define PD_QFWD_OCC2 2
#endif
#ifndef PD_QBWD_OCC1
#define PD_QBWD_OCC1 2
#endif
#ifndef PD_QBWD_OCC2
#define PD_QBWD_OCC2 2
#endif

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f buf_load4(Rsrc r, unsigned byte_off) {  // 16 bytes at any 4-byte-aligned offset
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, PD_LOAD_AUX));
}
__device__ __forceinline__ void buf_store4(Rsrc r, unsigned byte_off, v4f v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, (int)byte_off, 0, PD_STORE_AUX);
}

typedef float v3f __attribute__((ext_vector_type(3)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v3f buf_load3(Rsrc r, unsigned byte_off) {
  return __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, (int)byte_off, 0, PD_LOAD_AUX));
}
__device__ __forceinline__ void buf_store2(Rsrc r, unsigned byte_off, v2f v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)byte_off, 0, PD_STORE_AUX);
}

template <int Q> struct Run { float v[Q + 1]; };   // columns x0 .. x0+Q of one tensor row: both taps of the lane's Q pixels

template <int Q>
__device__ __forceinline__ Run<Q> load_run(Rsrc r, unsigned off) {
  Run<Q> o;
  static_assert(Q == 2 || Q == 4, "2 or 4 pixels per lane");
  if (Q == 4) {          // 16 + 4 bytes
    const v4f a = buf_load4(r, off);
    o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w;
    o.v[Q] = buf_load(r, off + 16u);
  } else {               // 12 bytes
    const v3f a = buf_load3(r, off);
    o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z;
  }
  return o;
}
// Q consecutive floats of a row (targets, stash, upstream gradients) and the matching store
template <int Q> struct Vec { float v[Q]; };
template <int Q>
__device__ __forceinline__ Vec<Q> load_vec(Rsrc r, unsigned off) {
  Vec<Q> o;
  if (Q == 4) { const v4f a = buf_load4(r, off); o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w; }
  else        { const v2f a = buf_load2(r, off); o.v[0] = a.x; o.v[1] = a.y; }
  return o;
}
template <int Q>
__device__ __forceinline__ void store_vec(Rsrc r, unsigned off, const float (&v)[Q]) {
  if (Q == 4) buf_store4(r, off, v4f{v[0], v[1], v[Q == 4 ? 2 : 0], v[Q - 1]});
  else        buf_store2(r, off, v2f{v[0], v[1]});
}

// Sampling columns of the lane's four pixels on one plane.
template <int Q>
struct QCoord {
  int x0[Q];
  float w0[Q], w1[Q];
};
template <int Q>
__device__ __forceinline__ QCoord<Q> make_qcoord(int xt0, float sd, float Wm1, float rcpWm1) {
  QCoord<Q> c;
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const ColTap t = make_col_tap((float)(xt0 + i) + sd, Wm1, rcpWm1);
    c.x0[i] = t.x0; c.w0[i] = t.w0; c.w1[i] = t.w1;
  }
  return c;
}
// "pixel i taps columns x0[0] + i, x0[0] + i + 1" for every pixel of the lane that exists, and the run starts inside
// or right of the row (a run that starts left of column 0 would read as all-zero: hardware range check)
template <int Q>
__device__ __forceinline__ bool lane_regular(const QCoord<Q>& c, int nact) {
  bool r = c.x0[0] >= 0;
#pragma unroll
  for (int i = 1; i < Q; ++i) r = r && (i >= nact || c.x0[i] == c.x0[0] + i);
  return r;
}

// General-path loops handle one pixel per iteration at index 0 and then rotate the lane's per-pixel arrays by one: after
// Q iterations everything is back in place.  (A rolled loop with a dynamic index would put the arrays in scratch
// memory; an unrolled one lets the scheduler hoist all four pixels' loads and blows the register budget of the main path.)
template <typename T, int Q>
__device__ __forceinline__ void rot4(T (&a)[Q]) {
  const T t = a[0];
#pragma unroll
  for (int i = 0; i + 1 < Q; ++i) a[i] = a[i + 1];
  a[Q - 1] = t;
}

// colour rows in LDS: float4 cells for columns -2 .. W + kQGuardR - 1
__device__ __forceinline__ unsigned qcell(int col) { return (unsigned)(col + 2) << 4; }

template <int NROWS>
__device__ __forceinline__ void stage_quad_rows(const SweepArgs& a, int b, const RowSel& r, float4* __restrict__ lrgb,
                                                float* __restrict__ sdisp, int yrow) {
  const int W = a.W, HW = a.H * a.W, RS = kQRowCells(W);
  const float* srcb = a.src + (long)b * 3 * HW;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const float* p = srcb + (long)r.yA * W + x;
    lrgb[2 + x] = make_float4(p[0], p[HW], p[2 * HW], 0.0f);
    if (NROWS == 2) {
      const float* q = srcb + (long)r.yB * W + x;
      lrgb[RS + 2 + x] = make_float4(q[0], q[HW], q[2 * HW], 0.0f);
    }
  }
  if (threadIdx.x < 2 + kQGuardR) {
    const int g = (threadIdx.x < 2) ? threadIdx.x : W + threadIdx.x;   // cells 0,1 and W+2 .. W+1+kQGuardR
    lrgb[g] = z;
    if (NROWS == 2) lrgb[RS + g] = z;
  }
  const float lim = (float)(W + 2);
  for (int i = threadIdx.x; i < a.N; i += blockDim.x) {
    const long di = (a.flags & PD_DISP_ROWS) ? ((long)b * a.N + i) * a.H + yrow : (long)b * a.N + i;
    const float sd = a.sign * a.plane[di];
    const bool masked = a.mask_rows && a.mask_rows[((long)b * a.N + i) * a.H + yrow] == 0.0f;   // see stage_row_constants
    sdisp[i] = (!masked && sd >= -lim && sd <= lim) ? sd : ((sd < 0.0f && !masked) ? -lim : lim);
  }
}

// The two colour taps of ONE pixel (general path), from the quad layout
struct CPair { float4 n0, n1, s0, s1; };
template <int NROWS>
__device__ __forceinline__ CPair load_cpair(const char* __restrict__ lrgb, int W, int x0) {
  CPair c;
  const unsigned off = qcell(min(max(x0, -2), W));
  c.n0 = *reinterpret_cast<const float4*>(lrgb + off);
  c.n1 = *reinterpret_cast<const float4*>(lrgb + off + 16);
  if (NROWS == 2) {
    const unsigned rb = (unsigned)kQRowCells(W) << 4;
    c.s0 = *reinterpret_cast<const float4*>(lrgb + rb + off);
    c.s1 = *reinterpret_cast<const float4*>(lrgb + rb + off + 16);
  }
  return c;
}

// One plane's prefetched global data for the lane's four pixels
// (Only the loaded runs live across the prefetch distance: the sampling columns and weights are recomputed when the
// plane is reduced — 40 VALU operations per plane against 12 registers per plane in flight.)
template <int NROWS, int Q>
struct QPlane {
  Run<Q> lA, sA, lB, sB;   // logits / sigma runs of source rows A (and B)
};

template <bool MIX, int NROWS, int Q>
__device__ __forceinline__ void qplane_issue(QPlane<NROWS, Q>& g, const SweepArgs& a, const RowSel& row,
                                             const float* __restrict__ sdisp, int b, int n, int xt0, int HW, float Wm1,
                                             float rcpWm1) {
  const int x00 = make_col_tap((float)xt0 + sdisp[n], Wm1, rcpWm1).x0;
  const unsigned off = (unsigned)max(x00, 0) << 2;   // irregular lanes reload per pixel; this keeps the issue branch-free
  const float* pl = plane_ptr(a.logits + (long)b * a.N * HW, n, HW);
  g.lA = load_run<Q>(row_rsrc(pl + (long)row.yA * a.W, a.W), off);
  if (NROWS == 2) g.lB = load_run<Q>(row_rsrc(pl + (long)row.yB * a.W, a.W), off);
  if (MIX) {
    const float* ps = plane_ptr(a.sigma + (long)b * a.N * HW, n, HW);
    g.sA = load_run<Q>(row_rsrc(ps + (long)row.yA * a.W, a.W), off);
    if (NROWS == 2) g.sB = load_run<Q>(row_rsrc(ps + (long)row.yB * a.W, a.W), off);
  }
}

// Sampled (logit, sigma, colour) of pixel i and, for the backward, their x-derivatives — fast path: from the runs
template <bool MIX, int NROWS, int Q>
__device__ __forceinline__ void qsample_fast(const QPlane<NROWS, Q>& g, const QCoord<Q>& ct, const float4* cA, const float4* cB,
                                             const RowSel& row, int i, float& l, float& s, float& c0, float& c1, float& c2) {
  const float w0 = ct.w0[i], w1 = ct.w1[i];
  if (NROWS == 1) {
    l = g.lA.v[i] * w0 + g.lA.v[i + 1] * w1;
    s = MIX ? g.sA.v[i] * w0 + g.sA.v[i + 1] * w1 : 0.0f;
    c0 = cA[i].x * w0 + cA[i + 1].x * w1;
    c1 = cA[i].y * w0 + cA[i + 1].y * w1;
    c2 = cA[i].z * w0 + cA[i + 1].z * w1;
  } else {
    const float a0 = w0 * row.wA, a1 = w1 * row.wA, b0 = w0 * row.wB, b1 = w1 * row.wB;
    l = g.lA.v[i] * a0 + g.lA.v[i + 1] * a1 + g.lB.v[i] * b0 + g.lB.v[i + 1] * b1;
    s = MIX ? g.sA.v[i] * a0 + g.sA.v[i + 1] * a1 + g.sB.v[i] * b0 + g.sB.v[i + 1] * b1 : 0.0f;
    c0 = cA[i].x * a0 + cA[i + 1].x * a1 + cB[i].x * b0 + cB[i + 1].x * b1;
    c1 = cA[i].y * a0 + cA[i + 1].y * a1 + cB[i].y * b0 + cB[i + 1].y * b1;
    c2 = cA[i].z * a0 + cA[i + 1].z * a1 + cB[i].z * b0 + cB[i + 1].z * b1;
  }
}

// General path: one pixel from its own taps (per-pixel loads; the x0 = -1 fix-up of the one-pixel kernels)
template <int NROWS>
struct PixTaps { Taps<NROWS> tl, ts; CPair c; };

template <bool MIX, int NROWS>
__device__ __forceinline__ PixTaps<NROWS> load_pixel(const SweepArgs& a, const RowSel& row, const char* __restrict__ lrgb,
                                                     int b, int n, int HW, int x0) {
  PixTaps<NROWS> p;
  ColTap ct; ct.x0 = x0; ct.w0 = ct.w1 = 0.0f;
  const TapPos tp = tap_pos(ct);
  const float* pl = plane_ptr(a.logits + (long)b * a.N * HW, n, HW);
  p.tl = load_taps<NROWS>(row_rsrc_uniform(pl + (long)row.yA * a.W, a.W), row_rsrc_uniform(pl + (long)row.yB * a.W, a.W), tp);
  fix_edge<NROWS>(p.tl, tp.edge);
  if (MIX) {
    const float* ps = plane_ptr(a.sigma + (long)b * a.N * HW, n, HW);
    p.ts = load_taps<NROWS>(row_rsrc_uniform(ps + (long)row.yA * a.W, a.W), row_rsrc_uniform(ps + (long)row.yB * a.W, a.W), tp);
    fix_edge<NROWS>(p.ts, tp.edge);
  } else {
    p.ts = p.tl;
  }
  p.c = load_cpair<NROWS>(lrgb, a.W, x0);
  return p;
}

template <bool MIX, int NROWS>
__device__ __forceinline__ void qsample_pixel(const PixTaps<NROWS>& p, const RowSel& row, float w0, float w1, float& l,
                                              float& s, float& c0, float& c1, float& c2) {
  const float a0 = (NROWS == 1) ? w0 : w0 * row.wA, a1 = (NROWS == 1) ? w1 : w1 * row.wA;
  l = p.tl.a0 * a0 + p.tl.a1 * a1;
  s = MIX ? p.ts.a0 * a0 + p.ts.a1 * a1 : 0.0f;
  c0 = p.c.n0.x * a0 + p.c.n1.x * a1;
  c1 = p.c.n0.y * a0 + p.c.n1.y * a1;
  c2 = p.c.n0.z * a0 + p.c.n1.z * a1;
  if (NROWS == 2) {
    const float b0 = w0 * row.wB, b1 = w1 * row.wB;
    l += p.tl.b0 * b0 + p.tl.b1 * b1;
    if (MIX) s += p.ts.b0 * b0 + p.ts.b1 * b1;
    c0 += p.c.s0.x * b0 + p.c.s1.x * b1;
    c1 += p.c.s0.y * b0 + p.c.s1.y * b1;
    c2 += p.c.s0.z * b0 + p.c.s1.z * b1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------------------------
template <int Q>
struct QTarget { float t0[Q], t1[Q], t2[Q], ea[Q]; };

template <bool MIX, bool AUTO, int NROWS, int Q>
__device__ __forceinline__ void qfwd_plane(const QPlane<NROWS, Q>& g, const SweepArgs& a, const RowSel& row,
                                           const char* __restrict__ lrgb, const float* __restrict__ sdisp, int b, int n,
                                           int HW, int xt0, int nact, float Wm1, float rcpWm1, const QTarget<Q>& tg,
                                           bool automask, FwdAcc (&acc)[Q]) {
  QCoord<Q> ct = make_qcoord<Q>(xt0, sdisp[n], Wm1, rcpWm1);
  const bool fast = __all(lane_regular(ct, nact) ? 1 : 0);
  if (fast) {
    float4 cA[Q + 1], cB[Q + 1];
    const unsigned off = qcell(min(ct.x0[0], a.W + 1));
#pragma unroll
    for (int j = 0; j <= Q; ++j) {
      cA[j] = *reinterpret_cast<const float4*>(lrgb + off + 16 * j);
      if (NROWS == 2) cB[j] = *reinterpret_cast<const float4*>(lrgb + ((unsigned)kQRowCells(a.W) << 4) + off + 16 * j);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      float l, s, c0, c1, c2;
      qsample_fast<MIX, NROWS, Q>(g, ct, cA, cB, row, i, l, s, c0, c1, c2);
      fwd_accumulate<MIX>(acc[i], l, s, c0, c1, c2, tg.t0[i], tg.t1[i], tg.t2[i], tg.ea[i], automask);
    }
  } else {
    QTarget<Q> t = tg;
#pragma unroll 1
    for (int it = 0; it < Q; ++it) {
      const PixTaps<NROWS> p = load_pixel<MIX, NROWS>(a, row, lrgb, b, n, HW, ct.x0[0]);
      float l, s, c0, c1, c2;
      qsample_pixel<MIX, NROWS>(p, row, ct.w0[0], ct.w1[0], l, s, c0, c1, c2);
      fwd_accumulate<MIX>(acc[0], l, s, c0, c1, c2, t.t0[0], t.t1[0], t.t2[0], t.ea[0], automask);
      rot4(acc); rot4(ct.x0); rot4(ct.w0); rot4(ct.w1); rot4(t.t0); rot4(t.t1); rot4(t.t2); rot4(t.ea);
    }
  }
}

// row descriptor of plane-like tensors [B][C][H][W]: row (b, c, y)
__device__ __forceinline__ Rsrc chan_row(const float* base, int b, int C, int c, int y, int H, int W) {
  return row_rsrc_uniform(base + (((long)b * C + c) * H + y) * W, W);
}

template <bool MIX, bool AUTO, int NROWS, int Q>
__device__ __forceinline__ float rowquad_fwd_body(const SweepArgs& a, const RowSel& row, float4* lrgb, float* sdisp,
                                                  float* __restrict__ rgb_rec, float* __restrict__ ph_map,
                                                  float* __restrict__ stash) {
  constexpr int U = (NROWS == 1) ? 2 * PD_QFWD_U : PD_QFWD_U;   // the same bytes in flight for both footprints
  const int y = block_row(wg_rowid(a.B, a.H), a.H), b = wg_image(a.B, a.H);
  const int HW = a.H * a.W, N = a.N, W = a.W, H = a.H;
  const bool automask = MIX ? AUTO : (bool)(a.flags & PD_AUTOMASK);
  const float Wm1 = (float)(W - 1), rcpWm1 = refined_rcp(Wm1);
  stage_quad_rows<NROWS>(a, b, row, lrgb, sdisp, y);
  __syncthreads();
  const char* lbytes = reinterpret_cast<const char*>(lrgb);
  const int lane = threadIdx.x & (kWave - 1), nwaves = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (W + seg_px(Q) - 1) / seg_px(Q);
  float ph_sum = 0.0f;
  for (int seg = wave; seg < nseg; seg += nwaves) {
    const int xt0 = seg * seg_px(Q) + lane * Q;
    const int nact = min(max(W - xt0, 0), Q);   // pixels of this lane inside the row
    const unsigned xoff = (unsigned)xt0 << 2;
    QTarget<Q> tg;
    {
      const Vec<Q> q0 = load_vec<Q>(chan_row(a.tgt, b, 3, 0, y, H, W), xoff), q1 = load_vec<Q>(chan_row(a.tgt, b, 3, 1, y, H, W), xoff),
                   q2 = load_vec<Q>(chan_row(a.tgt, b, 3, 2, y, H, W), xoff);
#pragma unroll
      for (int i = 0; i < Q; ++i) { tg.t0[i] = q0.v[i]; tg.t1[i] = q1.v[i]; tg.t2[i] = q2.v[i]; tg.ea[i] = 0.0f; }
      if (automask) {
        const Vec<Q> s0 = load_vec<Q>(chan_row(a.src, b, 3, 0, y, H, W), xoff), s1 = load_vec<Q>(chan_row(a.src, b, 3, 1, y, H, W), xoff),
                     s2 = load_vec<Q>(chan_row(a.src, b, 3, 2, y, H, W), xoff);
#pragma unroll
        for (int i = 0; i < Q; ++i) tg.ea[i] = fabsf(s0.v[i] - q0.v[i]) + fabsf(s1.v[i] - q1.v[i]) + fabsf(s2.v[i] - q2.v[i]);
      }
    }
    FwdAcc acc[Q];
    QPlane<NROWS, Q> g0[U], g1[U];
    const int nfull = N / U;
#define PD_QI(GR, I)                                                                                         \
  _Pragma("unroll") for (int u = 0; u < U; ++u)                                                              \
      qplane_issue<MIX, NROWS, Q>(GR[u], a, row, sdisp, b, (I) * U + u, xt0, HW, Wm1, rcpWm1)
#define PD_QC(GR, I)                                                                                         \
  _Pragma("unroll") for (int u = 0; u < U; ++u)                                                              \
      qfwd_plane<MIX, AUTO, NROWS, Q>(GR[u], a, row, lbytes, sdisp, b, (I) * U + u, HW, xt0, nact, Wm1, rcpWm1, tg, automask, acc)
    int gi = 0;
    if (nfull > 0) { PD_QI(g0, 0); }
    for (; gi + 2 <= nfull; gi += 2) {
      PD_QI(g1, gi + 1);
      PD_QC(g0, gi);
      PD_QI(g0, min(gi + 2, nfull - 1));   // unconditional: keeps the compiler's vmcnt bookkeeping exact (rowshift notes)
      PD_QC(g1, gi + 1);
    }
    if (gi < nfull) { PD_QC(g0, gi); }
#undef PD_QI
#undef PD_QC
    for (int n = nfull * U; n < N; ++n) {
      QPlane<NROWS, Q> gr;
      qplane_issue<MIX, NROWS, Q>(gr, a, row, sdisp, b, n, xt0, HW, Wm1, rcpWm1);
      qfwd_plane<MIX, AUTO, NROWS, Q>(gr, a, row, lbytes, sdisp, b, n, HW, xt0, nact, Wm1, rcpWm1, tg, automask, acc);
    }
    // results of the lane's four pixels: 16-byte stores (the row descriptors drop what lies beyond the row)
    float r0[Q], r1[Q], r2[Q], ph[Q], lse[Q], sn[Q], mx[Q], sel[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const FwdResult r = fwd_finish<MIX>(acc[i], tg.t0[i], tg.t1[i], tg.t2[i], tg.ea[i], automask);
      r0[i] = r.r0; r1[i] = r.r1; r2[i] = r.r2; ph[i] = r.ph; lse[i] = r.lse2; sn[i] = r.Sn; mx[i] = r.mx; sel[i] = r.sel;
      if (i < nact) ph_sum += r.ph;
    }
    store_vec<Q>(chan_row(rgb_rec, b, 3, 0, y, H, W), xoff, r0);
    store_vec<Q>(chan_row(rgb_rec, b, 3, 1, y, H, W), xoff, r1);
    store_vec<Q>(chan_row(rgb_rec, b, 3, 2, y, H, W), xoff, r2);
    store_vec<Q>(chan_row(ph_map, b, 1, 0, y, H, W), xoff, ph);
    store_vec<Q>(chan_row(stash, b, a.stash_k, 0, y, H, W), xoff, lse);
    store_vec<Q>(chan_row(stash, b, a.stash_k, 1, y, H, W), xoff, sn);
    store_vec<Q>(chan_row(stash, b, a.stash_k, 2, y, H, W), xoff, mx);
    store_vec<Q>(chan_row(stash, b, a.stash_k, 3, y, H, W), xoff, sel);
  }
  return ph_sum;
}

// One kernel per vertical footprint (NROWS live source rows): the register budget of the common one-row rows is not
// set by the two-row bodies.  Both are launched over all rows; a workgroup whose row is of the other kind retires at once.
template <bool MIX, bool AUTO, int NROWS, int Q>
__global__ __launch_bounds__(kQMaxWaves* kWave, NROWS == 1 ? PD_QFWD_OCC1 : PD_QFWD_OCC2) void rowquad_fwd_kernel(
    SweepArgs a, float* __restrict__ rgb_rec, float* __restrict__ ph_map, float* __restrict__ stash) {
  extern __shared__ float4 lds4[];
  // LDS: colour rows float4[2 * (W + 8)] | sdisp[N] | wave sums [8]
  float* sdisp = reinterpret_cast<float*>(lds4 + 2 * kQRowCells(a.W));
  float* wsum = sdisp + a.N;
  const int y = block_row(wg_rowid(a.B, a.H), a.H);
  const RowSel row = two_row_form(make_row_sel(y, a.H), a.row_eps);
  float ph_sum;
  if (NROWS == 0) {   // both bodies in this kernel
    if (row.nrows == 2) ph_sum = rowquad_fwd_body<MIX, AUTO, 2, Q>(a, row, lds4, sdisp, rgb_rec, ph_map, stash);
    else                ph_sum = rowquad_fwd_body<MIX, AUTO, 1, Q>(a, row, lds4, sdisp, rgb_rec, ph_map, stash);
  } else {
    if ((row.nrows == 2) != (NROWS == 2)) return;   // workgroup-uniform: this row belongs to the other kernel
    ph_sum = rowquad_fwd_body<MIX, AUTO, (NROWS == 0 ? 1 : NROWS), Q>(a, row, lds4, sdisp, rgb_rec, ph_map, stash);
  }
  if (a.ph_mean) {   // fused `.mean()` of trainer.py:742: wave totals -> LDS -> one atomic per workgroup
    const float v = wave_sum_hi(ph_sum);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
      unsafeAtomicAdd(a.ph_mean, t * a.inv_numel);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward
// ---------------------------------------------------------------------------------------------------------------
// Ownership as in the one-pixel kernels: slot xt of the row owns source pixel (xt + k) mod W, k = floor(sign * d) the
// plane's nominal shift; a lane owns four consecutive slots.  Pixel i's two contributions go to the slots of columns
// x0_i and x0_i + 1: in the regular case its own slot and the next one (the next pixel of the lane; the next lane's
// first slot for i = 3: one DPP shift; the segment's boundary record for the last pixel of the segment).
struct QBoundary {
  float* rec;       // LDS  [nseg*N][2 tensors]           slot last+1 of every (segment, plane)
  unsigned* irr;    // LDS  bitmap over (seg, plane): the two rare records were written to `side`
  float* side;      // HBM  [nseg*N][2 tensors][2]        slots T0-1 and last+2 (general path only)
  float* scratch;   // LDS  this wave's routing buffer [2 tensors][seg_px(Q) + 4] (general path only)
};
constexpr int scratch_stride(int Q) { return seg_px(Q) + 4; }

template <int Q>
struct QCtx { PixelCtx c[Q]; };

template <bool MIX, int Q>
__device__ __forceinline__ QCtx<Q> make_qctx(const SweepArgs& a, const BwdOut& o, int b, int y, unsigned xoff, int nact) {
  QCtx<Q> q;
  const int H = a.H, W = a.W;
  const Vec<Q> t0 = load_vec<Q>(chan_row(a.tgt, b, 3, 0, y, H, W), xoff), t1 = load_vec<Q>(chan_row(a.tgt, b, 3, 1, y, H, W), xoff),
               t2 = load_vec<Q>(chan_row(a.tgt, b, 3, 2, y, H, W), xoff);
  const Vec<Q> lse = load_vec<Q>(chan_row(o.stash, b, a.stash_k, 0, y, H, W), xoff), sn = load_vec<Q>(chan_row(o.stash, b, a.stash_k, 1, y, H, W), xoff),
               mx = load_vec<Q>(chan_row(o.stash, b, a.stash_k, 2, y, H, W), xoff), sel = load_vec<Q>(chan_row(o.stash, b, a.stash_k, 3, y, H, W), xoff);
  const Vec<Q> r0 = load_vec<Q>(chan_row(o.rgb_rec, b, 3, 0, y, H, W), xoff), r1 = load_vec<Q>(chan_row(o.rgb_rec, b, 3, 1, y, H, W), xoff),
               r2 = load_vec<Q>(chan_row(o.rgb_rec, b, 3, 2, y, H, W), xoff);
  Vec<Q> g0, g1, g2, gm;
#pragma unroll
  for (int i = 0; i < Q; ++i) g0.v[i] = g1.v[i] = g2.v[i] = gm.v[i] = 0.0f;
  if (o.g_rgb_rec) {
    g0 = load_vec<Q>(chan_row(o.g_rgb_rec, b, 3, 0, y, H, W), xoff);
    g1 = load_vec<Q>(chan_row(o.g_rgb_rec, b, 3, 1, y, H, W), xoff);
    g2 = load_vec<Q>(chan_row(o.g_rgb_rec, b, 3, 2, y, H, W), xoff);
  }
  if (o.g_ph_map) gm = load_vec<Q>(chan_row(o.g_ph_map, b, 1, 0, y, H, W), xoff);
  const float gmean = o.g_ph_mean ? o.g_ph_mean[0] * a.inv_numel : 0.0f;
#pragma unroll
  for (int i = 0; i < Q; ++i) {   // pd_sweep.h make_pixel_ctx, from the vector loads (pixels beyond the row read as 0)
    PixelCtx c;
    c.t0 = t0.v[i]; c.t1 = t1.v[i]; c.t2 = t2.v[i]; c.lse2 = lse.v[i]; c.mx = mx.v[i];
    const float gp = (sel.v[i] == 0.0f) ? gm.v[i] + gmean : 0.0f;
    c.gr0 = g0.v[i]; c.gr1 = g1.v[i]; c.gr2 = g2.v[i];
    if (!MIX) {
      c.gr0 += gp * sgn(r0.v[i] - c.t0) * (1.0f / 3.0f);
      c.gr1 += gp * sgn(r1.v[i] - c.t1) * (1.0f / 3.0f);
      c.gr2 += gp * sgn(r2.v[i] - c.t2) * (1.0f / 3.0f);
    }
    c.A = MIX ? gp / (c.mx + kLogEps) : 0.0f;
    c.invS = MIX ? 1.0f / sn.v[i] : 1.0f;
    c.gdotr = c.gr0 * r0.v[i] + c.gr1 * r1.v[i] + c.gr2 * r2.v[i];
    // pixels beyond the row: a finite all-zero context, so that everything they compute is an exact zero
    q.c[i] = (i < nact) ? c : zero_pixel_ctx();
  }
  return q;
}

struct QSeg {
  int seg, seg_prev, T0, xt0, lane, nact, nseg_act;   // nact: this lane's live pixels; nseg_act: the segment's
  bool last_lane;                                     // holds the segment's last live pixel
  int i_last;                                         // ... at this index
};

// one pixel's gradient package
struct QPix { float cl0, cl1, cs0, cs1, gd; int dl; };

template <bool MIX>
__device__ __forceinline__ QPix qpixel_grad(const PixelCtx& c, float l, float s, float c0, float c1, float c2, float dlx,
                                            float dsx, float d0x, float d1x, float d2x, int x0, float w0, float w1,
                                            float wy_main, int xt, int k, int W, bool live, float gix_scale,
                                            int want_plane) {
  QPix r;
  const PlaneGrad pg = plane_grad<MIX>(c, l, s, c0, c1, c2);
  const bool v0 = (unsigned)x0 < (unsigned)W, v1 = (unsigned)(x0 + 1) < (unsigned)W;
  const float lv = live ? wy_main : 0.0f;
  const float m0 = v0 ? w0 * lv : 0.0f, m1 = v1 ? w1 * lv : 0.0f;
  r.dl = (live && (v0 || v1)) ? x0 - xt - k : 0;
  r.cl0 = pg.g_l * m0; r.cl1 = pg.g_l * m1; r.cs0 = pg.g_s * m0; r.cs1 = pg.g_s * m1;
  r.gd = 0.0f;
  if (want_plane) {
    const float gd = (pg.g_l * dlx + pg.g_s * dsx + pg.gc0 * d0x + pg.gc1 * d1x + pg.gc2 * d2x) * gix_scale;
    r.gd = live ? gd : 0.0f;
  }
  return r;
}

template <bool MIX, int NROWS, int Q>
__device__ __forceinline__ void qbwd_plane(const QPlane<NROWS, Q>& g, const SweepArgs& a, const BwdOut& o, const RowSel& row,
                                           const char* __restrict__ lrgb, const float* __restrict__ sdisp,
                                           const int* __restrict__ kshift, float* __restrict__ red, const QBoundary& bnd,
                                           int b, int y, int n, const QSeg& sc, const QCtx<Q>& q, int HW, float Wm1,
                                           float rcpWm1, float gix_scale, int want_plane, int gl_bytes, int gs_bytes) {
  const int W = a.W, N = a.N;
  const int k = __builtin_amdgcn_readfirstlane(kshift[n]);
  const float wy_main = (NROWS == 1) ? 1.0f : row.wy_main;
  QPix px[Q];
  QCoord<Q> ct = make_qcoord<Q>(sc.xt0, sdisp[n], Wm1, rcpWm1);
  const bool fast_loads = __all(lane_regular(ct, sc.nact) ? 1 : 0);
  if (fast_loads) {
    float4 cA[Q + 1], cB[Q + 1];
    const unsigned off = qcell(min(ct.x0[0], W + 1));
#pragma unroll
    for (int j = 0; j <= Q; ++j) {
      cA[j] = *reinterpret_cast<const float4*>(lrgb + off + 16 * j);
      if (NROWS == 2) cB[j] = *reinterpret_cast<const float4*>(lrgb + ((unsigned)kQRowCells(W) << 4) + off + 16 * j);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      float l, s, c0, c1, c2;
      qsample_fast<MIX, NROWS, Q>(g, ct, cA, cB, row, i, l, s, c0, c1, c2);
      float dlx = g.lA.v[i + 1] - g.lA.v[i], dsx = MIX ? g.sA.v[i + 1] - g.sA.v[i] : 0.0f;
      float d0x = cA[i + 1].x - cA[i].x, d1x = cA[i + 1].y - cA[i].y, d2x = cA[i + 1].z - cA[i].z;
      if (NROWS == 2) {
        dlx = dlx * row.wA + (g.lB.v[i + 1] - g.lB.v[i]) * row.wB;
        if (MIX) dsx = dsx * row.wA + (g.sB.v[i + 1] - g.sB.v[i]) * row.wB;
        d0x = d0x * row.wA + (cB[i + 1].x - cB[i].x) * row.wB;
        d1x = d1x * row.wA + (cB[i + 1].y - cB[i].y) * row.wB;
        d2x = d2x * row.wA + (cB[i + 1].z - cB[i].z) * row.wB;
      }
      px[i] = qpixel_grad<MIX>(q.c[i], l, s, c0, c1, c2, dlx, dsx, d0x, d1x, d2x, ct.x0[i], ct.w0[i], ct.w1[i],
                               wy_main, sc.xt0 + i, k, W, i < sc.nact, gix_scale, want_plane);
    }
  } else {
    QCtx<Q> qq = q;
    int act[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) act[i] = (i < sc.nact) ? 1 : 0;
#pragma unroll 1
    for (int it = 0; it < Q; ++it) {
      const PixTaps<NROWS> p = load_pixel<MIX, NROWS>(a, row, lrgb, b, n, HW, ct.x0[0]);
      float l, s, c0, c1, c2;
      qsample_pixel<MIX, NROWS>(p, row, ct.w0[0], ct.w1[0], l, s, c0, c1, c2);
      const float dlx = tap_dx<NROWS>(p.tl, row), dsx = MIX ? tap_dx<NROWS>(p.ts, row) : 0.0f;
      float d0x = p.c.n1.x - p.c.n0.x, d1x = p.c.n1.y - p.c.n0.y, d2x = p.c.n1.z - p.c.n0.z;
      if (NROWS == 2) {
        d0x = d0x * row.wA + (p.c.s1.x - p.c.s0.x) * row.wB;
        d1x = d1x * row.wA + (p.c.s1.y - p.c.s0.y) * row.wB;
        d2x = d2x * row.wA + (p.c.s1.z - p.c.s0.z) * row.wB;
      }
      px[0] = qpixel_grad<MIX>(qq.c[0], l, s, c0, c1, c2, dlx, dsx, d0x, d1x, d2x, ct.x0[0], ct.w0[0], ct.w1[0], wy_main,
                               sc.xt0 + it, k, W, act[0] != 0, gix_scale, want_plane);
      rot4(px); rot4(qq.c); rot4(ct.x0); rot4(ct.w0); rot4(ct.w1); rot4(act);
    }
  }
  bool reg = true;
#pragma unroll
  for (int i = 0; i < Q; ++i) reg = reg && (px[i].dl == 0);
  const bool regular = __all(reg ? 1 : 0);
  const int sn = sc.seg * N + n;
  float outl[Q], outs[Q];
  if (regular) {
    const float inl = wave_shift_up1(px[Q - 1].cl1);
    const float ins = MIX ? wave_shift_up1(px[Q - 1].cs1) : 0.0f;
    outl[0] = px[0].cl0 + inl;
    outs[0] = px[0].cs0 + ins;
#pragma unroll
    for (int i = 1; i < Q; ++i) { outl[i] = px[i].cl0 + px[i - 1].cl1; outs[i] = px[i].cs0 + px[i - 1].cs1; }
    if (sc.last_lane) {   // the segment's last live pixel hands its right tap to the next segment's first slot
      float hl = px[0].cl1, hs = px[0].cs1;
#pragma unroll
      for (int i = 1; i < Q; ++i) { hl = (sc.i_last == i) ? px[i].cl1 : hl; hs = (sc.i_last == i) ? px[i].cs1 : hs; }
      bnd.rec[sn * 2] = hl;
      if (MIX) bnd.rec[sn * 2 + 1] = hs;
    }
  } else {
    // general routing through this wave's LDS buffer: slot j of the segment lives at scratch[j + 1]
    float* sl = bnd.scratch;
    float* ss = bnd.scratch + scratch_stride(Q);
    const int base = 1 + sc.lane * Q;
#pragma unroll
    for (int i = 0; i < Q; ++i) { sl[base + i] = 0.0f; if (MIX) ss[base + i] = 0.0f; }
    if (sc.lane < 4) {
      const int e = (sc.lane == 0) ? 0 : seg_px(Q) + sc.lane;   // 0 and the three cells right of the segment
      sl[e] = 0.0f; if (MIX) ss[e] = 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int j = base + i + px[i].dl;
      if (px[i].cl0 != 0.0f) lds_add(sl + j, px[i].cl0);
      if (px[i].cl1 != 0.0f) lds_add(sl + j + 1, px[i].cl1);
      if (MIX) {
        if (px[i].cs0 != 0.0f) lds_add(ss + j, px[i].cs0);
        if (px[i].cs1 != 0.0f) lds_add(ss + j + 1, px[i].cs1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < Q; ++i) { outl[i] = sl[base + i]; outs[i] = MIX ? ss[base + i] : 0.0f; }
    if (sc.lane == 0) {
      float* side2 = bnd.side + sn * 4;
      side2[0] = sl[0];                      // slot T0 - 1
      bnd.rec[sn * 2] = sl[1 + sc.nseg_act];       // slot last + 1
      side2[1] = sl[2 + sc.nseg_act];              // slot last + 2
      if (MIX) {
        side2[2] = ss[0];
        bnd.rec[sn * 2 + 1] = ss[1 + sc.nseg_act];
        side2[3] = ss[2 + sc.nseg_act];
      }
      atomicOr(bnd.irr + (sn >> 5), 1u << (sn & 31));
    }
    __builtin_amdgcn_wave_barrier();
  }
  // hand-over from the left neighbour segment: if its wave has been here already, take the value now (exchange with 0
  // so that it is added exactly once); whatever arrives later is added by the epilogue
  if (PD_BWD_HANDOVER && sc.lane == 0) {
    float* rp = bnd.rec + (sc.seg_prev * N + n) * 2;
    outl[0] += atomicExch(rp, 0.0f);
    if (MIX) outs[0] += atomicExch(rp + 1, 0.0f);
  }
  // stores: the lane's four slots are four consecutive ring positions; one 16-byte store unless the row's end or the
  // ring's wrap falls inside the quad
  {
    const int xs0 = sc.xt0 + k;   // |k| <= W
    const int xw0 = (xs0 < 0) ? xs0 + W : ((xs0 >= W) ? xs0 - W : xs0);
    const Rsrc rl = row_rsrc_bytes(plane_ptr(o.g_logits + (long)b * N * HW + (long)y * W, n, HW), gl_bytes);
    const Rsrc rs = row_rsrc_bytes(plane_ptr(o.g_sigma + (long)b * N * HW + (long)y * W, n, HW), gs_bytes);
    const bool whole = (sc.nact == Q) && (xw0 + Q <= W) && ((xs0 < 0) == (xs0 + Q - 1 < 0)) && ((xs0 >= W) == (xs0 + Q - 1 >= W));
    if (whole) {
      store_vec<Q>(rl, (unsigned)xw0 << 2, outl);
      if (MIX) store_vec<Q>(rs, (unsigned)xw0 << 2, outs);
    } else {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const int xs = xs0 + i;
        const int xw = (xs < 0) ? xs + W : ((xs >= W) ? xs - W : xs);
        const unsigned off = (i < sc.nact) ? (unsigned)xw << 2 : 0xFFFFFFF0u;
        buf_store(rl, off, outl[i]);
        if (MIX) buf_store(rs, off, outs[i]);
      }
    }
  }
  if (want_plane) {   // disparity gradient of this plane: lane total over its four pixels, wave total into LDS
    float gsum = px[0].gd;
#pragma unroll
    for (int i = 1; i < Q; ++i) gsum += px[i].gd;
    const float v = wave_sum_hi(gsum);
    if (sc.lane == kWave - 1) lds_add(&red[n], v);
  }
}

template <bool MIX, int NROWS, int Q>
__device__ __forceinline__ void rowquad_bwd_body(const SweepArgs& a, const BwdOut& o, const RowSel& row, float* sdisp,
                                                 int* kshift, float* red, QBoundary bnd, float4* lrgb) {
  constexpr int U = (NROWS == 1) ? 2 * PD_QBWD_U : PD_QBWD_U;
  const int y = block_row(bwd_rowid(a.B, a.H), a.H), b = wg_image(a.B, a.H);
  const int HW = a.H * a.W, W = a.W, N = a.N;
  const int lane = threadIdx.x & (kWave - 1), nwaves = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (W + seg_px(Q) - 1) / seg_px(Q);
  const int want_plane = __builtin_amdgcn_readfirstlane(o.g_plane != nullptr ? 1 : 0);
  const int gl_bytes = __builtin_amdgcn_readfirstlane(o.g_logits ? W * 4 : 0);   // 0: the stores become no-ops
  const int gs_bytes = __builtin_amdgcn_readfirstlane(o.g_sigma ? W * 4 : 0);
  stage_quad_rows<NROWS>(a, b, row, lrgb, sdisp, y);
  for (int i = threadIdx.x; i < nseg * N * 2; i += blockDim.x) bnd.rec[i] = 0.0f;
  for (int i = threadIdx.x; i < (nseg * N + 31) / 32; i += blockDim.x) bnd.irr[i] = 0u;
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    red[i] = 0.0f;
    kshift[i] = (int)fminf(fmaxf(floorf(sdisp[i]), -(float)W), (float)W);
  }
  __syncthreads();
  const char* lbytes = reinterpret_cast<const char*>(lrgb);
  const float Wm1 = (float)(W - 1), rcpWm1 = refined_rcp(Wm1);
  const float gix_scale = (Wm1 / 2) * 2.0f / Wm1 * a.sign;
  bnd.scratch += (long)wave * 2 * scratch_stride(Q);
  for (int seg = wave; seg < nseg; seg += nwaves) {
    QSeg sc;
    sc.seg = seg;
    sc.seg_prev = (seg == 0) ? nseg - 1 : seg - 1;
    sc.T0 = seg * seg_px(Q);
    sc.lane = lane;
    sc.xt0 = sc.T0 + lane * Q;
    sc.nact = min(max(W - sc.xt0, 0), Q);
    sc.nseg_act = min(seg_px(Q), W - sc.T0);
    const int p_last = sc.nseg_act - 1;
    sc.last_lane = (lane == p_last / Q);
    sc.i_last = p_last % Q;
    const QCtx<Q> q = make_qctx<MIX, Q>(a, o, b, y, (unsigned)sc.xt0 << 2, sc.nact);
    QPlane<NROWS, Q> g0[U], g1[U];
    const int nfull = N / U;
#define PD_QI(GR, I)                                                                                         \
  _Pragma("unroll") for (int u = 0; u < U; ++u)                                                              \
      qplane_issue<MIX, NROWS, Q>(GR[u], a, row, sdisp, b, (I) * U + u, sc.xt0, HW, Wm1, rcpWm1)
#define PD_QC(GR, I)                                                                                         \
  _Pragma("unroll") for (int u = 0; u < U; ++u)                                                              \
      qbwd_plane<MIX, NROWS, Q>(GR[u], a, o, row, lbytes, sdisp, kshift, red, bnd, b, y, (I) * U + u, sc, q, HW, Wm1, rcpWm1, \
                             gix_scale, want_plane, gl_bytes, gs_bytes)
    int gi = 0;
    if (nfull > 0) { PD_QI(g0, 0); }
    for (; gi + 2 <= nfull; gi += 2) {
      PD_QI(g1, gi + 1);
      PD_QC(g0, gi);
      PD_QI(g0, min(gi + 2, nfull - 1));
      PD_QC(g1, gi + 1);
    }
    if (gi < nfull) { PD_QC(g0, gi); }
#undef PD_QI
#undef PD_QC
    for (int n = nfull * U; n < N; ++n) {
      QPlane<NROWS, Q> gr;
      qplane_issue<MIX, NROWS, Q>(gr, a, row, sdisp, b, n, sc.xt0, HW, Wm1, rcpWm1);
      qbwd_plane<MIX, NROWS, Q>(gr, a, o, row, lbytes, sdisp, kshift, red, bnd, b, y, n, sc, q, HW, Wm1, rcpWm1, gix_scale,
                             want_plane, gl_bytes, gs_bytes);
    }
  }
  __syncthreads();
  // Deferred segment-boundary contributions (as in the one-pixel kernels): record (seg, n, j) targets ring slot g,
  // i.e. source (g + k) mod W.
  const int ntens = MIX ? 2 : 1;
  const int nrec = nseg * N * ntens;
  for (int i = threadIdx.x; i < nrec; i += blockDim.x) {
    const int tns = i % ntens, sn = i / ntens, n = sn % N, seg = sn / N;
    float* dst = (tns == 0) ? o.g_logits : o.g_sigma;
    if (!dst) continue;
    float v[3];
    v[1] = bnd.rec[sn * 2 + tns];
    v[0] = v[2] = 0.0f;
    if ((bnd.irr[sn >> 5] >> (sn & 31)) & 1u) {
      v[0] = bnd.side[sn * 4 + tns * 2];
      v[2] = bnd.side[sn * 4 + tns * 2 + 1];
    }
    const int T0 = seg * seg_px(Q), last = min(seg_px(Q) - 1, W - 1 - T0);
    const int k = kshift[n];
    float* drow = dst + ((long)b * N + n) * HW + (long)y * W;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (v[j] == 0.0f) continue;
      int g = (j == 0) ? T0 - 1 : T0 + last + j;
      g = ((g % W) + W) % W;
      int xs = g + k;
      xs = (xs >= W) ? xs - W : ((xs < 0) ? xs + W : xs);
      unsafeAtomicAdd(drow + xs, v[j]);
    }
  }
  if (want_plane) {
    if (a.flags & PD_DISP_ROWS) {
      for (int i = threadIdx.x; i < N; i += blockDim.x) o.g_plane[((long)b * N + i) * a.H + y] = red[i];
    } else {
      float* dstp = o.partials + ((long)b * a.H + y) * N;
      for (int i = threadIdx.x; i < N; i += blockDim.x) dstp[i] = red[i];
    }
  }
}

template <bool MIX, int NROWS, int Q>
__global__ __launch_bounds__(kQMaxWaves* kWave, NROWS == 1 ? PD_QBWD_OCC1 : PD_QBWD_OCC2) void rowquad_bwd_kernel(SweepArgs a, BwdOut o) {
  extern __shared__ float4 lds4[];
  // LDS: colour rows float4[2*(W+8)] | sdisp[N] | kshift[N] | red[N] | rec[nseg][N][2] | irr[] | scratch[nwaves][2][260]
  const RowSel row = two_row_form(make_row_sel(block_row(bwd_rowid(a.B, a.H), a.H), a.H), a.row_eps);
  if (NROWS != 0 && (row.nrows == 2) != (NROWS == 2)) return;   // workgroup-uniform: this row belongs to the other kernel
  float* sdisp = reinterpret_cast<float*>(lds4 + 2 * kQRowCells(a.W));
  int* kshift = reinterpret_cast<int*>(sdisp + a.N);
  float* red = sdisp + 2 * a.N;
  const int nseg = (a.W + seg_px(Q) - 1) / seg_px(Q);
  const int nsn = nseg * a.N;
  QBoundary bnd;
  bnd.rec = red + a.N;
  bnd.irr = reinterpret_cast<unsigned*>(bnd.rec + 2 * nsn);
  bnd.scratch = reinterpret_cast<float*>(bnd.irr + (nsn + 31) / 32);
  bnd.side = o.side + ((long)wg_image(a.B, a.H) * a.H + bwd_rowid(a.B, a.H)) * (4L * nsn);
  if (NROWS == 0) {
    if (row.nrows == 2) rowquad_bwd_body<MIX, 2, Q>(a, o, row, sdisp, kshift, red, bnd, lds4);
    else                rowquad_bwd_body<MIX, 1, Q>(a, o, row, sdisp, kshift, red, bnd, lds4);
  } else {
    rowquad_bwd_body<MIX, (NROWS == 0 ? 1 : NROWS), Q>(a, o, row, sdisp, kshift, red, bnd, lds4);
  }
}

__global__ void quad_reduce_rows_kernel(const float* __restrict__ partials, float* __restrict__ out, int R, int M) {
  const int j = blockIdx.x, b = blockIdx.y;
  const float* p = partials + (long)b * R * M + j;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < R; i += kWave) acc += p[(long)i * M];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * M + j] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int quad_waves(int W, int Q) {
  const int nseg = ceil_div(W, seg_px(Q));
  if (const char* e = getenv("PD_QUAD_WAVES")) {
    const int w = atoi(e);
    if (w >= 1 && w <= kQMaxWaves) return w < nseg ? w : nseg;
  }
  // Small workgroups: measured at 8x49x192x640, Q = 2 (5 segments per row): 1 / 2 / 3 / 5 waves per workgroup run the
  // forward in 0.148 / 0.112 / 0.133 / 0.179 ms — a 5-wave workgroup leaves room for ONE per CU at 154 VGPRs, 2-wave
  // workgroups pack six, and each wave then walks 2-3 segments.
  return nseg < 2 ? nseg : 2;
}
static size_t quad_fwd_lds(const pd_sweep_desc* d) {
  return (size_t)kQRowCells(d->W) * 2 * sizeof(float4) + ((size_t)d->N + kQMaxWaves) * sizeof(float);
}
static size_t quad_bwd_lds(const pd_sweep_desc* d, int Q) {
  const size_t nsn = (size_t)ceil_div(d->W, seg_px(Q)) * d->N;
  return (size_t)kQRowCells(d->W) * 2 * sizeof(float4) +
         ((size_t)3 * d->N + nsn * 2 + (nsn + 31) / 32 + (size_t)quad_waves(d->W, Q) * 2 * scratch_stride(Q)) * sizeof(float);
}

bool rowquad_applicable(const pd_sweep_desc* d, bool dense_mask) {
  // (no compositing code in these kernels: render_probability sweeps keep the row-shift kernels — ADVICE r2)
  return rowshift_applicable(d) && !dense_mask && !(d->flags & PD_RENDER_PROB) && d->W >= 8 && d->impl != PD_IMPL_ROWS1 && !getenv("PD_NO_ROWQUAD") &&
         quad_bwd_lds(d, PD_QBWD_Q) <= 160 * 1024 && quad_fwd_lds(d) <= 160 * 1024;
}

size_t rowquad_bwd_workspace_floats(const pd_sweep_desc* d) {
  return (size_t)d->B * d->H * d->N * (1 + 4 * (size_t)ceil_div(d->W, seg_px(PD_QBWD_Q)));
}

template <typename K>
static void q_allow_lds(K kernel, size_t shmem) {
  if (shmem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}

int rowquad_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash, hipStream_t stream) {
  constexpr int Q = PD_QFWD_Q;
  dim3 grid(d->H, d->B), block(quad_waves(d->W, Q) * kWave);
  const size_t shmem = quad_fwd_lds(d);
  const bool mix = (d->flags & PD_MIXTURE) != 0, am = (d->flags & PD_AUTOMASK) != 0;
#define PD_QF1(M, A, R)                                                                       \
  do {                                                                                        \
    q_allow_lds(rowquad_fwd_kernel<M, A, R, Q>, shmem);                                       \
    rowquad_fwd_kernel<M, A, R, Q><<<grid, block, shmem, stream>>>(a, rgb_rec, ph_map, stash);\
  } while (0)
#define PD_QF(M, A)                                                         \
  do {                                                                      \
    if (PD_QSPLIT) { PD_QF1(M, A, 2); PD_QF1(M, A, 1); } else PD_QF1(M, A, 0); \
  } while (0)
  if (mix) { if (am) PD_QF(true, true); else PD_QF(true, false); }
  else PD_QF(false, false);
#undef PD_QF
#undef PD_QF1
  return check_launch("rowquad_fwd_kernel");
}

int rowquad_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o_in, hipStream_t stream) {
  constexpr int Q = PD_QBWD_Q;
  dim3 grid(d->H, d->B), block(quad_waves(d->W, Q) * kWave);
  const size_t shmem = quad_bwd_lds(d, Q);
  BwdOut o = o_in;
  o.side = o_in.partials + (size_t)d->B * d->H * d->N;   // workspace: [B][H][N] partial sums | [B][H][nseg*N][4] spill
#define PD_QB(M, R)                                                    \
  do {                                                                  \
    q_allow_lds(rowquad_bwd_kernel<M, R, Q>, shmem);                    \
    rowquad_bwd_kernel<M, R, Q><<<grid, block, shmem, stream>>>(a, o);  \
  } while (0)
  if (d->flags & PD_MIXTURE) {
    if (PD_QSPLIT) { PD_QB(true, 2); PD_QB(true, 1); } else PD_QB(true, 0);
  } else {
    if (PD_QSPLIT) { PD_QB(false, 2); PD_QB(false, 1); } else PD_QB(false, 0);
  }
#undef PD_QB
  int rc = check_launch("rowquad_bwd_kernel");
  if (rc || !o.g_plane || (d->flags & PD_DISP_ROWS)) return rc;
  quad_reduce_rows_kernel<<<dim3(d->N, d->B), kWave, 0, stream>>>(o.partials, o.g_plane, d->H, d->N);
  return check_launch("quad_reduce_rows_kernel");
}

}  // namespace pd

// Test / tuning hook (not part of the public header): resident workgroups per CU the runtime reports for the row-quad
// kernels at this shape (mixture, no automask).  out[0] = forward, out[1] = backward, out[2..3] = their block sizes.
extern "C" int pd_debug_rowquad_occupancy(int W, int N, int* out) {
  pd_sweep_desc d;
  d.B = 1; d.N = N; d.H = 2; d.W = W; d.mode = PD_WARP_DISP; d.flags = PD_MIXTURE; d.sign = 1.0f; d.impl = PD_IMPL_AUTO;
  int nf = -1, nb = -1;
  const int bf = pd::quad_waves(W, PD_QFWD_Q) * pd::kWave, bb = pd::quad_waves(W, PD_QBWD_Q) * pd::kWave;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, (const void*)pd::rowquad_fwd_kernel<true, false, 0, PD_QFWD_Q>, bf,
                                                     pd::quad_fwd_lds(&d));
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pd::rowquad_bwd_kernel<true, 0, PD_QBWD_Q>, bb,
                                                     pd::quad_bwd_lds(&d, PD_QBWD_Q));
  out[0] = nf; out[1] = nb; out[2] = bf; out[3] = bb;
  return 0;
}

#endif  // PD_EXPERIMENTS
