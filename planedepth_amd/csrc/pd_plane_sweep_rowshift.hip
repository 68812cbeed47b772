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
#include <stdlib.h>

#include "pd_rowshift_common.h"

namespace pd {

// ---------------------------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------------------------
// LDS layout shared by both kernels: float4 colour[2*(W+4)] | float sdisp[N] | ...
// One group of U planes in flight: sampling positions + raw taps (+ padding-mask values).
template <int NROWS, int U>
struct PlaneGroup {
  ColTap ct[U];
  Taps<NROWS> tl[U], ts[U];
#if PD_TC_IN_GROUP
  ColourTaps<NROWS> tc[U];  // LDS colour taps ride along with the global loads (their latency overlaps too)
#endif
  float mval[U];
  float dist[U];  // PD_RENDER_PROB: the decoder's inter-plane distance at the TARGET pixel (trainer.py:587)
};

// Issue every global load of planes n0 .. n0+U-1 (no use of the results here: the caller overlaps the latency with
// the arithmetic of the previous group — one-group-ahead software prefetch).
template <bool MIX, bool HASMASK, int NROWS, int U, bool RENDER = false>
__device__ __forceinline__ void group_issue(PlaneGroup<NROWS, U>& g, const SweepArgs& a, const RowSel& row,
                                            const char* __restrict__ lrgb, const float* __restrict__ sdisp, int b,
                                            int y, int n0, int x, int HW, float Wm1, float rcpWm1) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = n0 + u;
    const float* pl = plane_ptr(a.logits + (long)b * a.N * HW, n, HW);  // workgroup-uniform
    if (kAblate & 8) {  // diagnostics: no coordinate chain (integer shift, constant weights)
      g.ct[u].x0 = x + (int)sdisp[n]; g.ct[u].w0 = 0.25f; g.ct[u].w1 = 0.75f;
    } else {
      g.ct[u] = make_col_tap((float)x + sdisp[n], Wm1, rcpWm1);
    }
#if PD_TC_IN_GROUP
    if (!(kAblate & 2)) g.tc[u] = load_colour_taps<NROWS>(lrgb, a.W, colour_off(g.ct[u].x0, a.W));
#endif
    if (RENDER)  // unshifted, coalesced: read where the pixel is, not where it samples
      g.dist[u] = (n < a.N - 1 && x < a.W) ? a.dists[((long)b * (a.N - 1) + n) * HW + (long)y * a.W + x] : 0.0f;
    g.mval[u] = 1.0f;
    if (HASMASK && !(kAblate & 16))
      g.mval[u] = buf_load(row_rsrc_uniform(plane_ptr(a.padding_mask + (long)b * a.N * HW + (long)y * a.W, n, HW), a.W), (unsigned)x << 2);
    if (kAblate & 1) {  // diagnostics: no logit / sigma loads
      g.tl[u].a0 = g.tl[u].a1 = g.tl[u].b0 = g.tl[u].b1 = g.ct[u].w0;
      g.ts[u] = g.tl[u];
    } else {
      const TapPos tp = tap_pos(g.ct[u]);
      g.tl[u] = load_taps<NROWS>(row_rsrc(pl + (long)row.yA * a.W, a.W), row_rsrc(pl + (long)row.yB * a.W, a.W), tp);
      if (MIX) {
        const float* ps = plane_ptr(a.sigma + (long)b * a.N * HW, n, HW);
        g.ts[u] = load_taps<NROWS>(row_rsrc(ps + (long)row.yA * a.W, a.W), row_rsrc(ps + (long)row.yB * a.W, a.W), tp);
      }
    }
  }
}

template <bool MIX, bool HASMASK, int NROWS, int U, bool RENDER = false>
__device__ __forceinline__ void fwd_compute(const PlaneGroup<NROWS, U>& g, const SweepArgs& a, const RowSel& row,
                                            const char* __restrict__ lrgb, int b, int n0, int pix, int HW, float t0,
                                            float t1, float t2, float ea, bool automask, FwdAcc& acc, uint32_t& bits,
                                            float* __restrict__ stash, RenderState* rs = nullptr) {
#if PD_TC_IN_GROUP
  const ColourTaps<NROWS>* tc = g.tc;
#else
  ColourTaps<NROWS> tc[U];  // all LDS reads of the group first, then the arithmetic
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (!(kAblate & 2)) tc[u] = load_colour_taps<NROWS>(lrgb, a.W, colour_off(g.ct[u].x0, a.W));
#endif
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = n0 + u;
    float live = 1.0f;
    if (HASMASK) {  // rec_features * padding_mask (trainer.py:580): a masked plane samples as all-zero features,
      const bool mk = g.mval[u] != 0.0f;  // i.e. every tap weight of the plane is zero
      if (mk) bits |= 1u << (n & 31);
      if ((n & 31) == 31 || n == a.N - 1) {
        stash[((long)b * a.stash_k + kStashBase + (n >> 5)) * HW + pix] = __uint_as_float(bits);
        bits = 0;
      }
      live = mk ? 1.0f : 0.0f;
    }
    const TapW w = tap_weights<NROWS>(g.ct[u], row, live);
    // x0 = -1: the load was issued at column 0, so its FIRST dword is the right tap; moving the weights instead of the
    // values costs two selects per row shared by logits and sigma (the colour taps come from LDS and are in place)
    const bool edge = (g.ct[u].x0 == -1);
    TapW we;
    we.a0 = edge ? w.a1 : w.a0;
    we.a1 = edge ? 0.0f : w.a1;
    we.b0 = (NROWS == 2) ? (edge ? w.b1 : w.b0) : 0.0f;
    we.b1 = (NROWS == 2) ? (edge ? 0.0f : w.b1) : 0.0f;
    const float l = tap_value<NROWS>(g.tl[u], we);
    const float s = MIX ? tap_value<NROWS>(g.ts[u], we) : 0.0f;
    float c0, c1, c2;
    if (kAblate & 2) { c0 = w.a0; c1 = w.a1; c2 = l; }  // diagnostics: no colour taps
    else colour_values<NROWS>(tc[u], w, c0, c1, c2);
    if (kAblate & 4) { acc.Z += l; acc.S += s; acc.C0 += c0; acc.C1 += c1; acc.C2 += c2; acc.m = 0.0f; }  // no softmax/mixture math
    else if (RENDER)  // alpha compositing front to back (trainer.py:584-591): the planes arrive in order
      mixture_accumulate<MIX>(acc, render_prob(*rs, render_alpha(l, g.dist[u], n == a.N - 1)), s, c0, c1, c2, t0, t1, t2,
                              ea, automask);
    else fwd_accumulate<MIX>(acc, l, s, c0, c1, c2, t0, t1, t2, ea, automask);
  }
}

// Work split inside a row workgroup.  Whole 64-pixel segments are dealt to the waves round-robin (wave w takes
// segments w, w+nwaves, ...: the waves of a workgroup then stream through ADJACENT parts of every plane row at the
// same time, which the memory system rewards — giving each wave a contiguous slice of the row instead measured 20%
// slower).  The r = nseg % nwaves segments left after the full rounds (W = 640: 10 segments over 4 waves leave 2) are
// not given to r of the waves (3,3,2,2 segments measured 10% slower per pixel than W = 512 or 768) but cut along the
// PLANE axis: their r*cps chunks of G planes are sliced evenly over all the waves.  A slice is shorter than a
// segment, so it touches at most two of the left-over segments.
struct RowWork {
  int full;          // rounds of whole segments
  int r;             // left-over segments
  int cps;           // plane chunks per segment
  int cb, ce;        // this wave's chunk slice of the left-over segments
};
__device__ __forceinline__ void slice_of(int r, int cps, int w, int nwaves, int& cb, int& ce) {
  cb = r * cps * w / nwaves;
  ce = r * cps * (w + 1) / nwaves;
}
__device__ __forceinline__ RowWork row_work(int nseg, int N, int G, int wave, int nwaves) {
  RowWork k;
  k.full = nseg / nwaves;
  k.r = nseg - k.full * nwaves;
  k.cps = (N + G - 1) / G;
  slice_of(k.r, k.cps, wave, nwaves, k.cb, k.ce);
  k.cb = __builtin_amdgcn_readfirstlane(k.cb);
  k.ce = __builtin_amdgcn_readfirstlane(k.ce);
  return k;
}
// Work item `it` of this wave: segment and plane range; piece = 0/1 for a slice piece of a left-over segment, else -1.
__device__ __forceinline__ bool work_item(const RowWork& k, int it, int wave, int nwaves, int N, int G, int& seg,
                                          int& n_lo, int& n_hi, int& piece) {
  if (it < k.full) {
    seg = it * nwaves + wave; n_lo = 0; n_hi = N; piece = -1;
    return true;
  }
  piece = it - k.full;
  const int j = k.cb / k.cps + piece;  // left-over segment index
  if (piece > 1 || j * k.cps >= k.ce) return false;
  // everything here derives from the wave index: say so (readfirstlane), or the plane-row descriptors built from n_lo
  // are treated as divergent and every buffer load gets a waterfall loop
  seg = __builtin_amdgcn_readfirstlane(k.full * nwaves + j);
  n_lo = __builtin_amdgcn_readfirstlane(max(k.cb - j * k.cps, 0) * G);
  n_hi = __builtin_amdgcn_readfirstlane(min((min(k.ce, (j + 1) * k.cps) - j * k.cps) * G, N));
  return true;
}

// Merge the partial sums of two plane ranges of the same pixel (split online softmax: common reference = the larger).
__device__ __forceinline__ FwdAcc merge_acc(const FwdAcc& a, const FwdAcc& b) {
  FwdAcc r;
  r.m = fmaxf(a.m, b.m);
  const float sa = exp2_fast(a.m - r.m), sb = exp2_fast(b.m - r.m);
  r.Z = a.Z * sa + b.Z * sb;
  r.S = a.S * sa + b.S * sb;
  r.C0 = a.C0 * sa + b.C0 * sb;
  r.C1 = a.C1 * sa + b.C1 * sb;
  r.C2 = a.C2 * sa + b.C2 * sb;
  r.Mx = a.Mx * sa + b.Mx * sb;
  r.Ma = a.Ma * sa + b.Ma * sb;
  return r;
}
__device__ __forceinline__ void park_acc(float* __restrict__ slot, int lane, const FwdAcc& a) {  // slot: [8][64]
  slot[0 * kWave + lane] = a.m;  slot[1 * kWave + lane] = a.Z;  slot[2 * kWave + lane] = a.S;
  slot[3 * kWave + lane] = a.C0; slot[4 * kWave + lane] = a.C1; slot[5 * kWave + lane] = a.C2;
  slot[6 * kWave + lane] = a.Mx; slot[7 * kWave + lane] = a.Ma;
}
__device__ __forceinline__ FwdAcc fetch_acc(const float* __restrict__ slot, int lane) {
  FwdAcc a;
  a.m = slot[0 * kWave + lane];  a.Z = slot[1 * kWave + lane];  a.S = slot[2 * kWave + lane];
  a.C0 = slot[3 * kWave + lane]; a.C1 = slot[4 * kWave + lane]; a.C2 = slot[5 * kWave + lane];
  a.Mx = slot[6 * kWave + lane]; a.Ma = slot[7 * kWave + lane];
  return a;
}

template <bool MIX>
__device__ __forceinline__ float fwd_store(const SweepArgs& a, const FwdAcc& acc, int b, int pix, int HW, float t0,
                                          float t1, float t2, float ea, bool automask, float* __restrict__ rgb_rec,
                                          float* __restrict__ ph_map, float* __restrict__ stash, bool normalise = true) {
  const FwdResult r = fwd_finish<MIX>(acc, t0, t1, t2, ea, automask, normalise);
  float* st = stash + (long)b * a.stash_k * HW + pix;
  st[0] = r.lse2;
  st[HW] = r.Sn;
  st[2 * HW] = r.mx;
  st[3 * HW] = r.sel;
  rgb_rec[((long)b * 3 + 0) * HW + pix] = r.r0;
  rgb_rec[((long)b * 3 + 1) * HW + pix] = r.r1;
  rgb_rec[((long)b * 3 + 2) * HW + pix] = r.r2;
  ph_map[(long)b * HW + pix] = r.ph;
  return r.ph;
}

template <bool MIX, bool HASMASK, bool AUTO, int NROWS, bool RENDER = false>
__device__ __forceinline__ float rowshift_fwd_body(const SweepArgs& a, const RowSel& row, float4* lrgb, float* sdisp,
                                                  float* parts, float* __restrict__ rgb_rec,
                                                  float* __restrict__ ph_map, float* __restrict__ stash) {
  // (compositing keeps a distance per plane of the group and its running state alive: half the group size, or the
  // register allocator spills 40-70 VGPRs at the 168 this kernel may use)
  constexpr int UB = (RENDER && PD_FWD_U > 1) ? PD_FWD_U / 2 : PD_FWD_U;
  constexpr int U = (NROWS == 1) ? UB : (UB > 1 ? UB / 2 : 1);
  constexpr int G = HASMASK ? 32 : U;  // chunk of the work split (mask words of the stash are written whole)
  static_assert(32 % U == 0, "plane groups must tile the 32-plane mask words");
  const int y = block_row(wg_rowid(a.B, a.H), a.H), b = wg_image(a.B, a.H);
  const int HW = a.H * a.W, N = a.N;
  // mixture kernels are specialised on the automask flag (it costs an exponential per plane); L1 reads it at run time
  const bool automask = MIX ? AUTO : (bool)(a.flags & PD_AUTOMASK);
  const float Wm1 = (float)(a.W - 1), rcpWm1 = refined_rcp(Wm1);
  const float* srcb = a.src + (long)b * 3 * HW;
  stage_row_constants<NROWS>(a, b, row, lrgb, sdisp, y);
  __syncthreads();
  const char* lbytes = reinterpret_cast<const char*>(lrgb);
  const int lane = threadIdx.x & (kWave - 1), nwaves = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (a.W + kWave - 1) / kWave;
  RowWork rw = row_work(nseg, N, G, wave, nwaves);
  if (RENDER) {   // compositing walks the planes of a pixel in order: whole segments only (a ragged last round)
    rw.full = (nseg + nwaves - 1) / nwaves;
    rw.r = 0;
  }
  float ph_sum = 0.0f;  // this lane's share of sum(ph_map) (returned: the kernel adds the wave totals to a.ph_mean)
  auto target_pixel = [&](int pix, float& t0, float& t1, float& t2, float& ea) {
    t0 = a.tgt[((long)b * 3 + 0) * HW + pix];
    t1 = a.tgt[((long)b * 3 + 1) * HW + pix];
    t2 = a.tgt[((long)b * 3 + 2) * HW + pix];
    ea = 0.0f;  // 3 x identity-reprojection error
    if (automask) ea = fabsf(srcb[pix] - t0) + fabsf(srcb[HW + pix] - t1) + fabsf(srcb[2 * HW + pix] - t2);
  };
  for (int it = 0;; ++it) {
    int seg, n_lo, n_hi, piece;
    if (!work_item(rw, it, wave, nwaves, N, G, seg, n_lo, n_hi, piece)) break;
    const int x = seg * kWave + lane;
    if (x < a.W) {
    const int pix = y * a.W + x;
    float t0, t1, t2, ea;
    target_pixel(pix, t0, t1, t2, ea);
    FwdAcc acc;
    RenderState rs;
    uint32_t bits = 0;
    // Groups of U planes through a software pipeline: while group i is reduced the loads of group i+1 (PD_PF_DEPTH 2;
    // measured best) or of groups i+1 and i+2 (PD_PF_DEPTH 3; no faster, more registers) are in flight.
    PlaneGroup<NROWS, U> g0, g1, g2;
    const int nfull = (n_hi - n_lo) / U;  // full groups
#define PD_FISSUE(GR, I) group_issue<MIX, HASMASK, NROWS, U, RENDER>(GR, a, row, lbytes, sdisp, b, y, n_lo + (I) * U, x, HW, Wm1, rcpWm1)
#define PD_FCOMP(GR, I) fwd_compute<MIX, HASMASK, NROWS, U, RENDER>(GR, a, row, lbytes, b, n_lo + (I) * U, pix, HW, t0, t1, t2, ea, automask, acc, bits, stash, &rs)
    int gi = 0;
    if (PD_FWD_PF && PD_PF_DEPTH == 2) {
      if (nfull > 0) PD_FISSUE(g0, 0);
      for (; gi + 2 <= nfull; gi += 2) {
        PD_FISSUE(g1, gi + 1);
        PD_FCOMP(g0, gi);
        // Unconditional on purpose: under an `if` the waitcnt pass has to assume the loads were NOT issued, counts too
        // few operations in flight and makes every second group wait for the loads issued right before it (found in
        // the ISA: vmcnt(7)..(0) instead of (15)..(8)).  On the last round this re-loads the final group; nobody reads it.
        PD_FISSUE(g0, min(gi + 2, nfull - 1));
        PD_FCOMP(g1, gi + 1);
      }
      if (gi < nfull) PD_FCOMP(g0, gi);
    } else if (PD_FWD_PF) {
      if (nfull > 0) PD_FISSUE(g0, 0);
      if (nfull > 1) PD_FISSUE(g1, 1);
      for (; gi + 3 <= nfull; gi += 3) {
        PD_FISSUE(g2, gi + 2);
        PD_FCOMP(g0, gi);
        PD_FISSUE(g0, min(gi + 3, nfull - 1));
        PD_FCOMP(g1, gi + 1);
        PD_FISSUE(g1, min(gi + 4, nfull - 1));
        PD_FCOMP(g2, gi + 2);
      }
      if (gi < nfull) PD_FCOMP(g0, gi);
      if (gi + 1 < nfull) PD_FCOMP(g1, gi + 1);
    } else {
      for (; gi < nfull; ++gi) { PD_FISSUE(g0, gi); PD_FCOMP(g0, gi); }
    }
#undef PD_FISSUE
#undef PD_FCOMP
    for (int n = n_lo + nfull * U; n < n_hi; ++n) {  // remainder planes (only at the end of the plane axis)
      PlaneGroup<NROWS, 1> gr;
      group_issue<MIX, HASMASK, NROWS, 1, RENDER>(gr, a, row, lbytes, sdisp, b, y, n, x, HW, Wm1, rcpWm1);
      fwd_compute<MIX, HASMASK, NROWS, 1, RENDER>(gr, a, row, lbytes, b, n, pix, HW, t0, t1, t2, ea, automask, acc, bits, stash, &rs);
    }
    if (piece < 0) ph_sum += fwd_store<MIX>(a, acc, b, pix, HW, t0, t1, t2, ea, automask, rgb_rec, ph_map, stash, !RENDER);
    else park_acc(parts + ((wave * 2 + piece) * 8) * kWave, lane, acc);
    }
  }
  if (rw.r == 0) return ph_sum;  // workgroup-uniform
  __syncthreads();
  if (wave < rw.r) {  // wave j merges the pieces of left-over segment j (in plane order) and finishes its pixels
    const int j = wave, x = (rw.full * nwaves + j) * kWave + lane;
    if (x < a.W) {
      FwdAcc acc;
      acc.m = -3.0e38f;  // finite: merging the empty sum must not produce inf - inf
      for (int w2 = 0; w2 < nwaves; ++w2) {
        int cb2, ce2;
        slice_of(rw.r, rw.cps, w2, nwaves, cb2, ce2);
        if (cb2 < (j + 1) * rw.cps && ce2 > j * rw.cps && ce2 > cb2)
          acc = merge_acc(acc, fetch_acc(parts + ((w2 * 2 + (cb2 < j * rw.cps ? 1 : 0)) * 8) * kWave, lane));
      }
      const int pix = y * a.W + x;
      float t0, t1, t2, ea;
      target_pixel(pix, t0, t1, t2, ea);
      ph_sum += fwd_store<MIX>(a, acc, b, pix, HW, t0, t1, t2, ea, automask, rgb_rec, ph_map, stash);
    }
  }
  return ph_sum;
}

// ---------------------------------------------------------------------------------------------------------------
// Row pairs for the rows whose vertical round trip is inexact.
// ---------------------------------------------------------------------------------------------------------------
// Such a row y samples (1 - eps) * row y + eps * row p with p = y +- 1 ("leans" on p).  Served alone it loads two
// source rows for one target row (a quarter of the rows at H = 192: +25% HBM reads in both kernels, which are bound by
// exactly that).  When p itself is exact, or leans back on y, one workgroup computes BOTH target rows from the two
// source rows it loads anyway (one thread = the same column of both rows; the sampling column does not depend on the row
// when disparities are per plane), and the workgroup of p retires at once.  The rule is local (rows y-1 .. y+1), so every
// workgroup decides its own role without a table:
//   * y leans on p, p leans back on y  -> the lower of the two leads;
//   * y leans on p, p exact            -> y leads unless p-1 also leans on p and y = p+1 (the upper neighbour wins);
//   * y leans on p, p leans elsewhere  -> y stays a single two-source-row row (a chain; rare).
// At H = 192: 48 inexact rows -> 26 pairs, 10 left alone (re-reads 25% -> 5% of the rows); H = 384: 94 -> 60 + 14.
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

struct PairPx { float t0, t1, t2, ea; FwdAcc acc; };

template <bool MIX, int U>
__device__ __forceinline__ void pair_accumulate(const PlaneGroup<2, U>& g, const PairW& pw, PairPx& pL, PairPx& pP,
                                                bool automask) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float w0 = g.ct[u].w0, w1 = g.ct[u].w1;
    const bool edge = (g.ct[u].x0 == -1);  // the loads were issued at column 0: their first dword is the RIGHT tap
    const float e0 = edge ? w1 : w0, e1 = edge ? 0.0f : w1;
    const Taps<2>& tl = g.tl[u];
    const Taps<2>& ts = g.ts[u];
    const ColourTaps<2>& tc = g.tc[u];
    const float hlA = tl.a0 * e0 + tl.a1 * e1, hlB = tl.b0 * e0 + tl.b1 * e1;
    const float hsA = MIX ? ts.a0 * e0 + ts.a1 * e1 : 0.0f, hsB = MIX ? ts.b0 * e0 + ts.b1 * e1 : 0.0f;
    const float rA = tc.nw.x * w0 + tc.ne.x * w1, gA = tc.nw.y * w0 + tc.ne.y * w1, bA = tc.nw.z * w0 + tc.ne.z * w1;
    const float rB = tc.sw.x * w0 + tc.se.x * w1, gB = tc.sw.y * w0 + tc.se.y * w1, bB = tc.sw.z * w0 + tc.se.z * w1;
    fwd_accumulate<MIX>(pL.acc, pw.a0 * hlA + pw.b0 * hlB, pw.a0 * hsA + pw.b0 * hsB, pw.a0 * rA + pw.b0 * rB,
                        pw.a0 * gA + pw.b0 * gB, pw.a0 * bA + pw.b0 * bB, pL.t0, pL.t1, pL.t2, pL.ea, automask);
    fwd_accumulate<MIX>(pP.acc, pw.a1 * hlB + pw.b1 * hlA, pw.a1 * hsB + pw.b1 * hsA, pw.a1 * rB + pw.b1 * rA,
                        pw.a1 * gB + pw.b1 * gA, pw.a1 * bB + pw.b1 * bA, pP.t0, pP.t1, pP.t2, pP.ea, automask);
  }
}

template <bool MIX, bool AUTO>
__device__ __forceinline__ float rowpair_fwd_body(const SweepArgs& a, int yL, int yP, int b, float4* lrgb, float* sdisp,
                                                  float* parts, float* __restrict__ rgb_rec,
                                                  float* __restrict__ ph_map, float* __restrict__ stash) {
  static_assert(PD_TC_IN_GROUP, "the pair bodies use the two-group pipeline with colour taps in the group");
  constexpr int U = PD_FWD_U > 1 ? PD_FWD_U / 2 : 1;  // planes per group; each carries both rows
  constexpr int G = U;
  const int HW = a.H * a.W, N = a.N;
  const bool automask = MIX ? AUTO : (bool)(a.flags & PD_AUTOMASK);
  const float Wm1 = (float)(a.W - 1), rcpWm1 = refined_rcp(Wm1);
  const float* srcb = a.src + (long)b * 3 * HW;
  const PairW pw = pair_weights(yL, yP, a.H);
  RowSel rows;  // the two-row loaders' "source rows A and B" are the leader's row and the partner's row
  rows.nrows = 2; rows.yA = yL; rows.yB = yP; rows.wA = rows.wB = rows.wy_main = 1.0f;
  stage_row_constants<2>(a, b, rows, lrgb, sdisp, yL);
  __syncthreads();
  const char* lbytes = reinterpret_cast<const char*>(lrgb);
  const int lane = threadIdx.x & (kWave - 1), nwaves = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (a.W + kWave - 1) / kWave;
  const RowWork rw = row_work(nseg, N, G, wave, nwaves);
  float ph_sum = 0.0f;
  auto target_pixel = [&](int pix, PairPx& p) {
    p.t0 = a.tgt[((long)b * 3 + 0) * HW + pix];
    p.t1 = a.tgt[((long)b * 3 + 1) * HW + pix];
    p.t2 = a.tgt[((long)b * 3 + 2) * HW + pix];
    p.ea = 0.0f;
    if (automask) p.ea = fabsf(srcb[pix] - p.t0) + fabsf(srcb[HW + pix] - p.t1) + fabsf(srcb[2 * HW + pix] - p.t2);
  };
  // parked partial sums: the single-row layout [nwaves][2 pieces][8][64] holds row L; row P follows in a second copy
  auto slot = [&](int w, int piece, int r) { return parts + (((r * nwaves + w) * 2 + piece) * 8) * kWave; };
  for (int it = 0;; ++it) {
    int seg, n_lo, n_hi, piece;
    if (!work_item(rw, it, wave, nwaves, N, G, seg, n_lo, n_hi, piece)) break;
    const int x = seg * kWave + lane;
    if (x < a.W) {
      const int pixL = yL * a.W + x, pixP = yP * a.W + x;
      PairPx pL, pP;
      target_pixel(pixL, pL);
      target_pixel(pixP, pP);
      PlaneGroup<2, U> g0, g1;
      const int nfull = (n_hi - n_lo) / U;
#define PD_PISSUE(GR, I) group_issue<MIX, false, 2, U>(GR, a, rows, lbytes, sdisp, b, yL, n_lo + (I) * U, x, HW, Wm1, rcpWm1)
      int gi = 0;
      if (nfull > 0) PD_PISSUE(g0, 0);
      for (; gi + 2 <= nfull; gi += 2) {
        PD_PISSUE(g1, gi + 1);
        pair_accumulate<MIX, U>(g0, pw, pL, pP, automask);
        PD_PISSUE(g0, min(gi + 2, nfull - 1));  // unconditional: see the single-row forward
        pair_accumulate<MIX, U>(g1, pw, pL, pP, automask);
      }
      if (gi < nfull) pair_accumulate<MIX, U>(g0, pw, pL, pP, automask);
#undef PD_PISSUE
      for (int n = n_lo + nfull * U; n < n_hi; ++n) {  // remainder planes (only at the end of the plane axis)
        PlaneGroup<2, 1> gr;
        group_issue<MIX, false, 2, 1>(gr, a, rows, lbytes, sdisp, b, yL, n, x, HW, Wm1, rcpWm1);
        pair_accumulate<MIX, 1>(gr, pw, pL, pP, automask);
      }
      if (piece < 0) {
        ph_sum += fwd_store<MIX>(a, pL.acc, b, pixL, HW, pL.t0, pL.t1, pL.t2, pL.ea, automask, rgb_rec, ph_map, stash);
        ph_sum += fwd_store<MIX>(a, pP.acc, b, pixP, HW, pP.t0, pP.t1, pP.t2, pP.ea, automask, rgb_rec, ph_map, stash);
      } else {
        park_acc(slot(wave, piece, 0), lane, pL.acc);
        park_acc(slot(wave, piece, 1), lane, pP.acc);
      }
    }
  }
  if (rw.r == 0) return ph_sum;  // workgroup-uniform
  __syncthreads();
  if (wave < rw.r) {  // wave j merges the pieces of left-over segment j (in plane order) for both rows
    const int j = wave, x = (rw.full * nwaves + j) * kWave + lane;
    if (x < a.W) {
      for (int r = 0; r < 2; ++r) {
        FwdAcc acc;
        acc.m = -3.0e38f;  // finite: merging the empty sum must not produce inf - inf
        for (int w2 = 0; w2 < nwaves; ++w2) {
          int cb2, ce2;
          slice_of(rw.r, rw.cps, w2, nwaves, cb2, ce2);
          if (cb2 < (j + 1) * rw.cps && ce2 > j * rw.cps && ce2 > cb2)
            acc = merge_acc(acc, fetch_acc(slot(w2, cb2 < j * rw.cps ? 1 : 0, r), lane));
        }
        const int pix = (r ? yP : yL) * a.W + x;
        PairPx p;
        target_pixel(pix, p);
        ph_sum += fwd_store<MIX>(a, acc, b, pix, HW, p.t0, p.t1, p.t2, p.ea, automask, rgb_rec, ph_map, stash);
      }
    }
  }
  return ph_sum;
}

template <bool MIX, bool HASMASK, bool AUTO, bool RENDER = false>
__global__ __launch_bounds__(kRowThreadsMax, HASMASK ? 3 : PD_FWD_OCC) void rowshift_fwd_kernel(SweepArgs a, float* __restrict__ rgb_rec,
                                                                     float* __restrict__ ph_map,
                                                                     float* __restrict__ stash) {
  extern __shared__ float4 lds4[];
  // LDS: colour rows float4[2*(W+4)] | sdisp[N] | parked partial sums [nwaves][2][8][64]
  float* sdisp = reinterpret_cast<float*>(lds4 + 2 * (a.W + 4));
  float* parts = sdisp + a.N;
  const int y = block_row(wg_rowid(a.B, a.H), a.H);
  const RowSel row = two_row_form(make_row_sel(y, a.H), a.fast_rows != 0);
  float ph_sum = 0.0f;
  int partner = y;
  // row pairs: per-plane scalar disparities and no per-pixel mask (then the sampling column is shared by the rows)
  const PairRole role = (a.pairs && !HASMASK && !RENDER) ? pair_role(y, a.H, partner) : kSingle;
  if (role == kAbsorbed) {
    // this row is computed by its neighbour's workgroup
  } else if (role == kLeader) {
    if (!HASMASK && !RENDER)
      ph_sum = rowpair_fwd_body<MIX, AUTO>(a, y, partner, wg_image(a.B, a.H), lds4, sdisp, parts, rgb_rec, ph_map, stash);
  } else if (row.nrows == 2) {
    ph_sum = rowshift_fwd_body<MIX, HASMASK, AUTO, 2, RENDER>(a, row, lds4, sdisp, parts, rgb_rec, ph_map, stash);
  } else {
    ph_sum = rowshift_fwd_body<MIX, HASMASK, AUTO, 1, RENDER>(a, row, lds4, sdisp, parts, rgb_rec, ph_map, stash);
  }
  if (a.ph_mean) {  // fused `.mean()` of trainer.py:742: wave totals -> LDS -> ONE atomic per workgroup (one per wave
    // measured +13 us on the forward: 6144 atomics on a single address serialise in L2)
    const float v = wave_sum_hi(ph_sum);
    __syncthreads();  // everybody is done with `parts`
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) parts[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += parts[w];
      unsafeAtomicAdd(a.ph_mean, t * a.inv_numel);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward: one workgroup per target row; 64-lane segments of the row; gather-form adjoint.
// ---------------------------------------------------------------------------------------------------------------
// Segment-boundary records.  A 64-lane segment's contributions reach three slots outside it: T0-1 (only when the first
// lane's left tap sits one column left of its slot), Tlast+1 (always: the last lane's right tap) and Tlast+2 (only when
// the last lane's taps sit one column right).  The common record lives in LDS, bnd1[(seg*N + n)*2 + tensor], zero-filled
// at kernel start; the two rare ones (fp32 coordinate noise, `regular` false) go to a per-workgroup global spill,
// side[((seg*N + n)*2 + tensor)*2 + {0: T0-1, 1: Tlast+2}], flagged in the LDS bitmap irr[] so that nothing has to be
// zero-filled there.  (All six records in LDS cost 23 KB at 384x1280x49 and one resident workgroup per CU.)
//
// Route lane contributions (c0 -> slot lane+dl, c1 -> slot lane+dl+1) to their slots inside the wave and return this
// lane's slot total.  `last` = last active lane of the segment.
struct Boundary {   // where a workgroup parks what leaves its segments (see route())
  float* rec;       // LDS  [nseg*N][2 tensors]
  unsigned* irr;    // LDS  bitmap over (seg, plane): the two rare records were written to `side`
  float* side;      // HBM  [nseg*N][2 tensors][2], this workgroup's slice
};

__device__ __forceinline__ float route(float c0, float c1, int dl, bool regular, int lane, int last,
                                       const Boundary& bnd, int sn, int tns) {
  float* b1 = bnd.rec + sn * 2 + tns;
  if (regular) {  // every lane's left tap is exactly its own slot: one wave-wide shift
    if (lane == last) *b1 = c1;  // leaves the segment on the right
    return c0 + wave_shift_up1(c1);
  }
  float* side2 = bnd.side + sn * 4 + tns * 2;
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
    side2[0] = (d_first == -1) ? c0_first : 0.0f;                                   // slot -1
    *b1 = ((d_last == 1) ? c0_last : 0.0f) + ((d_last == 0) ? c1_last : 0.0f) +
          ((last >= 1 && d_prev == 1) ? c1_prev : 0.0f);                            // slot last+1
    side2[1] = (d_last == 1) ? c1_last : 0.0f;                                      // slot last+2
    atomicOr(bnd.irr + (sn >> 5), 1u << (sn & 31));
  }
  return out;
}

struct SegCtx {
  int seg, seg_prev, T0, xt, last, lane, pix;   // seg_prev: the left neighbour on the ring
  bool active;
};

// PD_RENDER_PROB: what the front-to-back compositing carries from plane to plane of one pixel
struct RenderBwd {
  float T = 1.0f;        // transmittance in front of the current plane
  float prefix = 0.0f;   // sum_{k <= n} p_k dL/dp_k
  float Rtot = 0.0f;     // sum over all planes of the same, known in closed form from the pixel's stash
};

template <bool MIX, bool HASMASK, int NROWS, int U, bool RENDER = false>
__device__ __forceinline__ void bwd_compute(const PlaneGroup<NROWS, U>& g, const SweepArgs& a, const BwdOut& o,
                                            const RowSel& row, const char* __restrict__ lrgb,
                                            const int* __restrict__ kshift, float* __restrict__ red,
                                            const Boundary& bnd, int b, int y, int n0, const SegCtx& sc,
                                            const PixelCtx& c, int HW, float gix_scale, int want_plane,
                                            int gl_bytes, int gs_bytes, uint32_t& bits, RenderBwd* rb = nullptr) {
  const int W = a.W, N = a.N;
#if PD_TC_IN_GROUP
  const ColourTaps<NROWS>* tc = g.tc;
#else
  ColourTaps<NROWS> tc[U];  // all LDS reads of the group first, then the arithmetic
#pragma unroll
  for (int u = 0; u < U; ++u) tc[u] = load_colour_taps<NROWS>(lrgb, W, colour_off(g.ct[u].x0, W));
#endif
  float gds[U], outl[U], outs[U];
  unsigned xoffs[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int n = n0 + u;
    const int k = __builtin_amdgcn_readfirstlane(kshift[n]);  // nominal shift floor(s*d), |k| <= W; wave-uniform
    bool mk = sc.active;
    if (HASMASK) {
      if ((n & 31) == 0)
        bits = __float_as_uint((o.stash + ((long)b * a.stash_k + kStashBase + (n >> 5)) * HW)[(unsigned)sc.pix]);
      mk = sc.active && ((bits >> (n & 31)) & 1u);
    }
    const ColTap& t = g.ct[u];
    // forward values (a masked plane contributes nothing to its own gradient, so its samples need no zeroing here)
    const TapW w = tap_weights<NROWS>(t, row, 1.0f);
    float c0, c1, c2, d0x, d1x, d2x;
    colour_values<NROWS>(tc[u], w, c0, c1, c2);
    colour_dx<NROWS>(tc[u], row, d0x, d1x, d2x);
    const bool edge = (t.x0 == -1);
    Taps<NROWS> tl = g.tl[u], ts = g.ts[u];
    fix_edge<NROWS>(tl, edge);
    if (MIX) fix_edge<NROWS>(ts, edge);
    const float l = tap_value<NROWS>(tl, w);
    const float s = MIX ? tap_value<NROWS>(ts, w) : 0.0f;
    PlaneGrad pg;
    if (RENDER) {  // d prob_k / d alpha_n for k >= n through the transmittance (trainer.py:584-591)
      const bool last = (n == N - 1);
      const float lm = mk ? l : 0.0f;   // a masked plane samples as all-zero features: alpha = 0, the state passes through
      const float dist = g.dist[u];
      const float alpha = render_alpha(lm, dist, last);
      const float pn = alpha * rb->T;
      pg = plane_grad_p<MIX>(c, pn, s, c0, c1, c2);
      rb->prefix += pg.g_l * pn;
      const float keep = 1.0f - alpha + 1e-10f;
      const float g_alpha = pg.g_l * rb->T - (rb->Rtot - rb->prefix) / keep;
      const float da = 1.0f - alpha;   // d alpha / d (relu(l) * dist)
      pg.g_l = (!last && lm > 0.0f) ? g_alpha * dist * da : 0.0f;
      if (o.g_dists && !last && sc.active) o.g_dists[((long)b * (N - 1) + n) * HW + sc.pix] = g_alpha * fmaxf(lm, 0.0f) * da;
      rb->T *= keep;
    } else {
      pg = plane_grad<MIX>(c, l, s, c0, c1, c2);
    }
    // adjoint of the horizontal gather: contributions to source x0 (weight w0) and x0+1 (weight w1), if inside
    const bool v0 = (unsigned)t.x0 < (unsigned)W, v1 = (unsigned)(t.x0 + 1) < (unsigned)W;
    const float live = mk ? (NROWS == 1 ? 1.0f : row.wy_main) : 0.0f;   // padding mask x vertical adjoint weight of the own row
    const float m0 = v0 ? t.w0 * live : 0.0f, m1 = v1 ? t.w1 * live : 0.0f;
    const int dl = (mk && (v0 || v1)) ? t.x0 - sc.xt - k : 0;
    const float cl0 = pg.g_l * m0, cl1 = pg.g_l * m1, cs0 = pg.g_s * m0, cs1 = pg.g_s * m1;
    float gd = 0.0f;
    if (want_plane) {
      const float dlx = tap_dx<NROWS>(tl, row);
      const float dsx = MIX ? tap_dx<NROWS>(ts, row) : 0.0f;
      gd = (pg.g_l * dlx + pg.g_s * dsx + pg.gc0 * d0x + pg.gc1 * d1x + pg.gc2 * d2x) * gix_scale;
      gd = mk ? gd : 0.0f;
    }
    const bool regular = __all(dl == 0);
    // the source pixel this slot owns: (xt + k) mod W (ring of W slots); inactive lanes store out of range (dropped)
    const unsigned xs4 = (unsigned)(sc.xt + k) << 2, W4 = (unsigned)W << 2;
    const unsigned xw4 = (xs4 < W4) ? xs4 : xs4 + ((k > 0) ? 0u - W4 : W4);
    const unsigned xoff = sc.active ? xw4 : 0xFFFFFFF0u;
    const int sn = sc.seg * N + n;
    outl[u] = route(cl0, cl1, dl, regular, sc.lane, sc.last, bnd, sn, 0);
    if (MIX) outs[u] = route(cs0, cs1, dl, regular, sc.lane, sc.last, bnd, sn, 1);
    xoffs[u] = xoff;
    gds[u] = gd;
  }
  // Stores of the group, as late as possible: the first slot of the segment (lane 0) also receives what the last lane of
  // the left neighbour segment hands over (route()).  If that wave has already been here — the waves of a workgroup
  // run the same planes at about the same time — lane 0 takes the value out of LDS now (exchange with 0, so that it
  // is added exactly once) and it never becomes a deferred global atomic; otherwise the epilogue adds it as before.
  {
    float hl[U], hs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      hl[u] = hs[u] = 0.0f;
      if (PD_BWD_HANDOVER && sc.lane == 0) {
        float* rp = bnd.rec + (sc.seg_prev * N + n0 + u) * 2;
        hl[u] = atomicExch(rp, 0.0f);
        if (MIX) hs[u] = atomicExch(rp + 1, 0.0f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = n0 + u;
      buf_store(row_rsrc_bytes(plane_ptr(o.g_logits + (long)b * N * HW + (long)y * W, n, HW), gl_bytes), xoffs[u], outl[u] + hl[u]);
      if (MIX)
        buf_store(row_rsrc_bytes(plane_ptr(o.g_sigma + (long)b * N * HW + (long)y * W, n, HW), gs_bytes), xoffs[u], outs[u] + hs[u]);
    }
  }
  // Disparity gradient: wave totals into the row's LDS accumulators.  Two planes share one reduction: after
  // v_permlane32_swap the lower half-wave holds plane u's two half sums and the upper half plane u+1's, so five DPP
  // steps serve both (lanes 31 and 63 end up with the totals).  (Per-lane LDS partials, one ds_add_f32 per plane and
  // lane, measured 12% slower.)
  if (want_plane) {
#pragma unroll
    for (int u = 0; u + 1 < U; u += 2) {
      const float v = half_wave_sums_hi(gds[u], gds[u + 1]);
      if ((sc.lane & 31) == 31) lds_add(&red[n0 + u + (sc.lane >> 5)], v);
    }
    if (U & 1) {
      const float v = wave_sum_hi(gds[U - 1]);
      if (sc.lane == kWave - 1) lds_add(&red[n0 + U - 1], v);
    }
  }
}

template <bool MIX, bool HASMASK, int NROWS, bool RENDER = false>
__device__ __forceinline__ void rowshift_bwd_body(const SweepArgs& a, const BwdOut& o, const RowSel& row,
                                                  float* sdisp, int* kshift, float* red, const Boundary& bnd, float4* lrgb) {
  constexpr int U = PD_BWD_U;
  const int y = block_row(bwd_rowid(a.B, a.H), a.H), b = wg_image(a.B, a.H);
  const int HW = a.H * a.W, W = a.W, N = a.N;
  const int lane = threadIdx.x & (kWave - 1), nwaves = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: stays in SGPRs
  const int nseg = (W + kWave - 1) / kWave;
  // workgroup-uniform switches as scalars (as lane masks each costs two VALU instructions per plane and use)
  const int want_plane = __builtin_amdgcn_readfirstlane(o.g_plane != nullptr ? 1 : 0);
  const int gl_bytes = __builtin_amdgcn_readfirstlane(o.g_logits ? W * 4 : 0);  // 0: the stores become no-ops
  const int gs_bytes = __builtin_amdgcn_readfirstlane(o.g_sigma ? W * 4 : 0);
  stage_row_constants<NROWS>(a, b, row, lrgb, sdisp, y);
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
  const float gix_scale = (Wm1 / 2) * 2.0f / Wm1 * a.sign;  // d ix / d disp through un-normalise, *2, /(W-1)

  // balanced split of the row's (segment, plane chunk) list over the waves, as in the forward; the per-plane gradient
  // is closed-form given the pixel's stash, so a segment split between two waves needs no merge at all
  constexpr int G = HASMASK ? 32 : U;  // mask words are fetched whole
  static_assert(32 % U == 0, "plane groups must tile the 32-plane mask words");
  RowWork rw = row_work(nseg, N, G, wave, nwaves);
  if (RENDER) {   // the compositing state runs along the planes of a pixel: whole segments only (a ragged last round)
    rw.full = (nseg + nwaves - 1) / nwaves;
    rw.r = 0;
  }
  for (int it = 0;; ++it) {
    int seg, n_lo, n_hi, piece;
    if (!work_item(rw, it, wave, nwaves, N, G, seg, n_lo, n_hi, piece)) break;
    if (RENDER && seg >= nseg) break;   // (wave-uniform) nothing left in the ragged round
    SegCtx sc;
    sc.seg = seg;
    sc.seg_prev = (seg == 0) ? nseg - 1 : seg - 1;
    sc.T0 = seg * kWave;
    sc.xt = sc.T0 + lane;
    sc.lane = lane;
    sc.active = sc.xt < W;
    sc.last = min(kWave - 1, W - 1 - sc.T0);
    sc.pix = y * W + (sc.active ? sc.xt : 0);
    const PixelCtx c = sc.active ? make_pixel_ctx<MIX>(a, o, b, sc.pix, HW) : zero_pixel_ctx();
    RenderBwd rb;
    rb.Rtot = MIX ? -c.A * c.mx : c.gdotr;
    uint32_t bits = 0;
    // same software pipeline as the forward (the mask comes from the stash bits, not from memory)
    PlaneGroup<NROWS, U> g0, g1, g2;
    const int nfull = (n_hi - n_lo) / U;
#define PD_BISSUE(GR, I) group_issue<MIX, false, NROWS, U, RENDER>(GR, a, row, lbytes, sdisp, b, y, n_lo + (I) * U, sc.xt, HW, Wm1, rcpWm1)
#define PD_BCOMP(GR, I) bwd_compute<MIX, HASMASK, NROWS, U, RENDER>(GR, a, o, row, lbytes, kshift, red, bnd, b, y, n_lo + (I) * U, sc, c, HW, gix_scale, want_plane, gl_bytes, gs_bytes, bits, &rb)
    int gi = 0;
    if (PD_BWD_PF && PD_BWD_PF_DEPTH == 2) {
      if (nfull > 0) PD_BISSUE(g0, 0);
      for (; gi + 2 <= nfull; gi += 2) {
        PD_BISSUE(g1, gi + 1);
        PD_BCOMP(g0, gi);
        PD_BISSUE(g0, min(gi + 2, nfull - 1));  // unconditional: see the forward
        PD_BCOMP(g1, gi + 1);
      }
      if (gi < nfull) PD_BCOMP(g0, gi);
    } else if (PD_BWD_PF) {
      if (nfull > 0) PD_BISSUE(g0, 0);
      if (nfull > 1) PD_BISSUE(g1, 1);
      for (; gi + 3 <= nfull; gi += 3) {
        PD_BISSUE(g2, gi + 2);
        PD_BCOMP(g0, gi);
        PD_BISSUE(g0, min(gi + 3, nfull - 1));
        PD_BCOMP(g1, gi + 1);
        PD_BISSUE(g1, min(gi + 4, nfull - 1));
        PD_BCOMP(g2, gi + 2);
      }
      if (gi < nfull) PD_BCOMP(g0, gi);
      if (gi + 1 < nfull) PD_BCOMP(g1, gi + 1);
    } else {
      for (; gi < nfull; ++gi) { PD_BISSUE(g0, gi); PD_BCOMP(g0, gi); }
    }
#undef PD_BISSUE
#undef PD_BCOMP
    for (int n = n_lo + nfull * U; n < n_hi; ++n) {
      PlaneGroup<NROWS, 1> gr;
      group_issue<MIX, false, NROWS, 1, RENDER>(gr, a, row, lbytes, sdisp, b, y, n, sc.xt, HW, Wm1, rcpWm1);
      bwd_compute<MIX, HASMASK, NROWS, 1, RENDER>(gr, a, o, row, lbytes, kshift, red, bnd, b, y, n, sc, c, HW, gix_scale, want_plane, gl_bytes, gs_bytes, bits, &rb);
    }
  }
  __syncthreads();
  // Deferred segment-boundary contributions: record (seg, n, j) targets global slot g (ring), i.e. source (g+k) mod W.
  const int ntens = MIX ? 2 : 1;
  const int nrec = nseg * N * ntens;
  for (int i = threadIdx.x; i < nrec; i += blockDim.x) {
    const int tns = i % ntens, sn = i / ntens, n = sn % N, seg = sn / N;
    float* dst = (tns == 0) ? o.g_logits : o.g_sigma;
    if (!dst) continue;
    float v[3];
    v[1] = bnd.rec[sn * 2 + tns];
    v[0] = v[2] = 0.0f;
    if ((bnd.irr[sn >> 5] >> (sn & 31)) & 1u) {   // rare: the wave that handled (seg, n) took the general routing path
      v[0] = bnd.side[sn * 4 + tns * 2];
      v[2] = bnd.side[sn * 4 + tns * 2 + 1];
    }
    const int T0 = seg * kWave, last = min(kWave - 1, W - 1 - T0);
    const int k = kshift[n];
    float* drow = dst + ((long)b * N + n) * HW + (long)y * W;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (v[j] == 0.0f) continue;
      int g = (j == 0) ? T0 - 1 : T0 + last + j;   // j=1 -> last+1, j=2 -> last+2
      g = ((g % W) + W) % W;
      int xs = g + k;
      xs = (xs >= W) ? xs - W : ((xs < 0) ? xs + W : xs);
      unsafeAtomicAdd(drow + xs, v[j]);
    }
  }
  if (want_plane) {
    if (a.flags & PD_DISP_ROWS) {  // one disparity per (plane, row): this workgroup owns the whole sum
      for (int i = threadIdx.x; i < N; i += blockDim.x) o.g_plane[((long)b * N + i) * a.H + y] = red[i];
    } else {
      float* dstp = o.partials + ((long)b * a.H + y) * N;
      for (int i = threadIdx.x; i < N; i += blockDim.x) dstp[i] = red[i];
    }
  }
}

template <bool MIX, bool HASMASK, bool RENDER = false>
__global__ __launch_bounds__(kRowThreadsMax, PD_BWD_OCC) void rowshift_bwd_kernel(SweepArgs a, BwdOut o) {
  extern __shared__ float4 lds4[];
  // LDS: colour rows float4[2*(W+4)] | sdisp[N] | kshift[N] | red[N] | rec[nseg][N][2] | irr[ceil(nseg*N/32)]
  float* sdisp = reinterpret_cast<float*>(lds4 + 2 * (a.W + 4));
  int* kshift = reinterpret_cast<int*>(sdisp + a.N);
  float* red = sdisp + 2 * a.N;
  const int nsn = ((a.W + kWave - 1) / kWave) * a.N;
  Boundary bnd;
  bnd.rec = red + a.N;
  bnd.irr = reinterpret_cast<unsigned*>(bnd.rec + 2 * nsn);
  bnd.side = o.side + ((long)wg_image(a.B, a.H) * a.H + bwd_rowid(a.B, a.H)) * (4L * nsn);
  const RowSel row = two_row_form(make_row_sel(block_row(bwd_rowid(a.B, a.H), a.H), a.H), a.fast_rows != 0);
  if (row.nrows == 2) rowshift_bwd_body<MIX, HASMASK, 2, RENDER>(a, o, row, sdisp, kshift, red, bnd, lds4);
  else                rowshift_bwd_body<MIX, HASMASK, 1, RENDER>(a, o, row, sdisp, kshift, red, bnd, lds4);
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
// Waves per row-workgroup (each wave walks its share of the row's 64-lane segments).
static int row_threads(int W) {
  const int nseg = ceil_div(W, kWave);
  if (const int w = switches().row_waves) {  // tuning hook (PD_ROW_WAVES, read once)
    if (w >= 1 && w <= kRowThreadsMax / kWave) return (w < nseg ? w : nseg) * kWave;   // the kernels' launch bound
  }
  // Measured on MI355X (W=640, 10 segments): 4, 5, 8 and 10 waves per workgroup are within 3% of each other, 1-2
  // waves are 1.5-2.5x slower (too few waves in flight).  Take the largest divisor of nseg up to 8 for equal work
  // per wave; awkward (prime) segment counts fall back to 8 waves with a ragged last pass.
  // PMC (scripts/gpu_occ.sh): 5-wave workgroups kept 1.7-1.8 waves resident per SIMD, 4- and 8-wave ones 2.3-3.3
  // (the dispatcher deals a workgroup's waves to the SIMDs round-robin, so multiples of 4 pack; odd counts strand
  // slots) and ran 8-18% faster despite the uneven split of the 10 segments.
  const int waves = nseg >= 4 ? 4 : nseg;
  return waves * kWave;
}

bool rowshift_applicable(const pd_sweep_desc* d) {
  // LDS of the larger of the two kernels: the backward's boundary records, the forward's parked partial sums (row pairs
  // double them) — a shape that fits neither falls back to the general kernels instead of failing at launch
  const size_t colour = (size_t)(d->W + 4) * 2 * sizeof(float4);
  const size_t bwd = colour + ((size_t)3 * d->N + (size_t)ceil_div(d->W, kWave) * d->N * 3) * 4;
  const size_t fwd = colour + ((size_t)d->N + (size_t)(kRowThreadsMax / kWave) * 2 * 8 * kWave * 2) * 4;
  return d->mode == PD_WARP_DISP && !(d->flags & PD_DISP_DENSE) && d->H <= 65535 &&
         (long)d->N * d->H * d->W < (1L << 31) && bwd <= 160 * 1024 && fwd <= 160 * 1024;
}

size_t rowshift_bwd_workspace_floats(const pd_sweep_desc* d) {
  return (size_t)d->B * d->H * d->N * (1 + 4 * (size_t)ceil_div(d->W, kWave));   // partial sums + boundary spill
}

template <typename K>
static void allow_lds(K kernel, size_t shmem) {
  if (shmem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}


int rowshift_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash,
                 hipStream_t stream) {
  dim3 grid(d->H, d->B), block(row_threads(d->W));
  // parked partial softmax states exist only when segments are left over after the whole rounds (W=640: 10 segments on
  // 4 waves; W=1280: none — 32 KB less, which is a third resident workgroup per CU there)
  const int nwaves = block.x / kWave, nseg = ceil_div(d->W, kWave);
  const size_t park = (nseg % nwaves) ? (size_t)nwaves * 2 * 8 * kWave * (a.pairs ? 2 : 1) : (size_t)kWave;
  const size_t shmem = (size_t)(d->W + 4) * 2 * sizeof(float4) + ((size_t)d->N + park) * sizeof(float);
  const bool mix = (d->flags & PD_MIXTURE) != 0, hasmask = a.has_mask != 0, am = (d->flags & PD_AUTOMASK) != 0;
  const bool render = (d->flags & PD_RENDER_PROB) != 0;
#define PD_FWD_LAUNCH_R(M, K, A, R)                                                               \
  do {                                                                                            \
    allow_lds(rowshift_fwd_kernel<M, K, A, R>, shmem);                                            \
    rowshift_fwd_kernel<M, K, A, R><<<grid, block, shmem, stream>>>(a, rgb_rec, ph_map, stash);   \
  } while (0)
#define PD_FWD_LAUNCH(M, K, A)                                                              \
  do {                                                                                      \
    if (render) PD_FWD_LAUNCH_R(M, K, A, true); else PD_FWD_LAUNCH_R(M, K, A, false);       \
  } while (0)
  if (mix) {
    if (hasmask) { if (am) PD_FWD_LAUNCH(true, true, true); else PD_FWD_LAUNCH(true, true, false); }
    else         { if (am) PD_FWD_LAUNCH(true, false, true); else PD_FWD_LAUNCH(true, false, false); }
  } else {
    if (hasmask) PD_FWD_LAUNCH(false, true, false); else PD_FWD_LAUNCH(false, false, false);
  }
#undef PD_FWD_LAUNCH
#undef PD_FWD_LAUNCH_R
  return check_launch("rowshift_fwd_kernel");
}

int rowshift_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o_in, hipStream_t stream) {
  dim3 grid(d->H, d->B), block(row_threads(d->W));
  const int nseg = ceil_div(d->W, kWave);
  const size_t nsn = (size_t)nseg * d->N;
  const size_t shmem = (size_t)(d->W + 4) * 2 * sizeof(float4) + ((size_t)3 * d->N + nsn * 2 + (nsn + 31) / 32) * sizeof(float);
  BwdOut o = o_in;
  o.side = o_in.partials + (size_t)d->B * d->H * d->N;   // workspace: [B][H][N] partial sums | [B][H][nseg*N][4] spill
  {
    const bool mix = (d->flags & PD_MIXTURE) != 0, hasmask = a.has_mask != 0, render = (d->flags & PD_RENDER_PROB) != 0;
#define PD_BWD_LAUNCH(M, K, R)                                                         \
  do {                                                                                 \
    allow_lds(rowshift_bwd_kernel<M, K, R>, shmem);                                    \
    rowshift_bwd_kernel<M, K, R><<<grid, block, shmem, stream>>>(a, o);                \
  } while (0)
#define PD_BWD_PICK(M, K) do { if (render) PD_BWD_LAUNCH(M, K, true); else PD_BWD_LAUNCH(M, K, false); } while (0)
    if (mix) { if (hasmask) PD_BWD_PICK(true, true); else PD_BWD_PICK(true, false); }
    else     { if (hasmask) PD_BWD_PICK(false, true); else PD_BWD_PICK(false, false); }
#undef PD_BWD_PICK
#undef PD_BWD_LAUNCH
  }
  int rc = check_launch("rowshift_bwd_kernel");
  if (rc || !o.g_plane || (d->flags & PD_DISP_ROWS)) return rc;
  reduce_rows_kernel<<<dim3(d->N, d->B), kWave, 0, stream>>>(o.partials, o.g_plane, d->H, d->N);
  return check_launch("reduce_rows_kernel");
}

}  // namespace pd

// Test hook (not part of the public header): counts fast-vs-IEEE division mismatches over `count` samples.
extern "C" int pd_selftest_division(float Wm1, int count, float lo, float step, int* d_mismatches, void* stream) {
  pd::div_check_kernel<<<pd::ceil_div(count, 256), 256, 0, (hipStream_t)stream>>>(Wm1, count, lo, step, d_mismatches);
  return pd::check_launch("div_check_kernel");
}
