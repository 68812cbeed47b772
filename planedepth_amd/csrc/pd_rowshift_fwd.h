// Forward of the row-organised plane sweep (PD_WARP_DISP, one disparity per (image, plane) or per (image, plane, row)):
// the plane-group pipeline, the work split inside a row workgroup, the row pairs, and the per-row bodies.  Shared by
//   * pd_plane_sweep_rowshift.hip   — one workgroup per target row (any mask form, --render_probability), and
//   * pd_plane_sweep_rowpersist.hip — persistent workgroups that walk rows and stage the NEXT row's colour rows and
//     shifts by LDS-DMA while the current row's plane loop runs (the headline forward).
// The bodies take their row constants (colour rows, per-plane shifts) already staged in LDS; `CL` says how the colour
// rows lie there.
#pragma once
#include <stdlib.h>

#include "pd_rowshift_common.h"

namespace pd {

// ---- colour rows in LDS: two layouts ---------------------------------------------------------------------------------
// PackedColour: per live row W+4 float4 (r,g,b,-), two zero guard cells on each side (stage_row_constants); a tap pair is
//   two ds_read_b128.
// PlanarColour: per live row three channel arrays of W+8 floats, four zero guard cells on each side so that the interior
//   is 16-byte aligned — the layout an LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land contiguously) writes
//   straight from the [3,H,W] image; a tap pair is three 8-byte reads at 4-byte alignment (ds_read2_b32).
struct PackedColour {
  static constexpr bool kPlanar = false;
};
struct PlanarColour {
  static constexpr bool kPlanar = true;
  static constexpr int kGuard = 4;
  __host__ __device__ static __forceinline__ int chan_floats(int W) { return W + 2 * kGuard; }
  __host__ __device__ static __forceinline__ int row_floats(int W) { return 3 * chan_floats(W); }
};

struct __attribute__((packed, aligned(4))) F2a4 { float x, y; };   // 8 bytes at 4-byte alignment: one ds_read2_b32

template <class CL, int NROWS>
__device__ __forceinline__ ColourTaps<NROWS> colour_taps_at(const char* __restrict__ base, int W, int x0) {
  if constexpr (!CL::kPlanar) {
    return load_colour_taps<NROWS>(base, W, colour_off(x0, W));
  } else {
    ColourTaps<NROWS> c;
    const int xc = min(max(x0, -2), W) + PlanarColour::kGuard;
    const int cs = PlanarColour::chan_floats(W) * 4;
    const char* p = base + (xc << 2);
    const F2a4 r = *reinterpret_cast<const F2a4*>(p), g = *reinterpret_cast<const F2a4*>(p + cs),
               b = *reinterpret_cast<const F2a4*>(p + 2 * cs);
    c.nw = make_float4(r.x, g.x, b.x, 0.0f);
    c.ne = make_float4(r.y, g.y, b.y, 0.0f);
    if (NROWS == 2) {
      const char* q = p + 3 * cs;
      const F2a4 r2 = *reinterpret_cast<const F2a4*>(q), g2 = *reinterpret_cast<const F2a4*>(q + cs),
                 b2 = *reinterpret_cast<const F2a4*>(q + 2 * cs);
      c.sw = make_float4(r2.x, g2.x, b2.x, 0.0f);
      c.se = make_float4(r2.y, g2.y, b2.y, 0.0f);
    }
    return c;
  }
}

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
template <bool MIX, bool HASMASK, int NROWS, int U, bool RENDER = false, class CL = PackedColour>
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
    if (!(kAblate & 2)) g.tc[u] = colour_taps_at<CL, NROWS>(lrgb, a.W, g.ct[u].x0);
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

template <bool MIX, bool HASMASK, int NROWS, int U, bool RENDER = false, class CL = PackedColour>
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
    if (!(kAblate & 2)) tc[u] = colour_taps_at<CL, NROWS>(lrgb, a.W, g.ct[u].x0);
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
__host__ __device__ __forceinline__ void slice_of(int r, int cps, int w, int nwaves, int& cb, int& ce) {
  cb = r * cps * w / nwaves;
  ce = r * cps * (w + 1) / nwaves;
}
// Host side: the largest number of left-over segments any wave's slice touches (1 or 2; 0 without left-over segments) for
// plane chunks of G planes — sizes the parked partial sums.
static inline int fwd_pieces_per_wave(int nseg, int nwaves, int N, int G) {
  const int r = nseg % nwaves, cps = (N + G - 1) / G;
  int most = 0;
  for (int w = 0; w < nwaves && r; ++w) {
    int cb, ce;
    slice_of(r, cps, w, nwaves, cb, ce);
    int n = 0;
    for (int j = 0; j < r; ++j) n += (ce > cb && cb < (j + 1) * cps && ce > j * cps) ? 1 : 0;
    most = n > most ? n : most;
  }
  return most;
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

// One target row (b, y) whose row constants are staged: `lbytes` = the live source colour rows in layout CL, `sdisp` = the
// per-plane shifts, `parts` = room for the parked partial sums, [nwaves][ppw pieces][8][64] floats (ppw = 2 always suffices;
// fwd_pieces_per_wave() says when 1 does).  Returns this lane's share of sum(ph_map).  Contains workgroup barriers when segments are left over
// after the whole rounds (all threads of the workgroup must call it).
template <bool MIX, bool HASMASK, bool AUTO, int NROWS, bool RENDER = false, class CL = PackedColour>
__device__ __forceinline__ float rowshift_fwd_rows(const SweepArgs& a, const RowSel& row, int y, int b,
                                                  const char* __restrict__ lbytes, const float* __restrict__ sdisp,
                                                  float* parts, float* __restrict__ rgb_rec,
                                                  float* __restrict__ ph_map, float* __restrict__ stash, int ppw = 2) {
  // (compositing keeps a distance per plane of the group and its running state alive: half the group size, or the
  // register allocator spills 40-70 VGPRs at the 168 this kernel may use)
  constexpr int UB = (RENDER && PD_FWD_U > 1) ? PD_FWD_U / 2 : PD_FWD_U;
  constexpr int U = (NROWS == 1) ? UB : (UB > 1 ? UB / 2 : 1);
  constexpr int G = HASMASK ? 32 : U;  // chunk of the work split (mask words of the stash are written whole)
  static_assert(32 % U == 0, "plane groups must tile the 32-plane mask words");
  const int HW = a.H * a.W, N = a.N;
  // mixture kernels are specialised on the automask flag (it costs an exponential per plane); L1 reads it at run time
  const bool automask = MIX ? AUTO : (bool)(a.flags & PD_AUTOMASK);
  const float Wm1 = (float)(a.W - 1), rcpWm1 = refined_rcp(Wm1);
  const float* srcb = a.src + (long)b * 3 * HW;
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
#define PD_FISSUE(GR, I) group_issue<MIX, HASMASK, NROWS, U, RENDER, CL>(GR, a, row, lbytes, sdisp, b, y, n_lo + (I) * U, x, HW, Wm1, rcpWm1)
#define PD_FCOMP(GR, I) fwd_compute<MIX, HASMASK, NROWS, U, RENDER, CL>(GR, a, row, lbytes, b, n_lo + (I) * U, pix, HW, t0, t1, t2, ea, automask, acc, bits, stash, &rs)
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
      group_issue<MIX, HASMASK, NROWS, 1, RENDER, CL>(gr, a, row, lbytes, sdisp, b, y, n, x, HW, Wm1, rcpWm1);
      fwd_compute<MIX, HASMASK, NROWS, 1, RENDER, CL>(gr, a, row, lbytes, b, n, pix, HW, t0, t1, t2, ea, automask, acc, bits, stash, &rs);
    }
    if (piece < 0) ph_sum += fwd_store<MIX>(a, acc, b, pix, HW, t0, t1, t2, ea, automask, rgb_rec, ph_map, stash, !RENDER);
    else park_acc(parts + ((wave * ppw + piece) * 8) * kWave, lane, acc);
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
          acc = merge_acc(acc, fetch_acc(parts + ((w2 * ppw + (cb2 < j * rw.cps ? 1 : 0)) * 8) * kWave, lane));
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
// (PairRole, row_lean, leads, pair_role, PairW, pair_weights: pd_rowgeom.h — shared with the row-stream backward)

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

// Source rows of the pair (leader yL, partner yP) as the two-row loaders see them: "rows A and B" with unit weights.
__device__ __forceinline__ RowSel pair_rows(int yL, int yP) {
  RowSel rows;
  rows.nrows = 2; rows.yA = yL; rows.yB = yP; rows.wA = rows.wB = rows.wy_main = 1.0f;
  return rows;
}

// Both target rows of a pair; row constants staged as for a two-row row with source rows (yL, yP) (pair_rows).
template <bool MIX, bool AUTO, class CL = PackedColour>
__device__ __forceinline__ float rowpair_fwd_rows(const SweepArgs& a, int yL, int yP, int b,
                                                  const char* __restrict__ lbytes, const float* __restrict__ sdisp,
                                                  float* parts, float* __restrict__ rgb_rec,
                                                  float* __restrict__ ph_map, float* __restrict__ stash, int ppw = 2) {
  static_assert(PD_TC_IN_GROUP, "the pair bodies use the two-group pipeline with colour taps in the group");
  constexpr int U = PD_FWD_U > 1 ? PD_FWD_U / 2 : 1;  // planes per group; each carries both rows
  constexpr int G = U;
  const int HW = a.H * a.W, N = a.N;
  const bool automask = MIX ? AUTO : (bool)(a.flags & PD_AUTOMASK);
  const float Wm1 = (float)(a.W - 1), rcpWm1 = refined_rcp(Wm1);
  const float* srcb = a.src + (long)b * 3 * HW;
  const PairW pw = pair_weights(yL, yP, a.H);
  const RowSel rows = pair_rows(yL, yP);  // the two-row loaders' "source rows A and B" are the leader's row and the partner's row
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
  auto slot = [&](int w, int piece, int r) { return parts + (((r * nwaves + w) * ppw + piece) * 8) * kWave; };
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
#define PD_PISSUE(GR, I) group_issue<MIX, false, 2, U, false, CL>(GR, a, rows, lbytes, sdisp, b, yL, n_lo + (I) * U, x, HW, Wm1, rcpWm1)
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
        group_issue<MIX, false, 2, 1, false, CL>(gr, a, rows, lbytes, sdisp, b, yL, n, x, HW, Wm1, rcpWm1);
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


}  // namespace pd
