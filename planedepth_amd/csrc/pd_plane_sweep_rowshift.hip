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

#include "pd_rowshift_fwd.h"

namespace pd {

// ---------------------------------------------------------------------------------------------------------------
// Forward: the plane-group pipeline, work split, row pairs and per-row bodies live in pd_rowshift_fwd.h (shared with the
// persistent forward, pd_plane_sweep_rowpersist.hip); this kernel is the one-workgroup-per-target-row form.
// ---------------------------------------------------------------------------------------------------------------
// LDS: float4 colour[2*(W+4)] | float sdisp[N] | parked partial sums
template <bool MIX, bool HASMASK, bool AUTO, bool RENDER = false>
__global__ __launch_bounds__(kRowThreadsMax, HASMASK ? 3 : PD_FWD_OCC) void rowshift_fwd_kernel(SweepArgs a, float* __restrict__ rgb_rec,
                                                                     float* __restrict__ ph_map,
                                                                     float* __restrict__ stash) {
  extern __shared__ float4 lds4[];
  // LDS: colour rows float4[2*(W+4)] | sdisp[N] | parked partial sums [nwaves][2][8][64]
  float* sdisp = reinterpret_cast<float*>(lds4 + 2 * (a.W + 4));
  float* parts = sdisp + a.N;
  const int y = block_row(wg_rowid(a.B, a.H), a.H);
  const RowSel row = two_row_form(make_row_sel(y, a.H), a.row_eps);
  float ph_sum = 0.0f;
  int partner = y;
  // row pairs: per-plane scalar disparities and no per-pixel mask (then the sampling column is shared by the rows)
  const PairRole role = (a.pairs && !HASMASK && !RENDER) ? pair_role(y, a.H, partner) : kSingle;
  const int b = wg_image(a.B, a.H);
  const char* lbytes = reinterpret_cast<const char*>(lds4);
  if (role == kAbsorbed) {
    // this row is computed by its neighbour's workgroup
  } else if (role == kLeader) {
    if (!HASMASK && !RENDER) {
      stage_row_constants<2>(a, b, pair_rows(y, partner), lds4, sdisp, y);
      __syncthreads();
      ph_sum = rowpair_fwd_rows<MIX, AUTO>(a, y, partner, b, lbytes, sdisp, parts, rgb_rec, ph_map, stash);
    }
  } else if (row.nrows == 2) {
    stage_row_constants<2>(a, b, row, lds4, sdisp, y);
    __syncthreads();
    ph_sum = rowshift_fwd_rows<MIX, HASMASK, AUTO, 2, RENDER>(a, row, y, b, lbytes, sdisp, parts, rgb_rec, ph_map, stash);
  } else {
    stage_row_constants<1>(a, b, row, lds4, sdisp, y);
    __syncthreads();
    ph_sum = rowshift_fwd_rows<MIX, HASMASK, AUTO, 1, RENDER>(a, row, y, b, lbytes, sdisp, parts, rgb_rec, ph_map, stash);
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
  const RowSel row = two_row_form(make_row_sel(block_row(bwd_rowid(a.B, a.H), a.H), a.H), a.row_eps);
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
         (long)d->N * d->H * d->W < (1L << 31) && bwd <= device_lds_bytes() && fwd <= device_lds_bytes();
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
