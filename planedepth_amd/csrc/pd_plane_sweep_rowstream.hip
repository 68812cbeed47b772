// Row-stream backward of the fused plane sweep: PD_WARP_DISP with one disparity per (image, plane) or per (image,
// plane, row), no per-pixel mask, softmax probabilities (reference trainer.py:540-603 + 728-742 through autograd).
//
// The row-shift backward (pd_plane_sweep_rowshift.hip) gives every lane a TARGET pixel and routes the adjoint of the
// 2-tap gather to its source slot: loads and gradient stores start at x + k, k = floor(s*d) — 4-byte-aligned runs that
// straddle cache lines, one 4-byte store per lane, hand-overs between the segments of different waves.  Measured
// (scripts/probes/stream_probe.hip, 8x49x192x640): the memory pipeline of gfx950 is paced by instruction count — that
// shape tops out at 4.0-4.4 TB/s, 12-byte loads + aligned 8-byte stores reach 5.4 TB/s.  This kernel turns the
// ownership around:
//   * a lane owns two consecutive, 8-byte-ALIGNED SOURCE slots xs, xs+1 of the row; slot xs is paired with the target
//     pixel xt = xs - k, whose left tap is xs.  The taps (L[xs..xs+2]) are one aligned 12-byte load per tensor, the
//     gradient pair one aligned 8-byte store: full cache lines, no ring, no wrap, no edge fix-up, no validity masks
//     (a slot whose target is outside the row reads an all-zero context and stores the zero it has to store);
//   * what shifts with the plane is the per-target-pixel context (target colour, softmax statistics, upstream
//     gradients: 12 floats).  It is staged once per row in LDS and read at xs - k;
//   * a wave walks the 128-slot segments of one (plane, row) IN ORDER, so the right-tap contribution that leaves a
//     segment is carried to the next one in a register (v_readlane) — no LDS parking, no atomics, no neighbour waves.
//     The waves of a workgroup split the row's (plane, segment) list into contiguous ranges; the at most nwaves-1
//     range boundaries that fall inside a row are settled with one atomic per tensor after the loop;
//   * the disparity gradient is accumulated per lane over the whole row and reduced once per plane and wave.
//
// Exactness.  Slot xs is paired with target xt = xs - k on the premise floor(ix(xt)) = xt + k.  The reference's fp32
// coordinate chain can move ix across an integer when frac(s*d) is within ~4e-7 * W of 0 or 1 (pd_plane_sweep_rowshift's
// header): planes with frac(s*d) closer than kIrrTol(W) to an integer ("irregular", decided when the shifts are
// staged) take a general per-lane path instead — exact floor(ix) per target, gradient rows zero-filled up front,
// contributions added with atomics (two addends per slot: the order cannot change the sum).  For every other plane
// the premise holds with a margin of 2.4-3x the worst-case rounding error of the chain (bound in NOTEBOOK.md 3.6.3).
// Targets whose left tap is column -1 (negative shifts) have no slot: a short epilogue serves them (lanes = planes).
//
// Vertical: rows whose y round trip is inexact blend two source rows (weights 1-eps, eps); values and per-pixel
// gradients use both, the adjoint applies the own row's weight and drops the eps-weighted neighbour term exactly as
// the row-shift kernels do (same bound, same tests).
#include <stdlib.h>

#include "pd_rowshift_common.h"
#include "pd_tail_common.h"

namespace pd {

#ifndef PD_STREAM_D1
#define PD_STREAM_D1 3   // prefetch depth in (plane, segment) iterations, one live source row
#endif
#ifndef PD_STREAM_D2
#define PD_STREAM_D2 1   // two live source rows: twice the registers per group; depth 2 costs the third resident workgroup (90 vs 77 VGPRs)
#endif
#ifndef PD_STREAM_WAVES
#define PD_STREAM_WAVES 8   // waves per row workgroup: 42 KB of LDS per 640-pixel row -> three workgroups = 24 waves per CU
                            // (4 / 12 / 16 waves measured 0.193 / 0.183 / 0.230 ms against 0.176-0.186; rows wider than ~800: 2x)
#endif
#ifndef PD_STREAM_ABL
#define PD_STREAM_ABL 0  // timing experiments only (wrong results): 1 no context reads, 2 no per-plane gradient math,
#endif                   // 4 every row as one source row, 8 no gradient stores, 16 no coordinate chain
#ifndef PD_DIAGNOSTICS   // timing-ablation / trace code (results wrong by design) compiles only into a library that says so: pd_build_flags()
#if PD_STREAM_ABL
#error "timing-ablation / trace switches need -DPD_DIAGNOSTICS as well (pd_build_flags() then reports the build)"
#endif
#endif
constexpr int kStreamAbl = PD_STREAM_ABL;
#ifndef PD_STREAM_STORE_AUX
#define PD_STREAM_STORE_AUX 2   // cache-policy bits of the gradient stores (1 = sc0, 2 = nt, 16 = sc1; 0 = write-back).  nt: the
#endif                          // 385 MB of gradients stream past the caches instead of leaving ~256 MB of dirty lines behind for
                                // the next kernels to evict.  Measured (round 4, docs/archive/scripts/gpu_r4_nt.sh): in the hot-path loop this
                                // kernel pays its own writes (0.178 -> 0.194 ms) and the forward that follows stops paying them
                                // (0.126-0.133 -> 0.108 ms): step +2-3 %; inside the DDP training step, where the fused decoder
                                // tail's backward is the consumer, BOTH get faster (this kernel 0.177-0.195 -> 0.170 ms, the
                                // tail's backward 0.241 -> 0.224 ms)
#ifndef PD_STREAM_LOAD_AUX
#define PD_STREAM_LOAD_AUX 0    // same for the tap loads
#endif
#ifndef PD_STREAM_DEAD
#define PD_STREAM_DEAD 1  // (plane, segment) items none of whose slots has a target inside the row (shift beyond the segment: the near planes'
#endif                    // leading segments) issue no memory reads (zero-extent descriptors), read no context, do no arithmetic and store
                          // the zeros they have to store: 11 of the 245 items of a headline row (round 6)
#ifndef PD_STREAM_PRIO
#define PD_STREAM_PRIO 1  // wave priority of a row workgroup's phases (s_setprio): 1 = the staging and the ring's first loads at 3, the item loop
#endif                    // at 0 — a workgroup that has just arrived on the CU gets through its dependent round trips ahead of the two that are
                          // streaming (-0.9 % alone, -2.2 % with the max-ilp scheduler, __graft_entry__.FILE_FLAGS); 2 = the first D+1 items at 3
                          // as well (where -mllvm -amdgpu-set-wave-priority puts it: the same within the noise); 0 = none.  Measured and dropped
                          // (NOTEBOOK 11.5): a rotating leader per SIMD as in the forward (+13 %), loads above arithmetic or below it (+1-4 %)
#ifndef PD_STREAM_OCC
#define PD_STREAM_OCC 4  // launch bound (1024 threads): the allocator's cap is 128 VGPRs; the kernel uses 76 = 6 waves per SIMD
#endif

constexpr int kSlots = 2;               // source slots per lane
constexpr int kSeg = kWave * kSlots;    // slots per wave iteration
constexpr int kStreamThreadsMax = 1024;

typedef float v3f __attribute__((ext_vector_type(3)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v3f buf_load3(Rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, (int)voff, (int)soff, PD_STREAM_LOAD_AUX));
}
__device__ __forceinline__ void buf_store2(Rsrc r, unsigned voff, unsigned soff, float x, float y) {
  __builtin_amdgcn_raw_buffer_store_b64(v2u{__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y)}, r, (int)voff,
                                        (int)soff, PD_STREAM_STORE_AUX);
}

struct StreamLds {
  float4* ctx0;   // [CW] (t0, t1, t2, lse2)            cell = pixel + 2, zero-gradient guard cells around the row
  float4* ctx1;   // [CW] (gr0, gr1, gr2, gdotr)
  float4* ctx2;   // [CW] (invS, mx, A, -)
  float4* col;    // [CW] vertically blended source colour (r, g, b, -), zero beyond the row
  float2* colgb;  // PK layout (rows too wide for two workgroups per CU otherwise): (g, b) here, r in ctx2[].w, `col` unused
  int2* shift;    // [N]  (bits of s*d clamped, k << 1 | irregular) — integers: a float-typed slot may flush the denormal pattern
  float* red;     // [N]  disparity-gradient sums of the row
  float* hand;    // [nwaves][2] carries that leave a wave's range in the middle of a row
  int* special;   // [1]  any plane with a negative shift or an irregular one (the epilogue has work)
  float4* tail;   // TAIL: [CW] per SOURCE pixel (lse of the decoder's logits, 1 / sum pi/sigma, disp, d loss / d disp)
  float* dpl;     // TAIL: [N]  the planes' disparities (unsigned, unclamped: the decoder's disp_layered)
  int* live;      // [N]  PD_STREAM_DEAD: first live segment | (one past the last live segment) << 16
  int CW;
};

// The decoder tail's backward at one source element (networks/depth_decoder.py:258-291; pd_decoder_tail.hip: tail_bwd_kernel,
// no padding mask): with the decoder's own logit l and sigma sg = clamp(sigmoid(raw)) AT that element, P = softmax(l) / sg /
// sum(pi/sigma) and the upstream gD of disp (+ depth's share), t = gD (d_n - disp) P:
//   g_raw_logits = g_logits + t;   g_raw_sigma = (g_sigma - t / sg) sigmoid'(raw) where the clamp passed sg through, else 0.
// sigmoid' = sgu (1 - sgu) with sgu = sg wherever 0.01 < sg < 1; sg == 1 means sgu == 1 (sigmoid never exceeds it): the
// product is 0 either way; sg == 0.01 is the one case that needs raw_sigma (sgu < 0.01: gate closed; == 0.01: open).
struct TailTerm {
  float t;        // added to the logit gradient
  float fs;       // factor of the sigma gradient: sigmoid'(raw) * gate
  float tos;      // t / sg
  float gdl;      // gD * P: this element's share of d loss / d d_n through disp
};
__device__ __forceinline__ TailTerm tail_term(const float4 tc, float l, float sg, float dn, const float* __restrict__ raw_sigma_at) {
  TailTerm o;
  const float p = __expf(l - tc.x);
  const float rsg = fast_rcp(fmaxf(sg, kTailSigmaMin));   // (sg >= 0.01 inside the row; slots beyond it read 0 and carry tc = 0)
  const float P = p * rsg * tc.y;
  o.gdl = tc.w * P;
  o.t = o.gdl * (dn - tc.z);
  o.tos = o.t * rsg;
  o.fs = sg * (1.0f - sg);
  if (sg == kTailSigmaMin) {   // rare: the clamp's lower bound — the gate needs the unclamped value
    const float sgu = sigmoid_f(*raw_sigma_at);
    o.fs = (sgu == sg) ? o.fs : 0.0f;
  }
  return o;
}
// the sigma factor alone (contributions that reach an element on their own: atomics of the irregular planes, hand-overs)
__device__ __forceinline__ float tail_fs(float sg, const float* __restrict__ raw_sigma_at) {
  float fs = sg * (1.0f - sg);
  if (sg == kTailSigmaMin) fs = (sigmoid_f(*raw_sigma_at) == sg) ? fs : 0.0f;
  return fs;
}

__device__ __forceinline__ PixelCtx ctx_at(const StreamLds& L, int cell) {
  const float4 a = L.ctx0[cell], g = L.ctx1[cell], h = L.ctx2[cell];
  PixelCtx c;
  c.t0 = a.x; c.t1 = a.y; c.t2 = a.z; c.lse2 = a.w;
  c.gr0 = g.x; c.gr1 = g.y; c.gr2 = g.z; c.gdotr = g.w;
  c.invS = h.x; c.mx = h.y; c.A = h.z;
  return c;
}

// blended source colour of cell i; PK: 56 instead of 64 bytes of LDS per cell (the two spare floats of the plain layout gone)
template <bool PK>
__device__ __forceinline__ float4 col_at(const StreamLds& L, int i) {
  if (PK) {
    const float r = L.ctx2[i].w;
    const float2 gb = L.colgb[i];
    return make_float4(r, gb.x, gb.y, 0.0f);
  }
  return L.col[i];
}

template <int NROWS>
struct StreamGroup {   // the taps of one (plane, segment) iteration: L[xs .. xs+2] per live source row
  float l[NROWS][3], s[NROWS][3];
};

struct StreamRow {     // workgroup-uniform
  int b, y, yA, yB;
  float wA, wB, wy;    // vertical weights of the two source rows; adjoint weight of the own row
};

template <bool MIX, int NROWS>
__device__ __forceinline__ void stream_issue(StreamGroup<NROWS>& g, const SweepArgs& a, const StreamRow& r, int n, int seg,
                                             unsigned lane8, int HW, int ext = -1) {
  // ext: the row descriptors' extent in floats (wave-uniform); 0 turns the item's loads into hardware no-ops that return zeros
  const int Wx = ext < 0 ? a.W : ext;
  const unsigned soff = (unsigned)seg * (kSeg * 4);
  const float* pl = plane_ptr(a.logits + (long)r.b * a.N * HW, n, HW);
  const v3f la = buf_load3(row_rsrc(pl + (long)r.yA * a.W, Wx), lane8, soff);
  g.l[0][0] = la.x; g.l[0][1] = la.y; g.l[0][2] = la.z;
  if (NROWS == 2) {
    const v3f lb = buf_load3(row_rsrc(pl + (long)r.yB * a.W, Wx), lane8, soff);
    g.l[NROWS - 1][0] = lb.x; g.l[NROWS - 1][1] = lb.y; g.l[NROWS - 1][2] = lb.z;
  }
  if (MIX) {
    const float* ps = plane_ptr(a.sigma + (long)r.b * a.N * HW, n, HW);
    const v3f sa = buf_load3(row_rsrc(ps + (long)r.yA * a.W, Wx), lane8, soff);
    g.s[0][0] = sa.x; g.s[0][1] = sa.y; g.s[0][2] = sa.z;
    if (NROWS == 2) {
      const v3f sb = buf_load3(row_rsrc(ps + (long)r.yB * a.W, Wx), lane8, soff);
      g.s[NROWS - 1][0] = sb.x; g.s[NROWS - 1][1] = sb.y; g.s[NROWS - 1][2] = sb.z;
    }
  }
}

// One regular (plane, segment) iteration.  carry_*: right-tap contribution of the previous segment's last slot (wave
// uniform); returns this segment's in the same variables.
template <bool MIX, int NROWS, bool PK, bool TAIL>
__device__ __forceinline__ void stream_compute(const StreamGroup<NROWS>& g, const SweepArgs& a, const BwdOut& o,
                                               const StreamRow& r, const StreamLds& L, int n, int seg, int k, float sd,
                                               int lane, unsigned lane8, float lane2f, int HW, float Wm1, float rcpWm1,
                                               int want_plane, int gl_bytes, int gs_bytes, float& carry_l,
                                               float& carry_s, float& gacc) {
  const int xs0 = seg * kSeg + lane * kSlots;
  const float xs0f = (float)(seg * kSeg) + lane2f;
  const float xt0f = xs0f - (float)k;   // integers below 2^24: exact
  // context of the two paired targets xt, xt+1: adjacent cells (guard cells two deep on both sides keep them adjacent)
  // (PD_STREAM_ABL & 32, timing only: cells at a 16-byte lane stride — the context reads without their 2-way bank conflict)
  const int cell = (kStreamAbl & 32) ? min(max((xs0 >> 1) - k, -2), a.W) + 2 : min(max(xs0 - k, -2), a.W) + 2;
  const float4 cv0 = col_at<PK>(L, xs0 + 2), cv1 = col_at<PK>(L, xs0 + 3), cv2 = col_at<PK>(L, xs0 + 4);
  float cl0[kSlots], cl1[kSlots], cs0[kSlots], cs1[kSlots];
#pragma unroll
  for (int i = 0; i < kSlots; ++i) {
    PixelCtx c;
    if (kStreamAbl & 1) { c = zero_pixel_ctx(); c.t0 = lane2f; c.gr0 = sd; c.A = xs0f; } else c = ctx_at(L, cell + i);
    const float xsf = xs0f + (float)i;
    const float ix = (kStreamAbl & 16) ? xsf + 0.25f : stream_ix(xt0f + (float)i, sd, Wm1, rcpWm1);
    const float w1 = ix - xsf, w0 = (xsf + 1.0f) - ix;   // torch's (x1 - ix), (ix - x0) with x0 = xs
    float l, s = 0.0f, dlx, dsx = 0.0f;
    if (NROWS == 1) {
      l = g.l[0][i] * w0 + g.l[0][i + 1] * w1;
      dlx = g.l[0][i + 1] - g.l[0][i];
      if (MIX) {
        s = g.s[0][i] * w0 + g.s[0][i + 1] * w1;
        dsx = g.s[0][i + 1] - g.s[0][i];
      }
    } else {
      const float a0 = w0 * r.wA, a1 = w1 * r.wA, b0 = w0 * r.wB, b1 = w1 * r.wB;
      l = g.l[0][i] * a0 + g.l[0][i + 1] * a1 + g.l[NROWS - 1][i] * b0 + g.l[NROWS - 1][i + 1] * b1;
      dlx = (g.l[0][i + 1] - g.l[0][i]) * r.wA + (g.l[NROWS - 1][i + 1] - g.l[NROWS - 1][i]) * r.wB;
      if (MIX) {
        s = g.s[0][i] * a0 + g.s[0][i + 1] * a1 + g.s[NROWS - 1][i] * b0 + g.s[NROWS - 1][i + 1] * b1;
        dsx = (g.s[0][i + 1] - g.s[0][i]) * r.wA + (g.s[NROWS - 1][i + 1] - g.s[NROWS - 1][i]) * r.wB;
      }
    }
    const float4 ca = (i == 0) ? cv0 : cv1, cb = (i == 0) ? cv1 : cv2;
    const float c0 = ca.x * w0 + cb.x * w1, c1 = ca.y * w0 + cb.y * w1, c2 = ca.z * w0 + cb.z * w1;
    PlaneGrad pg;
    if (kStreamAbl & 2) { pg.g_l = l + c.t0; pg.g_s = s + c.gr0; pg.gc0 = c0 + c.A; pg.gc1 = c1; pg.gc2 = c2; }
    else pg = plane_grad<MIX>(c, l, s, c0, c1, c2);
    const float m0 = (NROWS == 1) ? w0 : w0 * r.wy, m1 = (NROWS == 1) ? w1 : w1 * r.wy;
    cl0[i] = pg.g_l * m0; cl1[i] = pg.g_l * m1;
    cs0[i] = pg.g_s * m0; cs1[i] = pg.g_s * m1;
    if (want_plane)
      gacc += pg.g_l * dlx + pg.g_s * dsx + pg.gc0 * (cb.x - ca.x) + pg.gc1 * (cb.y - ca.y) + pg.gc2 * (cb.z - ca.z);
  }
  // slot xs receives the left-tap part of its own target and the right-tap part of the target one slot to the left
  const float pl = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry_l), __float_as_int(cl1[kSlots - 1]), 0x138, 0xF, 0xF, false));
  float out_l0 = cl0[0] + pl, out_l1 = cl0[1] + cl1[0];
  carry_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cl1[kSlots - 1]), kWave - 1));   // (an int builtin)
  float out_s0 = 0.0f, out_s1 = 0.0f;
  if (MIX) {
    const float ps = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry_s), __float_as_int(cs1[kSlots - 1]), 0x138, 0xF, 0xF, false));
    out_s0 = cs0[0] + ps; out_s1 = cs0[1] + cs1[0];
    carry_s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs1[kSlots - 1]), kWave - 1));
  }
  if (TAIL) {   // the decoder tail's backward at the lane's two source elements (tail_term): the stores below then carry the
                // gradients of the decoder's conv outputs.  The element's own logit / sigma are the taps of its own row.
    const int own = (NROWS == 2 && r.yA != r.y) ? NROWS - 1 : 0;
    const float dn = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(L.dpl[n])));
    const float* rs = o.tail_raw_sigma + ((long)r.b * a.N + n) * HW + (long)r.y * a.W + xs0;
    const TailTerm t0 = tail_term(L.tail[xs0 + 2], g.l[own][0], g.s[own][0], dn, rs);
    const TailTerm t1 = tail_term(L.tail[xs0 + 3], g.l[own][1], g.s[own][1], dn, rs + 1);
    out_l0 += t0.t; out_l1 += t1.t;
    out_s0 = (out_s0 - t0.tos) * t0.fs; out_s1 = (out_s1 - t1.tos) * t1.fs;
    if (want_plane) gacc += a.sign * (t0.gdl + t1.gdl);   // (the row's sum is scaled by d ix / d disp = sign at the end: sign^2 = 1)
  }
  const unsigned soff = (unsigned)seg * (kSeg * 4);
  if (!(kStreamAbl & 8) || out_l0 == 123.456f)
  buf_store2(row_rsrc_bytes(plane_ptr(o.g_logits + (long)r.b * a.N * HW + (long)r.y * a.W, n, HW), gl_bytes), lane8, soff, out_l0, out_l1);
  if (MIX) {
    if (!(kStreamAbl & 8) || out_s0 == 123.456f)
    buf_store2(row_rsrc_bytes(plane_ptr(o.g_sigma + (long)r.b * a.N * HW + (long)r.y * a.W, n, HW), gs_bytes), lane8, soff, out_s0, out_s1);
  }
}

// General form for one paired slot xs of plane n (n may differ per lane): the target xt = xs - k with its EXACT
// floor(ix); contributions go to the gradient rows with atomics, the disparity-gradient term is returned.
// Used for irregular planes (rows zero-filled up front) and for the virtual slots of the epilogue.
template <bool MIX, int NROWS, bool PK, bool TAIL>
__device__ __forceinline__ float stream_general_slot(   // (as a call: 123 VGPRs + scratch — measured, NOTEBOOK.md 3.6.4)
    const SweepArgs& a, const BwdOut& o, const StreamRow& r,
                                                     const StreamLds& L, int n, int xs, int k, float sd, bool on, int HW,
                                                     float Wm1, float rcpWm1) {
  const int W = a.W;
  const int xt = xs - k;
  on = on && xt >= 0 && xt < W;
  const ColTap t = make_col_tap((float)xt + sd, Wm1, rcpWm1);
  const bool v0 = on && t.x0 >= 0 && t.x0 < W, v1 = on && t.x0 + 1 >= 0 && t.x0 + 1 < W;
  if (!(v0 || v1)) return 0.0f;   // nothing of this target is in view: no samples, no gradient, no disparity term
  const long rowA = ((long)r.b * a.N + n) * HW + (long)r.yA * W, rowB = ((long)r.b * a.N + n) * HW + (long)r.yB * W;
  const float la0 = v0 ? a.logits[rowA + t.x0] : 0.0f, la1 = v1 ? a.logits[rowA + t.x0 + 1] : 0.0f;
  float lb0 = 0.0f, lb1 = 0.0f, sa0 = 0.0f, sa1 = 0.0f, sb0 = 0.0f, sb1 = 0.0f;
  if (NROWS == 2) { lb0 = v0 ? a.logits[rowB + t.x0] : 0.0f; lb1 = v1 ? a.logits[rowB + t.x0 + 1] : 0.0f; }
  if (MIX) {
    sa0 = v0 ? a.sigma[rowA + t.x0] : 0.0f; sa1 = v1 ? a.sigma[rowA + t.x0 + 1] : 0.0f;
    if (NROWS == 2) { sb0 = v0 ? a.sigma[rowB + t.x0] : 0.0f; sb1 = v1 ? a.sigma[rowB + t.x0 + 1] : 0.0f; }
  }
  const float wA = (NROWS == 1) ? 1.0f : r.wA, wB = (NROWS == 1) ? 0.0f : r.wB;
  const float a0 = t.w0 * wA, a1 = t.w1 * wA, b0 = t.w0 * wB, b1 = t.w1 * wB;
  const float l = la0 * a0 + la1 * a1 + lb0 * b0 + lb1 * b1;
  const float s = sa0 * a0 + sa1 * a1 + sb0 * b0 + sb1 * b1;
  const float dlx = (la1 - la0) * wA + (lb1 - lb0) * wB, dsx = (sa1 - sa0) * wA + (sb1 - sb0) * wB;
  const float4 ca = col_at<PK>(L, min(max(t.x0, -2), L.CW - 4) + 2), cb = col_at<PK>(L, min(max(t.x0 + 1, -2), L.CW - 4) + 2);
  const float c0 = ca.x * t.w0 + cb.x * t.w1, c1 = ca.y * t.w0 + cb.y * t.w1, c2 = ca.z * t.w0 + cb.z * t.w1;
  const PixelCtx c = ctx_at(L, xt + 2);
  const PlaneGrad pg = plane_grad<MIX>(c, l, s, c0, c1, c2);
  const float wy = (NROWS == 1) ? 1.0f : r.wy;
  float* gl = o.g_logits ? o.g_logits + ((long)r.b * a.N + n) * HW + (long)r.y * W : nullptr;
  float* gs = (MIX && o.g_sigma) ? o.g_sigma + ((long)r.b * a.N + n) * HW + (long)r.y * W : nullptr;
  float f0 = 1.0f, f1 = 1.0f;   // TAIL: the sigma gradient arrives in the conv output's space (tail_fs of the destination element)
  if (TAIL && MIX) {
    const bool ownA = (NROWS == 1) || r.yA == r.y;
    const float* rs = o.tail_raw_sigma + ((long)r.b * a.N + n) * HW + (long)r.y * W;
    if (v0) f0 = tail_fs(ownA ? sa0 : sb0, rs + t.x0);
    if (v1) f1 = tail_fs(ownA ? sa1 : sb1, rs + t.x0 + 1);
  }
  if (v0) {
    if (gl) unsafeAtomicAdd(gl + t.x0, pg.g_l * (t.w0 * wy));
    if (gs) unsafeAtomicAdd(gs + t.x0, pg.g_s * (t.w0 * wy) * f0);
  }
  if (v1) {
    if (gl) unsafeAtomicAdd(gl + t.x0 + 1, pg.g_l * (t.w1 * wy));
    if (gs) unsafeAtomicAdd(gs + t.x0 + 1, pg.g_s * (t.w1 * wy) * f1);
  }
  return pg.g_l * dlx + pg.g_s * dsx + pg.gc0 * (cb.x - ca.x) + pg.gc1 * (cb.y - ca.y) + pg.gc2 * (cb.z - ca.z);
}

// Stage target row r.y of image r.b for its workgroup: per-target-pixel context (cell = pixel + 2, zero-gradient guard cells),
// the (vertically blended) source colour row and, TAIL, the decoder tail's per-source-pixel terms.
template <bool MIX, int NROWS, bool PK, bool TAIL>
__device__ __forceinline__ void stream_stage_ctx(const SweepArgs& a, const BwdOut& o, const StreamRow& r, const StreamLds& L, int HW) {
  const int W = a.W;
  const float* srcb = a.src + (long)r.b * 3 * HW;
  for (int cidx = threadIdx.x; cidx < L.CW; cidx += blockDim.x) {
    const int x = cidx - 2;
    PixelCtx c = zero_pixel_ctx();
    c.lse2 = 3.0e38f;   // probability 0: a slot whose target is outside the row gets exact zeros
    float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 0 && x < W) {
      c = make_pixel_ctx<MIX>(a, o, r.b, r.y * W + x, HW);
      const float* p = srcb + (long)r.yA * W + x;
      cc = make_float4(p[0], p[HW], p[2 * HW], 0.0f);
      if (NROWS == 2) {
        const float* q = srcb + (long)r.yB * W + x;
        // fl(B*wB + fl(A*wA)): the rounding of the four-tap sum of the target-ordered kernels when the column weights are (1, 0)
        // — with tgt == src and a zero disparity, sign(c - t) sits on the last ulp (knife edge (d))
        cc = make_float4(fmaf(q[0], r.wB, cc.x * r.wA), fmaf(q[HW], r.wB, cc.y * r.wA), fmaf(q[2 * HW], r.wB, cc.z * r.wA), 0.0f);
      }
    }
    if (TAIL) {   // per SOURCE pixel of the row: what the decoder tail's backward needs (tail_term); zeros outside the row
      float4 tc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x >= 0 && x < W) {
        const long pix = (long)r.b * HW + (long)r.y * W + x;
        const float dsp = o.tail_disp[pix];
        float gD = o.tail_g_disp ? o.tail_g_disp[pix] : 0.0f;
        if (o.tail_g_depth) gD -= o.tail_g_depth[pix] * (0.1f * 0.58f * (float)W) / (dsp * dsp);   // depth = 0.1 * 0.58 * W / disp
        tc = make_float4(o.tail_stash[(long)r.b * 2 * HW + (long)r.y * W + x], 1.0f / o.tail_stash[((long)r.b * 2 + 1) * HW + (long)r.y * W + x], dsp, gD);
      }
      L.tail[cidx] = tc;
    }
    L.ctx0[cidx] = make_float4(c.t0, c.t1, c.t2, c.lse2);
    L.ctx1[cidx] = make_float4(c.gr0, c.gr1, c.gr2, c.gdotr);
    L.ctx2[cidx] = make_float4(c.invS, c.mx, c.A, PK ? cc.x : 0.0f);
    if (PK) L.colgb[cidx] = make_float2(cc.y, cc.z);
    else L.col[cidx] = cc;
  }
}

// One plane's staged shift (stream_body's form) + its live segment range (PD_STREAM_DEAD).  Returns "special".
template <bool TAIL>
__device__ __forceinline__ int stage_shift(const SweepArgs& a, const StreamLds& L, int i, float plane_i, bool masked, int nseg) {
  const int W = a.W;
  const float lim = (float)(W + 2), tol = irregular_tol(W);
  const float sdr = a.sign * plane_i;
  const float sd = (!masked && sdr >= -lim && sdr <= lim) ? sdr : ((sdr < 0.0f && !masked) ? -lim : lim);  // NaN -> +lim
  const float fl = floorf(sd), fr = sd - fl;
  const int k = (int)fl;
  const bool inview = fabsf(sd) < (float)(W + 1);
  const int irr = (inview && (fr < tol || fr > 1.0f - tol)) ? 1 : 0;
  L.shift[i] = make_int2(__float_as_int(sd), k * 2 + irr);
  if (TAIL) L.dpl[i] = plane_i;
  L.red[i] = 0.0f;
  // Slot xs serves the targets xs - k (left tap) and xs - 1 - k (right tap).  k >= 0: none of segment s is inside the row when
  // s * kSeg + kSeg - 1 < k; k < 0: when s * kSeg - 1 - k >= W.  Irregular planes keep every segment (their general path decides).
  int lo = 0, hi = nseg;
  if (!irr) {
    if (k >= 0) lo = min(k / kSeg, nseg);
    else hi = min(nseg, max(0, (W + 1 + k + kSeg - 1) / kSeg));
  }
  L.live[i] = lo | (hi << 16);
  return (irr || (k < 0 && inview)) ? 1 : 0;
}

template <bool MIX, int NROWS, bool PK, bool TAIL>
__device__ __forceinline__ void stream_body(const SweepArgs& a, const BwdOut& o, int b, int y, const RowSel& row, const StreamLds& L) {
  constexpr int D = (NROWS == 1) ? PD_STREAM_D1 : PD_STREAM_D2;
  const int W = a.W, N = a.N, HW = a.H * a.W;
  StreamRow r;
  r.y = y;
  r.b = b;
  r.yA = row.yA; r.yB = (NROWS == 2) ? row.yB : row.yA;
  r.wA = row.wA; r.wB = (NROWS == 2) ? row.wB : 0.0f;
  r.wy = row.wy_main;
  const int lane = threadIdx.x & (kWave - 1);
  const int nwaves = __builtin_amdgcn_readfirstlane(blockDim.x >> 6), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nseg = (W + kSeg - 1) / kSeg;
  const int want_plane = __builtin_amdgcn_readfirstlane(o.g_plane != nullptr ? 1 : 0);   // (as a template parameter: 83 VGPRs, 0.189 -> 0.200-0.205 ms)
  const int gl_bytes = __builtin_amdgcn_readfirstlane(o.g_logits ? W * 4 : 0);   // 0: the stores become no-ops
  const int gs_bytes = __builtin_amdgcn_readfirstlane(o.g_sigma ? W * 4 : 0);
  const float Wm1 = (float)(W - 1), rcpWm1 = refined_rcp(Wm1);

  // ---- this wave's contiguous range of the row's (plane, segment) list -------------------------------------------
  const int items = N * nseg;
  const int i0 = __builtin_amdgcn_readfirstlane((int)((long)items * wave / nwaves));
  const int i1 = __builtin_amdgcn_readfirstlane((int)((long)items * (wave + 1) / nwaves));
  const unsigned lane8 = (unsigned)lane * (kSlots * 4);
  const float lane2f = (float)(lane * kSlots);
  float carry_l = 0.0f, carry_s = 0.0f, gacc = 0.0f;
  int n = i0 / nseg, seg = i0 - n * nseg;            // the item being computed
  int pn = n, pseg = seg;                            // the item being prefetched
  constexpr bool kDead = PD_STREAM_DEAD && !TAIL;    // (TAIL: a dead item's stores still carry the tail's own terms, which need the taps)
  int p_live = nseg << 16;                           // live segment range of the plane being prefetched (lo | hi << 16): all, until staged
  auto advance = [&](int& nn, int& ss) {
    ++ss;
    if (ss == nseg) { ss = 0; ++nn; }
  };
  StreamGroup<NROWS> g[D + 1];
  auto prefetch = [&](StreamGroup<NROWS>& grp) {
    int ext = -1;
    if (kDead) ext = (pseg < (p_live & 0xFFFF) || pseg >= (p_live >> 16)) ? 0 : W;   // a dead item's loads: no-ops that return zeros
    stream_issue<MIX, NROWS>(grp, a, r, min(pn, N - 1), pseg, lane8, HW, ext);   // past the end: re-load the last plane (unused)
    advance(pn, pseg);
    if (kDead && pseg == 0) p_live = __builtin_amdgcn_readfirstlane(L.live[min(pn, N - 1)]);   // (once per plane)
  };

  // ---- stage the row: per-target-pixel context, blended colour row, per-plane shifts -----------------------------
  if (PD_STREAM_PRIO) __builtin_amdgcn_s_setprio(3);   // a workgroup that has just arrived gets through its dependent round trips ahead of the streaming ones
  int special;
  stream_stage_ctx<MIX, NROWS, PK, TAIL>(a, o, r, L, HW);
  if (threadIdx.x == 0) *L.special = 0;
  __syncthreads();
  {
    int flag = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      const long di = (a.flags & PD_DISP_ROWS) ? ((long)r.b * N + i) * a.H + r.y : (long)r.b * N + i;
      const bool masked = a.mask_rows && a.mask_rows[((long)r.b * N + i) * a.H + r.y] == 0.0f;
      flag |= stage_shift<TAIL>(a, L, i, a.plane[di], masked, nseg);
    }
    if (flag) *L.special = 1;
  }
  __syncthreads();
  special = __builtin_amdgcn_readfirstlane(*L.special);

  if (special) {   // irregular planes: their gradient rows are accumulated with atomics, so they start from zero
    for (int n2 = 0; n2 < N; ++n2) {
      if (!(L.shift[n2].y & 1)) continue;
      for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const long at = ((long)r.b * N + n2) * HW + (long)r.y * W + x;
        float il = 0.0f, is = 0.0f;
        if (TAIL) {   // the tail's own term goes in first; the atomics then add the sweep's shares (sigma's already scaled)
          const TailTerm tt = tail_term(L.tail[x + 2], a.logits[at], a.sigma[at], L.dpl[n2], o.tail_raw_sigma + at);
          il = tt.t; is = -tt.tos * tt.fs;
          if (want_plane) atomicAdd(&L.red[n2], a.sign * tt.gdl);
        }
        if (o.g_logits) o.g_logits[at] = il;
        if (MIX && o.g_sigma) o.g_sigma[at] = is;
      }
    }
    __syncthreads();
  }
  if (kDead) p_live = __builtin_amdgcn_readfirstlane(L.live[min(pn, N - 1)]);

  auto step = [&](const StreamGroup<NROWS>& grp) {
    const int2 sh = L.shift[n];
    const int lv = kDead ? __builtin_amdgcn_readfirstlane(L.live[n]) : 0;
    const float sd = __int_as_float(__builtin_amdgcn_readfirstlane(sh.x));
    const int kk = __builtin_amdgcn_readfirstlane(sh.y);
    const int k = kk >> 1;
    if (seg == 0) carry_l = carry_s = 0.0f;
    if (kDead && (seg < (lv & 0xFFFF) || seg >= (lv >> 16))) {   // dead item (wave-uniform): zeros to store, nothing to read
      const unsigned soff = (unsigned)seg * (kSeg * 4);
      buf_store2(row_rsrc_bytes(plane_ptr(o.g_logits + (long)r.b * a.N * HW + (long)r.y * a.W, n, HW), gl_bytes), lane8, soff, 0.0f, 0.0f);
      if (MIX) buf_store2(row_rsrc_bytes(plane_ptr(o.g_sigma + (long)r.b * a.N * HW + (long)r.y * a.W, n, HW), gs_bytes), lane8, soff, 0.0f, 0.0f);
      carry_l = carry_s = 0.0f;
    } else if (kk & 1) {   // irregular plane (wave-uniform branch): exact per-lane floor, atomics into the zero-filled row
#pragma unroll
      for (int i = 0; i < kSlots; ++i) {
        const float gd = stream_general_slot<MIX, NROWS, PK, TAIL>(a, o, r, L, n, seg * kSeg + lane * kSlots + i, k, sd, true, HW, Wm1, rcpWm1);
        if (want_plane) gacc += gd;
      }
      carry_l = carry_s = 0.0f;
    } else {
      stream_compute<MIX, NROWS, PK, TAIL>(grp, a, o, r, L, n, seg, k, sd, lane, lane8, lane2f, HW, Wm1, rcpWm1, want_plane, gl_bytes,
                                 gs_bytes, carry_l, carry_s, gacc);
    }
    advance(n, seg);
    if (want_plane && (seg == 0)) {   // the plane's row is complete for this wave
      const float v = wave_sum_hi(gacc);
      if (lane == kWave - 1) lds_add(&L.red[n - 1], v);
      gacc = 0.0f;
    }
  };
  if (i0 < i1) {
#pragma unroll
    for (int j = 0; j < D; ++j) prefetch(g[j]);
    if (PD_STREAM_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    int it = i0;
    for (; it + (D + 1) <= i1; it += D + 1) {
#pragma unroll
      for (int j = 0; j <= D; ++j) {
        prefetch(g[(j + D) % (D + 1)]);
        step(g[j]);
      }
      if (PD_STREAM_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int j = 0; j <= D; ++j) {
      if (it + j < i1) {
        prefetch(g[(j + D) % (D + 1)]);
        step(g[j]);
      }
    }
    if (seg != 0) {   // the range ends inside a row: hand the carry over, flush the partial disparity sum
      if (lane == 0) { L.hand[wave * 2] = carry_l; L.hand[wave * 2 + 1] = carry_s; }
      if (want_plane) {
        const float v = wave_sum_hi(gacc);
        if (lane == kWave - 1) lds_add(&L.red[n], v);
      }
    }
  }
  __syncthreads();
  // ---- epilogue ---------------------------------------------------------------------------------------------------
  // (1) range boundaries inside a row: the first slot of the next wave's first segment lacks the carry
  if (wave > 0 && i0 < i1 && lane == 0) {
    const int ns = i0 / nseg, ss = i0 - ns * nseg;
    if (ss != 0 && !(L.shift[ns].y & 1)) {
      const float hl = L.hand[(wave - 1) * 2], hs = L.hand[(wave - 1) * 2 + 1];
      const long at = ((long)r.b * N + ns) * HW + (long)r.y * W + ss * kSeg;
      if (o.g_logits && hl != 0.0f) unsafeAtomicAdd(o.g_logits + at, hl);
      if (MIX && o.g_sigma && hs != 0.0f) unsafeAtomicAdd(o.g_sigma + at, TAIL ? hs * tail_fs(a.sigma[at], o.tail_raw_sigma + at) : hs);
    }
  }
  // (2) targets without a slot: x0 = -1 (negative shifts; their right tap is column 0) and, on irregular planes, the
  //     neighbours of the row's ends whose floor(ix) lands inside after all.  One lane per (plane, virtual slot).
  if (special) {
    const int s_end = nseg * kSeg;
    for (int j = threadIdx.x; j < 3 * N; j += blockDim.x) {
      const int pn2 = j / 3, which = j - pn2 * 3;
      const int2 sh = L.shift[pn2];
      const int kk = sh.y, k = kk >> 1;
      const bool irr = kk & 1;
      const int xs = (which == 0) ? -1 : ((which == 1) ? -2 : s_end);
      const bool on = (which == 0) || irr;
      const float gd = stream_general_slot<MIX, NROWS, PK, TAIL>(a, o, r, L, pn2, xs, k, __int_as_float(sh.x), on, HW, Wm1, rcpWm1);
      if (want_plane && gd != 0.0f) atomicAdd(&L.red[pn2], gd);
    }
    __syncthreads();
  }
  if (PD_STREAM_PRIO) __builtin_amdgcn_s_setprio(0);
  if (want_plane) {
    const float gix_scale = (Wm1 / 2) * 2.0f / Wm1 * a.sign;  // d ix / d disp through un-normalise, *2, /(W-1)
    if (a.flags & PD_DISP_ROWS) {  // one disparity per (plane, row): this workgroup owns the whole sum
      for (int i = threadIdx.x; i < N; i += blockDim.x) o.g_plane[((long)r.b * N + i) * a.H + r.y] = L.red[i] * gix_scale;
    } else if (a.flags & PD_BWD_PLANE_ZEROED) {   // the caller's g_plane holds zeros: the rows add up there (no partials, no reduction launch)
      for (int i = threadIdx.x; i < N; i += blockDim.x) unsafeAtomicAdd(o.g_plane + (long)r.b * N + i, L.red[i] * gix_scale);
    } else {
      float* dstp = o.partials + ((long)r.b * a.H + r.y) * N;
      for (int i = threadIdx.x; i < N; i += blockDim.x) dstp[i] = L.red[i] * gix_scale;
    }
  }
}

#ifdef PD_EXPERIMENTS   // row pairs in the backward (round 6): measured slower; the kernel lives in scripts/experiments/
#include "pd_rowstream_pairs.inc"
#endif

// (TAIL: the tail's terms take the kernel from 77 to 93 VGPRs, which costs the third resident workgroup; bounded at six waves per
// SIMD it keeps it for twelve spilled dwords outside the loop: 0.175 against 0.196 ms, next to 0.174 + 0.208 ms for the two kernels)
template <bool MIX, bool PK, bool TAIL>
__global__ __launch_bounds__(kStreamThreadsMax, TAIL ? 6 : PD_STREAM_OCC) void rowstream_bwd_kernel(SweepArgs a, BwdOut o) {
  extern __shared__ float4 lds4[];
  StreamLds L;
  const int nseg = (a.W + kSeg - 1) / kSeg;
  L.CW = nseg * kSeg + 4;
  L.ctx0 = lds4; L.ctx1 = lds4 + L.CW; L.ctx2 = lds4 + 2 * L.CW; L.col = lds4 + 3 * L.CW;
  L.colgb = reinterpret_cast<float2*>(lds4 + 3 * L.CW);
  L.tail = PK ? nullptr : lds4 + 4 * L.CW;                         // (TAIL comes with the plain layout only)
  float4* after = TAIL ? lds4 + 5 * L.CW : lds4 + 4 * L.CW;
  L.shift = PK ? reinterpret_cast<int2*>(L.colgb + L.CW) : reinterpret_cast<int2*>(after);
  L.red = reinterpret_cast<float*>(L.shift + a.N);
  L.hand = L.red + a.N;
  L.special = reinterpret_cast<int*>(L.hand + 2 * (blockDim.x >> 6));
  L.dpl = reinterpret_cast<float*>(L.special + 4);
  L.live = reinterpret_cast<int*>(L.dpl + a.N);
  const int y = block_row(bwd_rowid(a.B, a.H), a.H), b = wg_image(a.B, a.H);
  const RowSel row = two_row_form(make_row_sel(y, a.H), a.row_eps);
  if (row.nrows == 2 && !(kStreamAbl & 4)) stream_body<MIX, 2, PK, TAIL>(a, o, b, y, row, L);
  else                                     stream_body<MIX, 1, PK, TAIL>(a, o, b, y, row, L);
}

__global__ void reduce_rows_stream_kernel(const float* __restrict__ partials, float* __restrict__ out, int R, int M) {
  const int j = blockIdx.x, b = blockIdx.y;   // partials [B][R][M] -> out [B][M]; lanes stride over R; deterministic
  const float* p = partials + (long)b * R * M + j;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < R; i += kWave) acc += p[(long)i * M];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * M + j] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// LDS of a row workgroup.  Plain layout: four float4 per cell (64 B; the colour's and ctx2's fourth floats unused); packed:
// the colour's r in ctx2's spare float and (g, b) as a float2 (56 B) — two more LDS reads per item, so it is used only where
// it buys a workgroup per CU (192 x 640: 42 KB, three workgroups either way; 384 x 1280: 83 KB = ONE workgroup plain,
// 72 KB = two packed).
static size_t rowstream_lds_bytes(const pd_sweep_desc* d, int nwaves, bool packed, bool tail = false) {
  const size_t CW = (size_t)ceil_div(d->W, kSeg) * kSeg + 4;
  return CW * (packed ? 3 * sizeof(float4) + sizeof(float2) : (tail ? 5 : 4) * sizeof(float4)) +
         (size_t)d->N * (sizeof(float2) + 3 * sizeof(float)) + (size_t)nwaves * 2 * sizeof(float) + 32;
}
struct StreamShape { int nwaves; bool packed; size_t lds; };
static StreamShape rowstream_shape(const pd_sweep_desc* d, bool tail = false) {
  const size_t kCuLds = device_lds_bytes();
  // workgroups per CU by LDS (at most three: 24 waves per CU at the kernel's 77 VGPRs); waves per workgroup to fill them:
  // three workgroups of 8, two of 12, one of 16
  const int wg_plain = (int)(kCuLds / rowstream_lds_bytes(d, 2 * PD_STREAM_WAVES, false, tail));
  const int wg_packed = tail ? 0 : (int)(kCuLds / rowstream_lds_bytes(d, 2 * PD_STREAM_WAVES, true));
  StreamShape s;
  s.packed = wg_plain < 3 && wg_packed > wg_plain;
  const int wg = s.packed ? wg_packed : wg_plain;
  const int w = wg >= 3 ? PD_STREAM_WAVES : wg == 2 ? (3 * PD_STREAM_WAVES) / 2 : 2 * PD_STREAM_WAVES;
  const int items = d->N * ceil_div(d->W, kSeg);
  s.nwaves = items < w ? items : w;
  s.lds = rowstream_lds_bytes(d, s.nwaves, s.packed, tail);
  return s;
}

bool rowstream_bwd_applicable(const pd_sweep_desc* d, const SweepArgs& a) {
  return rowshift_applicable(d) && !(d->flags & PD_RENDER_PROB) && !a.has_mask && (d->W % 2 == 0) &&
         rowstream_shape(d).lds <= device_lds_bytes();   // else: the row-shift backward, which needs less
}

size_t rowstream_bwd_workspace_floats(const pd_sweep_desc* d) { return (size_t)d->B * d->H * d->N; }

template <bool MIX, bool PK, bool TAIL>
static int rowstream_launch(const SweepArgs& a, const BwdOut& o, dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  static LdsGrant granted;   // per instantiation and device: the attribute is set once (and checked), not per launch
  const int rc = grant_dynamic_lds((const void*)rowstream_bwd_kernel<MIX, PK, TAIL>, shmem, &granted, "rowstream_bwd_kernel");
  if (rc) return rc;
  rowstream_bwd_kernel<MIX, PK, TAIL><<<grid, block, shmem, stream>>>(a, o);
  return PD_OK;
}

// The fused decoder tail rides along (pd_plane_sweep_bwd_tail) where the plain LDS layout with one more float4 per cell fits:
// mixture, one disparity per plane, unit sign.
bool rowstream_bwd_tail_applicable(const pd_sweep_desc* d, const SweepArgs& a) {
  return rowstream_bwd_applicable(d, a) && (d->flags & PD_MIXTURE) && !(d->flags & (PD_DISP_ROWS | PD_MASK_ROWS)) &&
         (d->sign == 1.0f || d->sign == -1.0f) && rowstream_shape(d, true).lds <= device_lds_bytes();
}

#ifdef PD_EXPERIMENTS
#include "pd_rowstream_pairs_host.inc"
#else
static bool rowstream_pairs(const pd_sweep_desc*, const SweepArgs&, bool) { return false; }
#endif

int rowstream_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, hipStream_t stream) {
  const bool tail = o.tail_stash != nullptr;
  const bool mix = (d->flags & PD_MIXTURE) != 0;
  int rc;
#ifdef PD_EXPERIMENTS
  static const bool pairs_on = getenv("PD_BWD_PAIRS") != nullptr;
#else
  constexpr bool pairs_on = false;
#endif
  if (pairs_on && rowstream_pairs(d, a, tail)) {
#ifdef PD_EXPERIMENTS
    const int items = d->N * ceil_div(d->W, kSeg);
    const int nwaves = items < PD_STREAM_PAIR_WAVES ? items : PD_STREAM_PAIR_WAVES;
    const dim3 block(nwaves * kWave);
    const size_t lds = rowstream_pair_lds_bytes(d, nwaves);
    const StreamUnits& units = rowstream_units(d->H);
    rc = mix ? rowstream_pair_launch<true>(a, o, units, d->B, block, lds, stream)
             : rowstream_pair_launch<false>(a, o, units, d->B, block, lds, stream);
#else
    rc = PD_ERR_UNSUPPORTED;
#endif
  } else {
    const StreamShape sh = rowstream_shape(d, tail);
    dim3 grid(d->H, d->B), block(sh.nwaves * kWave);
    if (tail)     rc = rowstream_launch<true, false, true>(a, o, grid, block, sh.lds, stream);
    else if (mix) rc = sh.packed ? rowstream_launch<true, true, false>(a, o, grid, block, sh.lds, stream)
                                 : rowstream_launch<true, false, false>(a, o, grid, block, sh.lds, stream);
    else          rc = sh.packed ? rowstream_launch<false, true, false>(a, o, grid, block, sh.lds, stream)
                                 : rowstream_launch<false, false, false>(a, o, grid, block, sh.lds, stream);
  }
  if (rc) return rc;
  rc = check_launch("rowstream_bwd_kernel");
  if (rc || !o.g_plane || (d->flags & (PD_DISP_ROWS | PD_BWD_PLANE_ZEROED))) return rc;
  reduce_rows_stream_kernel<<<dim3(d->N, d->B), kWave, 0, stream>>>(o.partials, o.g_plane, d->H, d->N);
  return check_launch("reduce_rows_kernel");
}

}  // namespace pd
