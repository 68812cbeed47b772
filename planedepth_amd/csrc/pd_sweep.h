// Shared definitions of the plane-sweep kernels: argument blocks, per-pixel / per-plane math.
// The math follows reference trainer.py:580-603 (mask, softmax over planes, sigma clamp, mixture weights, composite)
// and trainer.py:728-742 + layers.py:454-466 (Laplacian-mixture NLL, automask min); the closed-form backward is
// derived in DESIGN.md §"Backward".
#pragma once
#include "pd_common.h"

namespace pd {

// Diagnostics only: -DPD_ABLATE=<bits> builds a library that skips parts of the row-shift kernels (scripts/gpu_ablate.sh).
#ifndef PD_ABLATE
#define PD_ABLATE 0
#endif
#ifndef PD_DIAGNOSTICS   // timing-ablation / trace code (results wrong by design) compiles only into a library that says so: pd_build_flags()
#if PD_ABLATE
#error "timing-ablation / trace switches need -DPD_DIAGNOSTICS as well (pd_build_flags() then reports the build)"
#endif
#endif
constexpr int kAblate = PD_ABLATE;

constexpr int kStashBase = 4;  // lse, S, Mx, flags  (then ceil(N/32) mask words in disp mode)
constexpr float kFastRowWeight = 1.52587890625e-05f;  // 2^-16: the threshold of PD_IMPL_FAST_ROWS (pd_rowshift_common.h: two_row_form)
constexpr float kSigmaMin = 0.01f, kSigmaMax = 1.0f, kLogEps = 1e-7f, kZMin = 1e-7f;

struct SweepArgs {
  int B, N, H, W;
  int flags;
  float sign;
  int stash_k;   // floats per pixel in the stash: kStashBase + ceil(N/32) mask words in disp mode
  int has_mask;  // disp mode with a padding_mask tensor: mask bits live in the stash words
  const float* src;
  const float* tgt;
  const float* logits;
  const float* sigma;
  const float* plane;
  const float* plane_aux;
  const float* inv_K3;
  const float* padding_mask;
  int pairs;               // forward: inexact rows take their neighbour along (per-plane disparities, no mask)
  int fast_rows;           // row_eps > 0
  float row_eps;           // the row kernels drop a second source row whose weight is below this (pd_rowshift_common.h: two_row_form)
  const float* mask_rows;  // PD_MASK_ROWS: [B,N,H] (row-shift kernels only; padding_mask is NULL then)
  const float* dists;  // PD_RENDER_PROB: [B,N-1,H,W] inter-plane distances at the TARGET pixel (trainer.py:587)
  float* ph_mean;      // forward, optional: one float that receives mean(ph_map) (block sums, one atomic per wave)
  float inv_numel;     // 1 / (B*H*W)
};

struct BwdOut {
  float* g_logits;
  float* g_sigma;
  float* g_plane;   // dense disp: written directly; otherwise via partials
  float* g_dists;   // PD_RENDER_PROB only (may be NULL)
  float* partials;  // per-block partial sums of the plane-parameter gradient
  float* side;      // row-shift backward: global spill of the rare out-of-segment records (see route())
  float* scratch;   // gather backward, pass 1: per-pixel, per-plane (g_l, g_s) [B][N][H*W] (float2 with PD_MIXTURE)
  const float* rgb_rec;
  const float* stash;
  const float* g_rgb_rec;
  const float* g_ph_map;
  const float* g_ph_mean;  // optional: device scalar, upstream gradient of mean(ph_map)
  // pd_plane_sweep_bwd_tail (row-stream backward only): the fused decoder tail's backward rides along — g_logits / g_sigma
  // receive the gradients of the decoder's CONV outputs (networks/depth_decoder.py:258-291 through pd_decoder_tail_fwd)
  const float* tail_raw_sigma = nullptr;  // [B,N,H,W] sigmaconv output (read only where sigma sits on the lower clamp bound)
  const float* tail_stash = nullptr;      // [B,2,H,W] pd_decoder_tail_fwd's stash: log-sum-exp of the logits, sum pi/sigma
  const float* tail_disp = nullptr;       // [B,1,H,W]
  const float* tail_g_disp = nullptr;     // upstream gradients of the tail's disp / depth outputs (either may be NULL)
  const float* tail_g_depth = nullptr;
};

// ---- forward: online softmax / mixture accumulators of ONE target pixel over the planes ---------------------------
// Everything exponential is evaluated in base 2 (v_exp_f32 / v_log_f32 are base-2 instructions): logits are scaled by
// log2(e) once, the stash keeps the log-sum-exp in log2 units.  The running reference `m` of the online softmax is only
// moved when a logit exceeds it by more than kRescaleThr (2^20): exact max-tracking would cost a divergent branch
// with a rescale of eight accumulators on ~ln(N) planes per pixel, any lane of the wave triggering it.
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
constexpr float kRescaleThr = 20.0f;

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_fast(float x) { return __builtin_amdgcn_logf(x); }

struct FwdAcc {
  float m = -INFINITY, Z = 0.0f, S = 0.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, Mx = 0.0f, Ma = 0.0f;
};

// One plane's (masked) samples l, s, c* enter the running sums.  t* = target colour, ea3 = 3 x the identity-
// reprojection error (sum over channels of |src - tgt|).  Mx / Ma omit the Laplacian's constant 1/2 (applied at the end).
// Unnormalised softmax weight of a plane (and the lazy rescale of everything accumulated so far).
__device__ __forceinline__ float softmax_weight(FwdAcc& a, float l) {
  const float l2 = l * kLog2e;
  if (l2 - a.m > kRescaleThr) {  // first plane (m = -inf) and rare large jumps only
    const float sc = exp2_fast(a.m - l2);
    a.Z *= sc; a.S *= sc; a.C0 *= sc; a.C1 *= sc; a.C2 *= sc; a.Mx *= sc; a.Ma *= sc;
    a.m = l2;
  }
  return exp2_fast(l2 - a.m);
}

// Accumulate one plane with (unnormalised) probability p.
template <bool MIX>
__device__ __forceinline__ void mixture_accumulate(FwdAcc& a, float p, float s, float c0, float c1, float c2, float t0,
                                                   float t1, float t2, float ea3, bool automask) {
  a.Z += p;
  if (MIX) {
    const float sg = fminf(fmaxf(s, kSigmaMin), kSigmaMax);  // trainer.py:597
    const float inv = fast_rcp(sg);
    const float u = p * inv;                                 // pi / sigma (trainer.py:600)
    a.S += u;
    a.C0 += c0 * u; a.C1 += c1 * u; a.C2 += c2 * u;
    const float e3 = fabsf(c0 - t0) + fabsf(c1 - t1) + fabsf(c2 - t2);  // 3 * mean_c |c - t|  (trainer.py:729)
    const float k = inv * (-kLog2e / 3.0f);
    a.Mx += u * exp2_fast(e3 * k);                                      // pi * 2 * laplacian(e; sigma)
    if (automask) a.Ma += u * exp2_fast(ea3 * k);
  } else {
    a.C0 += c0 * p; a.C1 += c1 * p; a.C2 += c2 * p;
  }
}

template <bool MIX>
__device__ __forceinline__ void fwd_accumulate(FwdAcc& a, float l, float s, float c0, float c1, float c2, float t0,
                                               float t1, float t2, float ea3, bool automask) {
  const float p = softmax_weight(a, l);
  mixture_accumulate<MIX>(a, p, s, c0, c1, c2, t0, t1, t2, ea3, automask);
}

// --render_probability (trainer.py:584-591): alpha compositing front to back instead of a softmax.
//   alpha_n = 1 - exp(-relu(l_n) * dists_n) (n < N-1), alpha_{N-1} = 1;  prob_n = alpha_n * prod_{m<n}(1 - alpha_m + 1e-10)
struct RenderState {
  float T = 1.0f;  // transmittance in front of the current plane
};
__device__ __forceinline__ float render_alpha(float l, float dist, bool last) {
  return last ? 1.0f : 1.0f - exp2_fast(-kLog2e * fmaxf(l, 0.0f) * dist);
}
__device__ __forceinline__ float render_prob(RenderState& r, float alpha) {
  const float p = alpha * r.T;
  r.T *= (1.0f - alpha + 1e-10f);
  return p;
}

struct FwdResult {
  float r0, r1, r2, ph;
  float lse2, Sn, mx, sel;  // stash: log2-sum-exp2 of the scaled logits, sum(pi/sigma), sum(pi*lap), automask flag
};

template <bool MIX>
__device__ __forceinline__ FwdResult fwd_finish(const FwdAcc& a, float t0, float t1, float t2, float ea3,
                                                bool automask, bool normalise = true) {
  FwdResult r;
  const float invZ = normalise ? 1.0f / a.Z : 1.0f;  // render_probability: the weights are used as they are
  r.sel = 0.0f;
  if (MIX) {
    const float invS = 1.0f / a.S;
    r.r0 = a.C0 * invS; r.r1 = a.C1 * invS; r.r2 = a.C2 * invS;
    r.mx = 0.5f * a.Mx * invZ;
    r.Sn = a.S * invZ;
    r.ph = -kLn2 * log2_fast(r.mx + kLogEps);  // layers.py:466
    if (automask) {
      const float pa = -kLn2 * log2_fast(0.5f * a.Ma * invZ + kLogEps);
      if (pa < r.ph) { r.ph = pa; r.sel = 1.0f; }  // torch.min over cat([ph, ph_auto]) keeps the first on ties
    }
  } else {
    r.r0 = a.C0 * invZ; r.r1 = a.C1 * invZ; r.r2 = a.C2 * invZ;
    r.mx = 0.0f;
    r.Sn = 0.0f;
    r.ph = (fabsf(r.r0 - t0) + fabsf(r.r1 - t1) + fabsf(r.r2 - t2)) / 3.0f;  // trainer.py:738
    const float ea = ea3 / 3.0f;
    if (automask && ea < r.ph) { r.ph = ea; r.sel = 1.0f; }
  }
  r.lse2 = normalise ? a.m + log2_fast(a.Z) : 0.0f;
  return r;
}

// ---- backward: per-target-pixel context and per-plane gradients w.r.t. the SAMPLED features -----------------------
struct PixelCtx {
  float t0, t1, t2;     // target colour
  float lse2;           // log2-sum-exp2 of the sampled logits scaled by log2(e)
  float invS, mx, A;    // mixture: 1 / sum(pi/sigma), sum(pi*lap), g_ph / (mx + 1e-7)
  float gr0, gr1, gr2;  // upstream gradient of rgb_rec (+ the L1 photometric term when !MIX)
  float gdotr;          // gr . rgb_rec
};

template <bool MIX>
__device__ __forceinline__ PixelCtx make_pixel_ctx(const SweepArgs& a, const BwdOut& o, int b, int pix, int HW) {
  PixelCtx c;
  c.t0 = a.tgt[((long)b * 3 + 0) * HW + pix];
  c.t1 = a.tgt[((long)b * 3 + 1) * HW + pix];
  c.t2 = a.tgt[((long)b * 3 + 2) * HW + pix];
  const float* st = o.stash + (long)b * a.stash_k * HW + pix;
  c.lse2 = st[0];
  const float Sn = st[HW];
  c.mx = st[2 * HW];
  const float sel = st[3 * HW];
  float gp = 0.0f;  // d loss / d ph_map at this pixel: per-pixel upstream gradient + the fused mean's share
  if (sel == 0.0f) {
    if (o.g_ph_map) gp = o.g_ph_map[(long)b * HW + pix];
    if (o.g_ph_mean) gp += o.g_ph_mean[0] * a.inv_numel;
  }
  const float r0 = o.rgb_rec[((long)b * 3 + 0) * HW + pix];
  const float r1 = o.rgb_rec[((long)b * 3 + 1) * HW + pix];
  const float r2 = o.rgb_rec[((long)b * 3 + 2) * HW + pix];
  c.gr0 = c.gr1 = c.gr2 = 0.0f;
  if (o.g_rgb_rec) {
    c.gr0 = o.g_rgb_rec[((long)b * 3 + 0) * HW + pix];
    c.gr1 = o.g_rgb_rec[((long)b * 3 + 1) * HW + pix];
    c.gr2 = o.g_rgb_rec[((long)b * 3 + 2) * HW + pix];
  }
  if (!MIX) {  // L1 branch: ph = mean_c |rgb_rec - tgt| feeds straight into the rgb_rec gradient
    c.gr0 += gp * sgn(r0 - c.t0) * (1.0f / 3.0f);
    c.gr1 += gp * sgn(r1 - c.t1) * (1.0f / 3.0f);
    c.gr2 += gp * sgn(r2 - c.t2) * (1.0f / 3.0f);
  }
  c.A = MIX ? gp / (c.mx + kLogEps) : 0.0f;  // -d ph / d Mx (layers.py:466)
  c.invS = MIX ? 1.0f / Sn : 1.0f;
  c.gdotr = c.gr0 * r0 + c.gr1 * r1 + c.gr2 * r2;
  return c;
}

__device__ __forceinline__ PixelCtx zero_pixel_ctx() {
  PixelCtx c;
  c.t0 = c.t1 = c.t2 = c.lse2 = 0.0f;
  c.invS = 1.0f; c.mx = 1.0f; c.A = 0.0f;
  c.gr0 = c.gr1 = c.gr2 = c.gdotr = 0.0f;
  return c;
}

struct PlaneGrad {
  float g_l, g_s, gc0, gc1, gc2;  // d loss / d sampled (logit, sigma, r, g, b) of this plane at this target pixel
};

// Gradients given the plane's probability p (softmax: pi_n; render: alpha_n T_n).  g.g_l holds d loss / d p here;
// the caller turns it into the logit gradient.
template <bool MIX>
__device__ __forceinline__ PlaneGrad plane_grad_p(const PixelCtx& c, float p, float s, float c0, float c1, float c2) {
  PlaneGrad g;
  if (MIX) {
    const float sg = fminf(fmaxf(s, kSigmaMin), kSigmaMax);
    const float inv = fast_rcp(sg);
    const float u = p * inv;
    const float e3 = fabsf(c0 - c.t0) + fabsf(c1 - c.t1) + fabsf(c2 - c.t2);          // 3 e
    const float ei = e3 * inv * (1.0f / 3.0f);                                        // e / sigma
    const float q = 0.5f * exp2_fast(-kLog2e * ei) * inv;                             // laplacian(e; sigma)
    const float gu = (c.gr0 * c0 + c.gr1 * c1 + c.gr2 * c2 - c.gdotr) * c.invS;       // d (g . rgb_rec) / d u_n
    g.g_l = -c.A * q + gu * inv;                                                      // d loss / d p
    const float g_sig = -c.A * p * q * (ei * inv - inv) - gu * u * inv;               // d / d sigma_n
    g.g_s = (s == sg) ? g_sig : 0.0f;  // clamp passes the gradient exactly where it left s untouched ([min,max])
    const float g_e3 = c.A * u * q * (1.0f / 3.0f);                                   // d ph / d e_n, per channel
    const float w = u * c.invS;
    g.gc0 = c.gr0 * w + g_e3 * sgn_fast(c0 - c.t0);
    g.gc1 = c.gr1 * w + g_e3 * sgn_fast(c1 - c.t1);
    g.gc2 = c.gr2 * w + g_e3 * sgn_fast(c2 - c.t2);
  } else {
    g.g_l = c.gr0 * c0 + c.gr1 * c1 + c.gr2 * c2;
    g.g_s = 0.0f;
    g.gc0 = c.gr0 * p; g.gc1 = c.gr1 * p; g.gc2 = c.gr2 * p;
  }
  return g;
}

// Softmax over planes: sum_k pi_k (d loss / d pi_k) = -A*mx (mixture) or g.rgb_rec (L1), so the softmax backward is
// g_l = pi_n (g_pi_n - that constant) in one pass.
template <bool MIX>
__device__ __forceinline__ PlaneGrad plane_grad(const PixelCtx& c, float l, float s, float c0, float c1, float c2) {
  const float p = exp2_fast(l * kLog2e - c.lse2);  // pi_n
  PlaneGrad g = plane_grad_p<MIX>(c, p, s, c0, c1, c2);
  g.g_l = p * (g.g_l - (MIX ? -c.A * c.mx : c.gdotr));
  return g;
}

// Row-shift specialisation (pd_plane_sweep_rowshift.hip): disp mode, per-plane scalar disparities.
bool rowshift_applicable(const pd_sweep_desc* d);
int rowshift_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash,
                 hipStream_t stream);
int rowshift_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, hipStream_t stream);
size_t rowshift_bwd_workspace_floats(const pd_sweep_desc* d);

// Segment-stream forward (pd_plane_sweep_fwdstream.hip): one wave per 128-pixel segment of a target row, two pixels per
// lane, one plane per iteration behind a deep register ring of 12-byte tap loads.
bool fwdstream_applicable(const pd_sweep_desc* d, const SweepArgs& a);
int fwdstream_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash, hipStream_t stream);

// Row-stream backward (pd_plane_sweep_rowstream.hip): lanes own aligned source slots, waves stream along plane rows.
bool rowstream_bwd_applicable(const pd_sweep_desc* d, const SweepArgs& a);
bool rowstream_bwd_tail_applicable(const pd_sweep_desc* d, const SweepArgs& a);
int rowstream_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, hipStream_t stream);
size_t rowstream_bwd_workspace_floats(const pd_sweep_desc* d);

#ifdef PD_EXPERIMENTS
// Four-pixels-per-lane row kernels (pd_plane_sweep_rowquad.hip): same contract as the row-shift ones, wide memory accesses.
bool rowquad_applicable(const pd_sweep_desc* d, bool dense_mask);
int rowquad_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash,
                hipStream_t stream);
int rowquad_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, hipStream_t stream);
size_t rowquad_bwd_workspace_floats(const pd_sweep_desc* d);

// Scatter-free general backward (pd_plane_sweep_tile.hip): homography mode, source tiles owned by workgroups.
bool tile_bwd_applicable(const pd_sweep_desc* d);
size_t tile_bwd_workspace_floats(const pd_sweep_desc* d);
int tile_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, float* workspace, hipStream_t stream);
#endif  // PD_EXPERIMENTS
// Plane-uniform homography (pd_plane_sweep_uniform.hip, PD_HOMO_UNIFORM)
int uniform_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash, hipStream_t stream);
int uniform_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, float* workspace, hipStream_t stream);
int uniform_fwd_pair(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash,
                     const SweepArgs& b, float* rgb_rec_b, float* ph_map_b, float* stash_b, hipStream_t stream);
int uniform_bwd_pair(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& oa, float* workspace_a, const SweepArgs& b,
                     const BwdOut& ob, float* workspace_b, float* g_logits, float* g_sigma, hipStream_t stream);

// Per-plane homographies (6-DoF poses) without atomics (pd_plane_sweep_gather.hip): pass 1 is sweep_bwd_kernel<.., true>
// (pd_plane_sweep.hip, launched by the caller between gather_bwd_prepare and gather_bwd_finish).
bool gather_bwd_applicable(const pd_sweep_desc* d);
size_t gather_bwd_workspace_floats(const pd_sweep_desc* d);
struct GatherPlan { float* partials; float* scratch; void* prep; int* flags; int nblk; };
GatherPlan gather_bwd_plan(const pd_sweep_desc* d, float* workspace);
int gather_bwd_prepare(const pd_sweep_desc* d, const SweepArgs& a, const GatherPlan& gp, hipStream_t stream);
int gather_bwd_finish(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, const GatherPlan& gp, hipStream_t stream);
size_t uniform_bwd_workspace_floats(const pd_sweep_desc* d);
int uniform_gather_pair(const pd_sweep_desc* d, const float* plane_a, const float* inv_K3_a, float* workspace_a,
                        const float* plane_b, const float* inv_K3_b, float* workspace_b, float* g_logits, float* g_sigma,
                        hipStream_t stream);
// partials [B][nblk][M] -> out [B][M], fixed summation order (pd_plane_sweep.hip)
int reduce_partials(const float* partials, float* out, int nblk, int M, int B, hipStream_t stream);

}  // namespace pd
