// Fused orthogonal-plane sweep: grid generation -> bilinear warp of (rgb, logit, sigma) -> mask -> softmax over
// planes -> mixture weights -> composite + photometric loss, forward and backward, for gfx950.
//
// Replaces Trainer.pred_novel_images (reference trainer.py:523-603) and the photometric part of
// Trainer.compute_losses (trainer.py:717-742, layers.py:454-466).  The reference materialises
// [B,N,5,H,W] intermediates (~1 GB / image); here one thread owns one target pixel and STREAMS over the N
// planes with an online softmax, so HBM sees logits+sigma once per pass and nothing else of size N*HW.
//
// Kernels in this file (general path; the row-shift specialisation of the backward lives in
// pd_plane_sweep_rowshift.hip):
//   sweep_fwd_kernel      one pass over N: rgb_rec, ph_map, stash (lse, S, Mx, automask flag, mask bits)
//   sweep_bwd_kernel      one pass over N: re-sample, per-plane gradients, atomic scatter to g_logits/g_sigma,
//                         block-reduced plane-parameter gradient partials
//   reduce_partials_kernel  deterministic second stage for the plane-parameter gradient
//   sweep_layers_kernel   optional materialisation of the per-plane tensors the reference stores in `outputs`
#include <stdlib.h>

#include "pd_sweep.h"
#include "pd_sweep_geom.h"

namespace pd {

// ---------------------------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool MIX>
__global__ __launch_bounds__(kBlock) void sweep_fwd_kernel(SweepArgs a, float* __restrict__ rgb_rec,
                                                           float* __restrict__ ph_map, float* __restrict__ stash) {
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  const int b = blockIdx.y;
  float ph_val = 0.0f;  // this pixel's ph (0 for lanes past the image), summed per wave for the fused mean
  if (pix < HW) {
  const int y = pix / a.W, x = pix - y * a.W;
  const bool automask = a.flags & PD_AUTOMASK;
  const bool has_mask = (MODE == PD_WARP_DISP) && a.has_mask;

  const float* srcb = a.src + (long)b * 3 * HW;
  const float t0 = a.tgt[((long)b * 3 + 0) * HW + pix];
  const float t1 = a.tgt[((long)b * 3 + 1) * HW + pix];
  const float t2 = a.tgt[((long)b * 3 + 2) * HW + pix];
  float ea = 0.0f;  // 3 x identity-reprojection error: sum_c |src - tgt| (trainer.py:732 / 740)
  if (automask) ea = fabsf(srcb[pix] - t0) + fabsf(srcb[HW + pix] - t1) + fabsf(srcb[2 * HW + pix] - t2);
  const float iy_disp = (MODE == PD_WARP_DISP) ? normalise_roundtrip((float)y, (float)(a.H - 1)) : 0.0f;
  const CoordNorm cn = make_coord_norm(a.W, a.H);

  const bool render = a.flags & PD_RENDER_PROB;
  FwdAcc acc;
  RenderState rs;
  uint32_t bits = 0;
  for (int n = 0; n < a.N; ++n) {
    bool mk;
    const PlaneGeom g = plane_coords<MODE>(a, cn, b, n, x, y, iy_disp, mk);
    if (has_mask) {
      mk = read_mask(a, b, n, x, y);
      if (mk) bits |= 1u << (n & 31);
      if ((n & 31) == 31 || n == a.N - 1) {
        stash[((long)b * a.stash_k + kStashBase + (n >> 5)) * HW + pix] = __uint_as_float(bits);
        bits = 0;
      }
    }
    float l = 0.0f, s = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    if (mk) {  // rec_features * padding_mask (trainer.py:580): a masked plane samples as all-zero features
      const TapK t = tap_kernel(make_tap(g.ix, g.iy, a.W, a.H), a.W, a.H);
      const long pl = ((long)b * a.N + n) * HW;
      l = sample_k(a.logits + pl, t);
      if (MIX) s = sample_k(a.sigma + pl, t);
      c0 = sample_k(srcb, t);
      c1 = sample_k(srcb + HW, t);
      c2 = sample_k(srcb + 2 * HW, t);
    }
    if (render) {
      const bool last = (n == a.N - 1);
      const float dist = last ? 0.0f : a.dists[((long)b * (a.N - 1) + n) * HW + pix];
      const float p = render_prob(rs, render_alpha(l, dist, last));
      mixture_accumulate<MIX>(acc, p, s, c0, c1, c2, t0, t1, t2, ea, automask);
    } else {
      fwd_accumulate<MIX>(acc, l, s, c0, c1, c2, t0, t1, t2, ea, automask);
    }
  }
  const FwdResult r = fwd_finish<MIX>(acc, t0, t1, t2, ea, automask, !render);
  float* st = stash + (long)b * a.stash_k * HW + pix;
  st[0] = r.lse2;
  st[HW] = r.Sn;
  st[2 * HW] = r.mx;
  st[3 * HW] = r.sel;
  rgb_rec[((long)b * 3 + 0) * HW + pix] = r.r0;
  rgb_rec[((long)b * 3 + 1) * HW + pix] = r.r1;
  rgb_rec[((long)b * 3 + 2) * HW + pix] = r.r2;
  ph_map[(long)b * HW + pix] = r.ph;
  ph_val = r.ph;
  }
  if (a.ph_mean) {  // fused `.mean()` of trainer.py:742 (all threads get here): one atomic per workgroup
    __shared__ float wsum[kBlock / kWave];
    const float v = wave_sum(ph_val);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
      for (int w = 0; w < kBlock / kWave; ++w) t += wsum[w];
      unsafeAtomicAdd(a.ph_mean, t * a.inv_numel);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward (general: any warp, atomics for the bilinear adjoint)
// ---------------------------------------------------------------------------------------------------------------
// TOSCRATCH (the gather backward's pass 1, pd_plane_sweep_gather.hip): instead of scattering, every pixel writes its
// per-plane (g_l, g_s) side by side to o.scratch [B][N][H*W] (zeros for masked planes); no ghost lane then.
template <int MODE, bool MIX, bool TOSCRATCH = false>
__global__ __launch_bounds__(kBlock) void sweep_bwd_kernel(SweepArgs a, BwdOut o) {
  extern __shared__ float red[];  // [N*K] block accumulators of the plane-parameter gradient
  constexpr int K = (MODE == PD_WARP_DISP) ? 1 : 9;
  const int HW = a.H * a.W;
  // A wave covers 63 consecutive pixels; its lane 0 is a GHOST that repeats the previous wave's last pixel with all its
  // gradients forced to zero.  The wave-cooperative scatter hands every lane's left taps to the previous lane's right
  // taps — lane 1's now go to the ghost, so a coherent wave issues no second, nearly empty atomic instruction per row
  // for its first pixel's left taps.  That instruction cost as much as a full one: the L2's atomic unit is busy per
  // (instruction, cache line), not per lane (measured: dropping the left-over atomics halved the scatter time).
  const int lane_id = threadIdx.x & (kWave - 1);
  const bool ghost = !TOSCRATCH && (lane_id == 0);
  const int pix = TOSCRATCH ? (int)(blockIdx.x * kBlock + threadIdx.x)
                            : (int)((blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * (kWave - 1)) + lane_id - 1;
  const int b = blockIdx.y;
  const bool want_plane = (o.g_plane != nullptr);
  const bool dense = (MODE == PD_WARP_DISP) && (a.flags & PD_DISP_DENSE);
  const bool reduce_plane = want_plane && !dense;
  if (reduce_plane) {
    for (int i = threadIdx.x; i < a.N * K; i += kBlock) red[i] = 0.0f;
    __syncthreads();
  }
  const bool active = pix >= 0 && pix < HW;
  const int y = active ? pix / a.W : 0, x = active ? pix - y * a.W : 0;
  const bool has_mask = (MODE == PD_WARP_DISP) && a.has_mask;
  const float* srcb = a.src + (long)b * 3 * HW;
  const int SK = a.stash_k;

  const PixelCtx c = active ? make_pixel_ctx<MIX>(a, o, b, pix, HW) : zero_pixel_ctx();
  const float iy_disp = (MODE == PD_WARP_DISP) ? normalise_roundtrip((float)y, (float)(a.H - 1)) : 0.0f;
  const CoordNorm cn = make_coord_norm(a.W, a.H);
  const float gscale_x = (float)(a.W - 1) / 2 * 2.0f / (float)(a.W - 1), gscale_y = (float)(a.H - 1) / 2 * 2.0f / (float)(a.H - 1);

  const bool render = a.flags & PD_RENDER_PROB;
  const float Rtot = MIX ? -c.A * c.mx : c.gdotr;  // sum_k p_k dL/dp_k, known in closed form
  float T = 1.0f, prefix = 0.0f;
  uint32_t bits = 0;
  for (int n = 0; n < a.N; ++n) {
    float gk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) gk[k] = 0.0f;
    // what this lane scatters for plane n (filled in below; the scatter itself runs wave-wide after the branches)
    Tap st;
    st.x0 = st.y0 = 0; st.wx0 = st.wx1 = st.wy0 = st.wy1 = 0.0f; st.vx0 = st.vx1 = st.vy0 = st.vy1 = false;
    float sg_l = 0.0f, sg_s = 0.0f;
    bool live = false;
    if (active) {
      bool mk;
      const PlaneGeom g = plane_coords<MODE>(a, cn, b, n, x, y, iy_disp, mk);
      if (has_mask) {
        if ((n & 31) == 0) bits = __float_as_uint(o.stash[((long)b * SK + kStashBase + (n >> 5)) * HW + pix]);
        mk = (bits >> (n & 31)) & 1u;
      }
      const long pl = ((long)b * a.N + n) * HW;
      float gd_dense = 0.0f;
      if (mk) {
        const Tap t = make_tap(g.ix, g.iy, a.W, a.H);
        float dlx, dly, dsx = 0, dsy = 0, d0x, d0y, d1x, d1y, d2x, d2y;
        const TapK tk = tap_kernel(t, a.W, a.H);
        const float l = sample_vg_k(a.logits + pl, tk, dlx, dly);
        const float c0 = sample_vg_k(srcb, tk, d0x, d0y);
        const float c1 = sample_vg_k(srcb + HW, tk, d1x, d1y);
        const float c2 = sample_vg_k(srcb + 2 * HW, tk, d2x, d2y);
        const float s = MIX ? sample_vg_k(a.sigma + pl, tk, dsx, dsy) : 0.0f;
        PlaneGrad pg;
        if (render) {  // alpha compositing: d prob_k / d alpha_n for k >= n through the transmittance (trainer.py:584-591)
          const bool last = (n == a.N - 1);
          const float dist = last ? 0.0f : a.dists[((long)b * (a.N - 1) + n) * HW + pix];
          const float alpha = render_alpha(l, dist, last);
          const float pn = alpha * T;
          pg = plane_grad_p<MIX>(c, pn, s, c0, c1, c2);
          prefix += pg.g_l * pn;
          const float keep = 1.0f - alpha + 1e-10f;
          const float g_alpha = pg.g_l * T - (Rtot - prefix) / keep;
          const float da = (1.0f - alpha);  // d alpha / d (relu(l) * dist)
          pg.g_l = (!last && l > 0.0f) ? g_alpha * dist * da : 0.0f;
          if (o.g_dists && !last && !ghost) o.g_dists[((long)b * (a.N - 1) + n) * HW + pix] = g_alpha * fmaxf(l, 0.0f) * da;
          T *= keep;
        } else {
          pg = plane_grad<MIX>(c, l, s, c0, c1, c2);
        }
        const float g_l = pg.g_l, g_s = pg.g_s, gc0 = pg.gc0, gc1 = pg.gc1, gc2 = pg.gc2;
        st = t; sg_l = g_l; sg_s = g_s; live = true;
        if (want_plane) {
          // d loss / d (ix, iy) in pixels, then back through grid_sample's un-normalisation ((size-1)/2) and the
          // reference's normalisation (*2, /(size-1)) in autograd's order.
          const float gix = g_l * dlx + g_s * dsx + gc0 * d0x + gc1 * d1x + gc2 * d2x;
          const float gpx = gix * gscale_x;  // (W-1)/2 * 2 / (W-1): gradient side, no bit-exactness at stake
          if (MODE == PD_WARP_DISP) {
            gk[0] = gpx * a.sign;
            gd_dense = gk[0];
          } else {
            const float giy = g_l * dly + g_s * dsy + gc0 * d0y + gc1 * d1y + gc2 * d2y;
            const float gpy = giy * gscale_y;
            // gradient side: a refined reciprocal instead of three IEEE divisions (measured: no time difference — the
            // kernel is bound by the L2's atomic rate, not by VALU — but 30 instructions less)
            float inv_z = fast_rcp(g.zc);
            inv_z = fmaf(fmaf(-g.zc, inv_z, 1.0f), inv_z, inv_z);   // one Newton step: ~0.5 ulp
            const float gp0 = gpx * inv_z, gp1 = gpy * inv_z;
            const float gz = g.z_clamped ? 0.0f : -(gp0 * g.p0 + gp1 * g.p1) * inv_z;

            const float fx = (float)x, fy = (float)y;
            gk[0] = gp0 * fx; gk[1] = gp0 * fy; gk[2] = gp0;
            gk[3] = gp1 * fx; gk[4] = gp1 * fy; gk[5] = gp1;
            gk[6] = gz * fx;  gk[7] = gz * fy;  gk[8] = gz;
          }
        }
      }
      if (render && !mk && o.g_dists && n < a.N - 1 && !ghost) o.g_dists[((long)b * (a.N - 1) + n) * HW + pix] = 0.0f;
      if (dense && want_plane && !ghost) o.g_plane[pl + pix] = gd_dense;
    }
    if (ghost) {  // the ghost only lends its right-tap slots to lane 1
      sg_l = sg_s = 0.0f;
#pragma unroll
      for (int k = 0; k < K; ++k) gk[k] = 0.0f;
    }
    if (TOSCRATCH) {
      if (active) {
        const long e = ((long)b * a.N + n) * HW + pix;
        if (MIX) reinterpret_cast<float2*>(o.scratch)[e] = make_float2(sg_l, sg_s);
        else o.scratch[e] = sg_l;
      }
    } else {  // adjoint of the bilinear gather, all 64 lanes together (pd_common.h: neighbours share their atomics)
      const long pl = ((long)b * a.N + n) * HW;
#ifdef PD_GEN_NOSCATTER  // diagnostics: the kernel without its atomics (keeps the values alive through one lane's store)
      if (sg_l + sg_s == 123.456f) o.g_logits[pl] = sg_l;
#else
      const ScatterPlan sp = plan_scatter(st, a.W, live);
      if (MIX && o.g_sigma) bilinear_scatter_wave(o.g_sigma + pl, st, a.W, sg_s, live, sp);
      if (o.g_logits) bilinear_scatter_wave(o.g_logits + pl, st, a.W, sg_l, live, sp);
#endif
    }
    if (reduce_plane) {
      const int ln = threadIdx.x & (kWave - 1);
#pragma unroll
      for (int k = 0; k + 1 < K; k += 2) {  // two components per DPP reduction: totals in lanes 31 and 63
        const float v = half_wave_sums_hi(gk[k], gk[k + 1]);
        if ((ln & 31) == 31) lds_add(&red[n * K + k + (ln >> 5)], v);  // LDS atomic, 4 waves
      }
      if (K & 1) {
        const float v = wave_sum_hi(gk[K - 1]);
        if (ln == kWave - 1) lds_add(&red[n * K + K - 1], v);
      }
    }
  }
  if (reduce_plane) {
    __syncthreads();
    float* dst = o.partials + ((long)b * gridDim.x + blockIdx.x) * a.N * K;
    for (int i = threadIdx.x; i < a.N * K; i += kBlock) dst[i] = red[i];
  }
}

// partials [B][nblk][M] -> out [B][M], summed in a fixed order (deterministic)
__global__ void reduce_partials_kernel(const float* __restrict__ partials, float* __restrict__ out, int nblk, int M) {
  const int j = blockIdx.x, b = blockIdx.y;  // one wave per output element; lanes stride over the blocks
  const float* p = partials + (long)b * nblk * M + j;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < nblk; i += kWave) acc += p[(long)i * M];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * M + j] = acc;
}

int reduce_partials(const float* partials, float* out, int nblk, int M, int B, hipStream_t stream) {
  reduce_partials_kernel<<<dim3(M, B), kWave, 0, stream>>>(partials, out, nblk, M);
  return check_launch("reduce_partials_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// Per-plane tensors of trainer.py:582-602 (forward values only)
// ---------------------------------------------------------------------------------------------------------------
struct LayersOut {
  float* rgb_rec_layered;  // [B,N,3,H,W]
  float* logit_rec;        // [B,N,H,W]
  float* probability_rec;  // [B,N,H,W]  softmax, or mixture weights when MIX (trainer.py:602)
  float* sigma_rec;        // [B,N,H,W]
  float* pi_rec;           // [B,N,H,W]
};

template <int MODE, bool MIX>
__global__ __launch_bounds__(kBlock) void sweep_layers_kernel(SweepArgs a, LayersOut o) {
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= HW) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const bool has_mask = (MODE == PD_WARP_DISP) && a.padding_mask != nullptr;
  const float* srcb = a.src + (long)b * 3 * HW;
  const float iy_disp = (MODE == PD_WARP_DISP) ? normalise_roundtrip((float)y, (float)(a.H - 1)) : 0.0f;
  const CoordNorm cn = make_coord_norm(a.W, a.H);
  // pass 1: softmax statistics; pass 2: write
  const bool render = a.flags & PD_RENDER_PROB;
  RenderState rs;
  float Srender = 0.0f;
  float m_run = -INFINITY, Z = 0.0f, S = 0.0f;
  for (int pass = 0; pass < 2; ++pass) {
    const float lse2 = (pass == 1) ? m_run + log2_fast(Z) : 0.0f;
    const float invSn = (pass == 1 && MIX) ? Z / S : 0.0f;
    for (int n = 0; n < a.N; ++n) {
      bool mk;
      const PlaneGeom g = plane_coords<MODE>(a, cn, b, n, x, y, iy_disp, mk);
      if (has_mask) mk = read_mask(a, b, n, x, y);
      float l = 0, s = 0, c0 = 0, c1 = 0, c2 = 0;
      const long pl = ((long)b * a.N + n) * HW;
      if (mk) {
        const TapK t = tap_kernel(make_tap(g.ix, g.iy, a.W, a.H), a.W, a.H);
        l = sample_k(a.logits + pl, t);
        if (MIX) s = sample_k(a.sigma + pl, t);
        if (pass == 1 && o.rgb_rec_layered) {
          c0 = sample_k(srcb, t);
          c1 = sample_k(srcb + HW, t);
          c2 = sample_k(srcb + 2 * HW, t);
        }
      }
      const float sg = fminf(fmaxf(s, kSigmaMin), kSigmaMax);
      const float l2 = l * kLog2e;
      if (render) {  // alpha compositing needs no normaliser: a single (second) pass
        if (pass == 0) continue;
        const bool last = (n == a.N - 1);
        const float dist = last ? 0.0f : a.dists[((long)b * (a.N - 1) + n) * HW + pix];
        const float pr = render_prob(rs, render_alpha(l, dist, last));
        if (o.rgb_rec_layered) {
          float* q = o.rgb_rec_layered + ((long)b * a.N + n) * 3 * HW + pix;
          q[0] = c0; q[HW] = c1; q[2 * HW] = c2;
        }
        if (o.logit_rec) o.logit_rec[pl + pix] = l;
        if (MIX) {
          if (o.sigma_rec) o.sigma_rec[pl + pix] = sg;
          if (o.pi_rec) o.pi_rec[pl + pix] = pr;
          if (o.probability_rec) o.probability_rec[pl + pix] = pr / sg;  // normalised by the caller-side sum below
        } else if (o.probability_rec) {
          o.probability_rec[pl + pix] = pr;
        }
        Srender += pr / sg;
        continue;
      }
      if (pass == 0) {
        if (l2 > m_run) {
          const float sc = exp2_fast(m_run - l2);
          Z *= sc; S *= sc;
          m_run = l2;
        }
        const float p = exp2_fast(l2 - m_run);
        Z += p;
        if (MIX) S += p / sg;
      } else {
        const float pi = exp2_fast(l2 - lse2);
        if (o.rgb_rec_layered) {
          float* q = o.rgb_rec_layered + ((long)b * a.N + n) * 3 * HW + pix;
          q[0] = c0; q[HW] = c1; q[2 * HW] = c2;
        }
        if (o.logit_rec) o.logit_rec[pl + pix] = l;
        if (MIX) {
          if (o.sigma_rec) o.sigma_rec[pl + pix] = sg;
          if (o.pi_rec) o.pi_rec[pl + pix] = pi;
          if (o.probability_rec) o.probability_rec[pl + pix] = pi / sg * invSn;
        } else if (o.probability_rec) {
          o.probability_rec[pl + pix] = pi;
        }
      }
    }
  }
  if (render && MIX && o.probability_rec) {  // weights_rec = (pi/sigma) / sum(pi/sigma)  (trainer.py:600-602)
    const float inv = 1.0f / Srender;
    for (int n = 0; n < a.N; ++n) o.probability_rec[((long)b * a.N + n) * HW + pix] *= inv;
  }
}

}  // namespace pd

// =================================================================================================================
// C ABI
// =================================================================================================================
using namespace pd;

static int validate(const pd_sweep_desc* d, const float* src, const float* logits, const float* sigma,
                    const float* plane, const float* plane_aux, const float* inv_K3, const float* padding_mask) {
  PD_REQUIRE(d != nullptr, "desc is NULL");
  PD_REQUIRE(d->B > 0 && d->N > 0 && d->H > 1 && d->W > 1, "bad shape B=%d N=%d H=%d W=%d (need H,W >= 2)", d->B, d->N,
             d->H, d->W);
  PD_REQUIRE(d->B <= 65535, "B=%d exceeds the grid.y limit", d->B);
  PD_REQUIRE(d->mode == PD_WARP_DISP || d->mode == PD_WARP_HOMOGRAPHY, "unknown warp mode %d", d->mode);
  PD_REQUIRE(src && logits && plane, "src/logits/plane must not be NULL");
  PD_REQUIRE(!(d->flags & PD_MIXTURE) || sigma, "PD_MIXTURE needs sigma");
  if (d->mode == PD_WARP_HOMOGRAPHY) {
    PD_REQUIRE(plane_aux && inv_K3, "homography mode needs plane_aux (R n) and inv_K3");
    PD_REQUIRE(padding_mask == nullptr || (d->flags & PD_HOMO_UNIFORM),
               "homography mode computes its own padding mask; pass NULL (PD_HOMO_UNIFORM: the [B,N,3] translation weights)");
    PD_REQUIRE(!(d->flags & (PD_DISP_DENSE | PD_DISP_ROWS)), "PD_DISP_DENSE / PD_DISP_ROWS are disp-mode flags");
  } else {
    PD_REQUIRE(!(d->flags & PD_HOMO_UNIFORM), "PD_HOMO_UNIFORM is a homography-mode flag");
  }
  PD_REQUIRE(!(d->flags & PD_BWD_DEFER_GATHER) || (d->flags & PD_HOMO_UNIFORM), "PD_BWD_DEFER_GATHER goes with PD_HOMO_UNIFORM");
  PD_REQUIRE(!(d->flags & PD_HOMO_UNIFORM) || (size_t)d->N * d->H * d->W < ((size_t)1 << 29),
             "PD_HOMO_UNIFORM: one image's N*H*W = %zu needs 32-bit byte offsets (< 2^29 elements)", (size_t)d->N * d->H * d->W);
  PD_REQUIRE(!((d->flags & PD_DISP_DENSE) && (d->flags & PD_DISP_ROWS)), "PD_DISP_DENSE and PD_DISP_ROWS exclude each other");
  if ((d->flags & PD_DISP_ROWS) && !pd_sweep_uses_rowshift(d)) {
    set_error("PD_DISP_ROWS is served by the row-shift kernels only (pd_sweep_uses_rowshift); pass a dense map instead");
    return PD_ERR_UNSUPPORTED;
  }
  if (d->flags & PD_MASK_ROWS) {
    PD_REQUIRE(d->mode == PD_WARP_DISP && padding_mask, "PD_MASK_ROWS needs disp mode and a [B,N,H] padding mask");
    if (!pd_sweep_uses_rowshift(d)) {
      set_error("PD_MASK_ROWS is served by the row-shift kernels only (pd_sweep_uses_rowshift); pass the dense mask instead");
      return PD_ERR_UNSUPPORTED;
    }
  }
  return PD_OK;
}

// What PD_IMPL_AUTO does with a second source row whose weight is fp32 noise of the reference's y round trip
// (pd_rowshift_common.h: two_row_form): it is dropped when the weight is below PD_AUTO_ROW_EPS (0: never).  Decided by the
// parity suite run under the modes (tests/test_gpu_parity.py: row_mode; profiles/r05_parity.md).  PD_ROW_EPS in the
// environment (read once, diagnostics) overrides the compiled-in value.
#ifndef PD_AUTO_ROW_EPS
#define PD_AUTO_ROW_EPS 0.0f
#endif
static float auto_row_eps() {
  static const float eps = [] { const char* e = getenv("PD_ROW_EPS"); return e ? (float)atof(e) : (float)(PD_AUTO_ROW_EPS); }();
  return eps;
}
static SweepArgs make_args(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                           const float* sigma, const float* plane, const float* plane_aux, const float* inv_K3,
                           const float* padding_mask, const float* dists = nullptr) {
  SweepArgs a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W;
  a.flags = d->flags;
  a.sign = d->sign;
  a.stash_k = kStashBase + ((d->mode == PD_WARP_DISP) ? (d->N + 31) / 32 : 0);
  const bool mask_rows = (d->flags & PD_MASK_ROWS) != 0;
  a.row_eps = (d->impl == PD_IMPL_FAST_ROWS) ? kFastRowWeight : (d->impl == PD_IMPL_EXACT_ROWS) ? 0.0f : auto_row_eps();
  a.fast_rows = a.row_eps > 0.0f ? 1 : 0;
  // row pairs need one scalar disparity per plane (the sampling column is then the same in both rows) and no per-pixel
  // or per-row mask; PD_NO_ROWPAIR=1 (environment) switches them off for A/B runs
  a.pairs = (d->mode == PD_WARP_DISP && !(d->flags & (PD_DISP_DENSE | PD_DISP_ROWS | PD_MASK_ROWS | PD_RENDER_PROB)) &&
             padding_mask == nullptr && !a.fast_rows && !switches().no_rowpair) ? 1 : 0;
  a.has_mask = (d->mode == PD_WARP_DISP && padding_mask != nullptr && !mask_rows) ? 1 : 0;
  a.src = src; a.tgt = tgt; a.logits = logits; a.sigma = sigma;
  a.plane = plane; a.plane_aux = plane_aux; a.inv_K3 = inv_K3;
  a.padding_mask = mask_rows ? nullptr : padding_mask;
  a.mask_rows = mask_rows ? padding_mask : nullptr;
  a.dists = dists;
  a.ph_mean = nullptr;
  a.inv_numel = 1.0f / ((float)d->B * (float)d->H * (float)d->W);
  return a;
}

extern "C" float pd_sweep_auto_row_eps(void) { return auto_row_eps(); }

static bool wants_rowshift(const pd_sweep_desc* d) {
  return d->impl == PD_IMPL_AUTO || d->impl == PD_IMPL_FAST_ROWS || d->impl == PD_IMPL_ROWS1 || d->impl == PD_IMPL_UNIFORM_DIRECT ||
         d->impl == PD_IMPL_EXACT_ROWS;
}

extern "C" int pd_sweep_uses_rowshift(const pd_sweep_desc* d) {
  return (d && wants_rowshift(d) && rowshift_applicable(d)) ? 1 : 0;
}

extern "C" int pd_sweep_bwd_accumulates(const pd_sweep_desc* d) {
  if (!d) return 0;
  if (wants_rowshift(d) && rowshift_applicable(d)) return 0;   // owner-computes ring stores: no read-modify-write form
  if (d->mode == PD_WARP_HOMOGRAPHY && (d->flags & PD_HOMO_UNIFORM)) return 1;
#ifdef PD_EXPERIMENTS
  if (tile_bwd_applicable(d)) return 0;
#endif
  return 1;                                                    // the atomic scatter accumulates by nature
}

extern "C" size_t pd_sweep_stash_floats(const pd_sweep_desc* d) {
  if (!d) return 0;
  const size_t words = (d->mode == PD_WARP_DISP) ? (size_t)(d->N + 31) / 32 : 0;
  return (size_t)(kStashBase + words) * d->H * d->W;
}

// workgroups of the general backward: four waves of 63 pixels + 1 ghost lane each
static int bwd_blocks(int HW) { return ceil_div(HW, (kBlock / kWave) * (kWave - 1)); }

extern "C" size_t pd_sweep_bwd_workspace_floats(const pd_sweep_desc* d) {
  if (!d) return 0;
  const size_t K = (d->mode == PD_WARP_DISP) ? 1 : 9;
  const size_t general = (size_t)d->B * bwd_blocks(d->H * d->W) * d->N * K;
  const size_t rows = rowshift_applicable(d) ? rowshift_bwd_workspace_floats(d) : 0;
#ifdef PD_EXPERIMENTS
  const size_t tiles = tile_bwd_applicable(d) ? tile_bwd_workspace_floats(d) : 0;
#else
  const size_t tiles = 0;
#endif
  const size_t uni = (d->mode == PD_WARP_HOMOGRAPHY && (d->flags & PD_HOMO_UNIFORM)) ? uniform_bwd_workspace_floats(d) : 0;
  const size_t gat = gather_bwd_applicable(d) ? gather_bwd_workspace_floats(d) : 0;
  size_t m = general > rows ? general : rows;
  m = m > tiles ? m : tiles;
  m = m > gat ? m : gat;
  return m > uni ? m : uni;
}

#define PD_DISPATCH(KERNEL, mode, mix, grid, block, shmem, stream, ...)                                   \
  do {                                                                                                     \
    if ((mode) == PD_WARP_DISP) {                                                                          \
      if (mix) KERNEL<PD_WARP_DISP, true><<<grid, block, shmem, stream>>>(__VA_ARGS__);                    \
      else     KERNEL<PD_WARP_DISP, false><<<grid, block, shmem, stream>>>(__VA_ARGS__);                   \
    } else {                                                                                               \
      if (mix) KERNEL<PD_WARP_HOMOGRAPHY, true><<<grid, block, shmem, stream>>>(__VA_ARGS__);              \
      else     KERNEL<PD_WARP_HOMOGRAPHY, false><<<grid, block, shmem, stream>>>(__VA_ARGS__);             \
    }                                                                                                      \
  } while (0)

extern "C" int pd_plane_sweep_fwd(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                                  const float* sigma, const float* plane, const float* plane_aux, const float* inv_K3,
                                  const float* padding_mask, const float* dists, float* rgb_rec, float* ph_map,
                                  float* ph_mean, float* stash, pd_stream_t stream) {
  int rc = validate(d, src, logits, sigma, plane, plane_aux, inv_K3, padding_mask);
  if (rc) return rc;
  PD_REQUIRE(tgt && rgb_rec && ph_map && stash, "tgt/rgb_rec/ph_map/stash must not be NULL");
  PD_REQUIRE(!(d->flags & PD_RENDER_PROB) || (dists && d->N >= 2), "PD_RENDER_PROB needs dists [B,N-1,H,W] and N >= 2");
  // the stash always reserves the mask words in disp mode (pd_sweep_stash_floats); they are written when a mask exists
  SweepArgs a = make_args(d, src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists);
  if (ph_mean) {  // the kernels add into it: start from zero (an async memset on the same stream, unless the caller did it)
    a.ph_mean = ph_mean;
    if (!(d->flags & PD_PH_MEAN_ZEROED) &&
        hipMemsetAsync(ph_mean, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return check_launch("hipMemsetAsync");
  }
  if (wants_rowshift(d) && rowshift_applicable(d)) {
    // default for the headline shape: one wave per 128-pixel segment streams over the planes (pd_plane_sweep_fwdstream.hip);
    // PD_IMPL_ROWS1 keeps the plane-group row-shift forward (cross-check, A/B)
    if (d->impl != PD_IMPL_ROWS1 && fwdstream_applicable(d, a)) return fwdstream_fwd(d, a, rgb_rec, ph_map, stash, (hipStream_t)stream);
#ifdef PD_EXPERIMENTS   // wide-access forward (2 / 4 pixels per lane): faster isolated, slower inside the step (NOTEBOOK.md 3.5.6)
    if (rowquad_applicable(d, a.has_mask != 0) && getenv("PD_QUAD_FWD"))
      return rowquad_fwd(d, a, rgb_rec, ph_map, stash, (hipStream_t)stream);
#endif
    return rowshift_fwd(d, a, rgb_rec, ph_map, stash, (hipStream_t)stream);
  }
  if (d->mode == PD_WARP_HOMOGRAPHY && (d->flags & PD_HOMO_UNIFORM))
    return uniform_fwd(d, a, rgb_rec, ph_map, stash, (hipStream_t)stream);
  dim3 grid(ceil_div(d->H * d->W, kBlock), d->B);
  PD_DISPATCH(sweep_fwd_kernel, d->mode, (d->flags & PD_MIXTURE) != 0, grid, dim3(kBlock), 0, (hipStream_t)stream, a,
              rgb_rec, ph_map, stash);
  return check_launch("sweep_fwd_kernel");
}

struct TailIn { const float* raw_sigma; const float* stash; const float* disp; const float* g_disp; const float* g_depth; };

static int sweep_bwd_impl(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                          const float* sigma, const float* plane, const float* plane_aux, const float* inv_K3,
                          const float* padding_mask, const float* dists, const float* rgb_rec,
                          const float* stash, const float* g_rgb_rec, const float* g_ph_map,
                          const float* g_ph_mean, float* g_logits, float* g_sigma, float* g_plane,
                          float* g_dists, float* workspace, pd_stream_t stream_, const TailIn* tail) {
  int rc = validate(d, src, logits, sigma, plane, plane_aux, inv_K3, padding_mask);
  if (rc) return rc;
  PD_REQUIRE(tgt && rgb_rec && stash, "tgt/rgb_rec/stash must not be NULL");
  PD_REQUIRE(!(d->flags & PD_RENDER_PROB) || (dists && d->N >= 2), "PD_RENDER_PROB needs dists [B,N-1,H,W] and N >= 2");
  const bool dense = (d->flags & (PD_DISP_DENSE | PD_DISP_ROWS)) != 0;
  PD_REQUIRE(!g_plane || dense || workspace, "g_plane needs workspace (pd_sweep_bwd_workspace_floats)");
  PD_REQUIRE(d->impl >= PD_IMPL_AUTO && d->impl <= PD_IMPL_EXACT_ROWS, "unknown impl %d", d->impl);
  hipStream_t stream = (hipStream_t)stream_;
  const bool mix = (d->flags & PD_MIXTURE) != 0;
  const bool accumulate = (d->flags & PD_BWD_ACCUMULATE) != 0;
  PD_REQUIRE(!accumulate || pd_sweep_bwd_accumulates(d), "PD_BWD_ACCUMULATE is not served for this descriptor (pd_sweep_bwd_accumulates)");
  SweepArgs ak = make_args(d, src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists);
  BwdOut o;
  o.g_logits = g_logits; o.g_sigma = mix ? g_sigma : nullptr; o.g_plane = g_plane; o.partials = workspace;
  o.side = nullptr; o.scratch = nullptr;
  o.g_dists = (d->flags & PD_RENDER_PROB) ? g_dists : nullptr;
  o.rgb_rec = rgb_rec; o.stash = stash; o.g_rgb_rec = g_rgb_rec; o.g_ph_map = g_ph_map; o.g_ph_mean = g_ph_mean;
  if (wants_rowshift(d) && rowshift_applicable(d)) {
    PD_REQUIRE(workspace, "the row-shift backward needs workspace (pd_sweep_bwd_workspace_floats)");
#ifdef PD_EXPERIMENTS
    if (rowquad_applicable(d, ak.has_mask != 0) && getenv("PD_QUAD_BWD")) return rowquad_bwd(d, ak, o, stream);
#endif
    // default: lanes own aligned source slots, waves stream along plane rows (pd_plane_sweep_rowstream.hip);
    // PD_IMPL_ROWS1 keeps the target-ordered row-shift backward (cross-check, A/B)
    if (tail) {   // pd_plane_sweep_bwd_tail: the row-stream backward with the decoder tail's backward riding along
      if (d->impl == PD_IMPL_ROWS1 || !rowstream_bwd_tail_applicable(d, ak)) {
        set_error("pd_plane_sweep_bwd_tail: not served for this descriptor (pd_sweep_bwd_tail_fuses)");
        return PD_ERR_UNSUPPORTED;
      }
      PD_REQUIRE(g_logits && g_sigma, "pd_plane_sweep_bwd_tail writes both g_raw_logits and g_raw_sigma");
      o.tail_raw_sigma = tail->raw_sigma; o.tail_stash = tail->stash; o.tail_disp = tail->disp;
      o.tail_g_disp = tail->g_disp; o.tail_g_depth = tail->g_depth;
      return rowstream_bwd(d, ak, o, stream);
    }
    if (d->impl != PD_IMPL_ROWS1 && rowstream_bwd_applicable(d, ak)) return rowstream_bwd(d, ak, o, stream);
    return rowshift_bwd(d, ak, o, stream);
  }
  if (tail) {
    set_error("pd_plane_sweep_bwd_tail: not served for this descriptor (pd_sweep_bwd_tail_fuses)");
    return PD_ERR_UNSUPPORTED;
  }
  if (d->mode == PD_WARP_HOMOGRAPHY && (d->flags & PD_HOMO_UNIFORM)) {
    PD_REQUIRE(workspace, "the plane-uniform backward needs workspace (pd_sweep_bwd_workspace_floats)");
    return uniform_bwd(d, ak, o, workspace, stream);
  }
#ifdef PD_EXPERIMENTS
  if (tile_bwd_applicable(d)) {   // PD_IMPL_TILE: source tiles owned by workgroups, no atomics, no zero-fill
    PD_REQUIRE(workspace, "the tile backward needs workspace (pd_sweep_bwd_workspace_floats)");
    return tile_bwd(d, ak, o, workspace, stream);
  }
#else
  if (d->impl == PD_IMPL_TILE) {
    set_error("PD_IMPL_TILE (the owned-tile backward) is built with -DPD_EXPERIMENTS only: it is slower than the default kernels");
    return PD_ERR_UNSUPPORTED;
  }
#endif
  if (gather_bwd_applicable(d) && (g_logits || g_sigma)) {   // one homography per plane: two passes, no atomics (pd_plane_sweep_gather.hip)
    PD_REQUIRE(workspace, "the gather backward needs workspace (pd_sweep_bwd_workspace_floats)");
    const GatherPlan gp = gather_bwd_plan(d, workspace);
    o.partials = gp.partials; o.scratch = gp.scratch;
    rc = gather_bwd_prepare(d, ak, gp, stream);
    if (rc) return rc;
    const dim3 grid1(gp.nblk, d->B);
    const size_t shmem1 = (size_t)d->N * 9 * sizeof(float);
    if (mix) sweep_bwd_kernel<PD_WARP_HOMOGRAPHY, true, true><<<grid1, kBlock, shmem1, stream>>>(ak, o);
    else     sweep_bwd_kernel<PD_WARP_HOMOGRAPHY, false, true><<<grid1, kBlock, shmem1, stream>>>(ak, o);
    rc = check_launch("sweep_bwd_kernel (scratch)");
    if (rc) return rc;
    rc = gather_bwd_finish(d, ak, o, gp, stream);
    if (rc || !g_plane) return rc;
    const int M = d->N * 9;
    reduce_partials_kernel<<<dim3(M, d->B), kWave, 0, stream>>>(gp.partials, g_plane, gp.nblk, M);
    return check_launch("reduce_partials_kernel");
  }
  const size_t plane_bytes = (size_t)d->B * d->N * d->H * d->W * sizeof(float);
  const int HW = d->H * d->W;
  dim3 grid(bwd_blocks(HW), d->B);
  const int K = (d->mode == PD_WARP_DISP) ? 1 : 9;
  // general path: the bilinear adjoint is an atomic scatter into zero-filled gradients
  if (g_logits && !accumulate) (void)hipMemsetAsync(g_logits, 0, plane_bytes, stream);
  if (g_sigma && !accumulate) (void)hipMemsetAsync(g_sigma, 0, plane_bytes, stream);
  const size_t shmem = (size_t)d->N * K * sizeof(float);
  PD_DISPATCH(sweep_bwd_kernel, d->mode, mix, grid, dim3(kBlock), shmem, stream, ak, o);
  rc = check_launch("sweep_bwd_kernel");
  if (rc) return rc;
  if (g_plane && !dense) {
    const int M = d->N * K;
    reduce_partials_kernel<<<dim3(M, d->B), kWave, 0, stream>>>(workspace, g_plane, grid.x, M);
    rc = check_launch("reduce_partials_kernel");
  }
  return rc;
}

extern "C" int pd_plane_sweep_bwd(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                                  const float* sigma, const float* plane, const float* plane_aux, const float* inv_K3,
                                  const float* padding_mask, const float* dists, const float* rgb_rec,
                                  const float* stash, const float* g_rgb_rec, const float* g_ph_map,
                                  const float* g_ph_mean, float* g_logits, float* g_sigma, float* g_plane,
                                  float* g_dists, float* workspace, pd_stream_t stream) {
  return sweep_bwd_impl(d, src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, rgb_rec, stash, g_rgb_rec,
                        g_ph_map, g_ph_mean, g_logits, g_sigma, g_plane, g_dists, workspace, stream, nullptr);
}

extern "C" int pd_sweep_bwd_plane_adds(const pd_sweep_desc* d) {
  if (!d || !(wants_rowshift(d) && rowshift_applicable(d)) || d->impl == PD_IMPL_ROWS1 || (d->flags & (PD_DISP_ROWS | PD_DISP_DENSE))) return 0;
  SweepArgs probe;
  probe.has_mask = 0;   // (the caller passes no per-pixel mask when it relies on this: pd_plane_sweep_bwd falls back to overwriting otherwise)
  return rowstream_bwd_applicable(d, probe) ? 1 : 0;
}

extern "C" int pd_sweep_bwd_tail_fuses(const pd_sweep_desc* d) {
  if (!d || !(wants_rowshift(d) && rowshift_applicable(d)) || d->impl == PD_IMPL_ROWS1) return 0;
  SweepArgs probe;
  probe.has_mask = 0;
  return rowstream_bwd_tail_applicable(d, probe) ? 1 : 0;
}

extern "C" int pd_plane_sweep_bwd_tail(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                                       const float* sigma, const float* plane, const float* rgb_rec, const float* stash,
                                       const float* g_rgb_rec, const float* g_ph_map, const float* g_ph_mean,
                                       const float* raw_sigma, const float* tail_stash, const float* disp,
                                       const float* g_disp, const float* g_depth, float* g_raw_logits,
                                       float* g_raw_sigma, float* g_plane, float* workspace, pd_stream_t stream) {
  PD_REQUIRE(raw_sigma && tail_stash && disp, "raw_sigma / tail_stash / disp must not be NULL");
  const TailIn t = {raw_sigma, tail_stash, disp, g_disp, g_depth};
  return sweep_bwd_impl(d, src, tgt, logits, sigma, plane, nullptr, nullptr, nullptr, nullptr, rgb_rec, stash, g_rgb_rec,
                        g_ph_map, g_ph_mean, g_raw_logits, g_raw_sigma, g_plane, nullptr, workspace, stream, &t);
}

extern "C" int pd_uniform_gather_pair(const pd_sweep_desc* d, const float* plane_a, const float* inv_K3_a, float* workspace_a,
                                      const float* plane_b, const float* inv_K3_b, float* workspace_b, float* g_logits,
                                      float* g_sigma, pd_stream_t stream) {
  PD_REQUIRE(d != nullptr, "desc is NULL");
  PD_REQUIRE(d->mode == PD_WARP_HOMOGRAPHY && (d->flags & PD_HOMO_UNIFORM) && (d->flags & PD_BWD_DEFER_GATHER),
             "pd_uniform_gather_pair takes the descriptor of two PD_HOMO_UNIFORM | PD_BWD_DEFER_GATHER backward calls");
  PD_REQUIRE(d->B > 0 && d->B <= 65535 && d->N > 0 && d->H > 1 && d->W > 1, "bad shape");
  PD_REQUIRE(plane_a && inv_K3_a && workspace_a && plane_b && inv_K3_b && workspace_b && g_logits, "NULL argument");
  PD_REQUIRE(!(d->flags & PD_MIXTURE) || g_sigma, "PD_MIXTURE needs g_sigma");
  return uniform_gather_pair(d, plane_a, inv_K3_a, workspace_a, plane_b, inv_K3_b, workspace_b, g_logits, g_sigma,
                             (hipStream_t)stream);
}

static int validate_pair(const pd_sweep_desc* d, const float* src, const float* logits, const float* sigma,
                         const pd_sweep_view* va, const pd_sweep_view* vb) {
  PD_REQUIRE(d && va && vb, "desc / view is NULL");
  PD_REQUIRE(d->mode == PD_WARP_HOMOGRAPHY && (d->flags & PD_HOMO_UNIFORM), "the pair entry points serve PD_HOMO_UNIFORM views");
  for (const pd_sweep_view* v : {va, vb}) {
    const int rc = validate(d, src, logits, sigma, v->plane, v->plane_aux, v->inv_K3, v->padding_mask);
    if (rc) return rc;
    PD_REQUIRE(v->tgt && v->rgb_rec && v->stash, "view: tgt / rgb_rec / stash must not be NULL");
    PD_REQUIRE(!(d->flags & PD_RENDER_PROB) || (v->dists && d->N >= 2), "PD_RENDER_PROB needs dists [B,N-1,H,W] and N >= 2");
  }
  return PD_OK;
}

extern "C" int pd_uniform_fwd_pair(const pd_sweep_desc* d, const float* src, const float* logits, const float* sigma,
                                   const pd_sweep_view* va, const pd_sweep_view* vb, pd_stream_t stream) {
  int rc = validate_pair(d, src, logits, sigma, va, vb);
  if (rc) return rc;
  PD_REQUIRE(va->ph_map && vb->ph_map, "view: ph_map must not be NULL");
  PD_REQUIRE((va->ph_mean == nullptr) == (vb->ph_mean == nullptr), "ph_mean: both views or neither");
  SweepArgs a = make_args(d, src, va->tgt, logits, sigma, va->plane, va->plane_aux, va->inv_K3, nullptr, va->dists);
  SweepArgs b = make_args(d, src, vb->tgt, logits, sigma, vb->plane, vb->plane_aux, vb->inv_K3, nullptr, vb->dists);
  a.ph_mean = va->ph_mean; b.ph_mean = vb->ph_mean;
  if (a.ph_mean && !(d->flags & PD_PH_MEAN_ZEROED)) {
    if (hipMemsetAsync(a.ph_mean, 0, sizeof(float), (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(b.ph_mean, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return check_launch("hipMemsetAsync");
  }
  return uniform_fwd_pair(d, a, va->rgb_rec, va->ph_map, va->stash, b, vb->rgb_rec, vb->ph_map, vb->stash, (hipStream_t)stream);
}

extern "C" int pd_uniform_bwd_pair(const pd_sweep_desc* d, const float* src, const float* logits, const float* sigma,
                                   const pd_sweep_view* va, const pd_sweep_view* vb, float* g_logits, float* g_sigma,
                                   pd_stream_t stream) {
  int rc = validate_pair(d, src, logits, sigma, va, vb);
  if (rc) return rc;
  PD_REQUIRE(va->workspace && vb->workspace, "view: workspace (pd_sweep_bwd_workspace_floats) must not be NULL");
  const bool mix = (d->flags & PD_MIXTURE) != 0;
  PD_REQUIRE(!g_logits || !mix || g_sigma, "PD_MIXTURE needs g_sigma next to g_logits");
  if (switches().uni_chunk && switches().uni_chunk < d->B) {
    set_error("pd_uniform_bwd_pair needs the whole batch's scratch (PD_UNI_CHUNK is set)");
    return PD_ERR_UNSUPPORTED;
  }
  SweepArgs a = make_args(d, src, va->tgt, logits, sigma, va->plane, va->plane_aux, va->inv_K3, va->padding_mask, va->dists);
  SweepArgs b = make_args(d, src, vb->tgt, logits, sigma, vb->plane, vb->plane_aux, vb->inv_K3, vb->padding_mask, vb->dists);
  BwdOut oa{}, ob{};
  for (int i = 0; i < 2; ++i) {
    const pd_sweep_view* v = i ? vb : va;
    BwdOut& o = i ? ob : oa;
    o.g_logits = nullptr; o.g_sigma = nullptr; o.g_plane = v->g_plane; o.partials = v->workspace; o.side = nullptr; o.scratch = nullptr;
    o.g_dists = (d->flags & PD_RENDER_PROB) ? v->g_dists : nullptr;
    o.rgb_rec = v->rgb_rec; o.stash = v->stash; o.g_rgb_rec = v->g_rgb_rec; o.g_ph_map = v->g_ph_map; o.g_ph_mean = v->g_ph_mean;
  }
  return uniform_bwd_pair(d, a, oa, va->workspace, b, ob, vb->workspace, g_logits, mix ? g_sigma : nullptr, (hipStream_t)stream);
}

extern "C" int pd_debug_gather_flags(const pd_sweep_desc* d, const float* workspace, int* host_out, pd_stream_t stream) {
  PD_REQUIRE(d && workspace && host_out, "NULL argument");
  PD_REQUIRE(gather_bwd_applicable(d), "this descriptor does not run the gather backward");
  const GatherPlan gp = gather_bwd_plan(d, const_cast<float*>(workspace));
  if (hipMemcpyAsync(host_out, gp.flags, 2 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return check_launch("pd_debug_gather_flags");
  return PD_OK;
}

extern "C" int pd_plane_sweep_layers(const pd_sweep_desc* d, const float* src, const float* logits,
                                     const float* sigma, const float* plane, const float* plane_aux,
                                     const float* inv_K3, const float* padding_mask, const float* dists,
                                     float* rgb_rec_layered, float* logit_rec, float* probability_rec,
                                     float* sigma_rec, float* pi_rec, pd_stream_t stream) {
  int rc = validate(d, src, logits, sigma, plane, plane_aux, inv_K3, padding_mask);
  if (rc) return rc;
  PD_REQUIRE(!(d->flags & PD_RENDER_PROB) || (dists && d->N >= 2), "PD_RENDER_PROB needs dists [B,N-1,H,W] and N >= 2");
  PD_REQUIRE(!(d->flags & PD_HOMO_UNIFORM), "pd_plane_sweep_layers takes one homography per plane (expand the matrix)");
  SweepArgs a = make_args(d, src, nullptr, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists);
  LayersOut o{rgb_rec_layered, logit_rec, probability_rec, sigma_rec, pi_rec};
  dim3 grid(ceil_div(d->H * d->W, kBlock), d->B);
  PD_DISPATCH(sweep_layers_kernel, d->mode, (d->flags & PD_MIXTURE) != 0, grid, dim3(kBlock), 0, (hipStream_t)stream, a,
              o);
  return check_launch("sweep_layers_kernel");
}
