// Plane-uniform homography sweep: homography_warp when EVERY plane of an image shares one homography.
//
// That is what BASELINE configs[3] feeds the path for the novel frames -1 / +1: Trainer.predict_poses without COLMAP
// builds Rt with the pose net's rotation (conjugated by the crop matrix) and ZERO translation (reference
// trainer.py:386-400), so Rtnd = R + t n^T / d = R for every plane (layers.py:216) and
// H_t2s = inverse(K R K^-1) does not depend on the plane.  Only the facing test (K^-1 p).(R n) > 0 (layers.py:223) still
// does, through the plane normal.  PD_HOMO_UNIFORM: `plane` is [B,4,3,3] (slice 0: the image's H_t2s; slices 1..3: the
// virtual planes that carry the translation's gradient, see uniform_bwd_pass1_kernel), `plane_aux` is [B*N,3].
//
// Forward: the sampling position, the four tap offsets / weights and the three colour samples are computed ONCE per
// target pixel; the plane loop is 8 loads + 8 FMAs + the online softmax (or, PD_RENDER_PROB, the compositing step) — the
// general kernel spends ~190 VALU per pixel and plane on the geometry it re-derives 49 times.
//
// Backward without atomics, in two passes (the scatter pattern is the same for all planes, so it can be inverted once
// per SOURCE pixel and reused 49 times):
//   pass 1 (target-anchored): per pixel and plane the closed-form gradients w.r.t. the sampled logit / sigma
//           (pd_sweep.h) go side by side, as float2, to a scratch [B][N][H*W] with coalesced 8-byte stores (zeros for
//           masked planes); the homography gradient is accumulated per thread over ALL planes and reduced once per
//           workgroup;
//   pass 2 (source-anchored): every source pixel finds the target pixels whose bilinear footprint covers it — the
//           integer points in the pre-image of its 2x2 neighbourhood, located with the forward homography (fp64
//           adjugate), each CONFIRMED with the bit-exact forward coordinate chain, which also yields the exact bilinear
//           weight — keeps up to 12 (slot, weight) pairs in registers, and then gathers
//           g[n][s] = sum_k w_k * scratch[n][t_k] for all planes with plain coalesced stores (PD_BWD_ACCUMULATE: added to
//           what another target view left there).  The scratch reaches the gather through LDS: a workgroup owns a
//           32 x 16 source tile and stages the tile's pre-image box two planes at a time
//           (uniform_bwd_pass2_staged_kernel; uniform_bwd_pass2_kernel is the direct-gather form and the follow-up for
//           source pixels with more than 12 contributors, i.e. strong minification).  Every element of g_logits /
//           g_sigma is written exactly once: no zero-fill, no atomics, deterministic.
// Two views of one source image (the novel frames -1 / +1): pd_uniform_fwd_pair runs both forwards in one launch (PairSlot),
// pd_uniform_bwd_pair the whole backward in one call (first passes, pair gather, reductions).
// The whole batch goes through each pass in one launch (parallelism beat keeping one image's scratch in the 256 MB
// memory-side cache: uniform_chunk); workgroups are dealt to the XCDs in contiguous bands of the image (xcd_banded).
// Opt-in alternative (PD_UNI_FUSED): both passes in one kernel with an LDS hand-over per plane — exact, slower.
// Measured and dropped for the target-side kernels: the two horizontal taps of a row as one 8-byte buffer load with the
// weights permuted onto the pair (4 instead of 8 memory instructions per pixel and plane): forward 0.147 -> 0.157 ms,
// pass 1 0.189 -> 0.211 ms — unlike the row kernels' shifted streams these pairs are not 8-byte aligned AND not
// contiguous across lanes once the view is rotated; and LDS staging of logits / sigma (see uniform_bwd_pass2_staged_kernel).
#include <type_traits>

#include "pd_sweep_geom.h"

namespace pd {

constexpr int kUniK = 12;  // gather entries kept in registers per source pixel (a near-isometric map gives 4, up to 9 where
                           // a perspective term makes the lattices beat)
constexpr int kUniH = 4 * 9;   // floats per image in `plane` ([B,4,3,3], see uniform_bwd_pass1_kernel)

// Per-pixel geometry shared by all planes
struct UniGeom {
  PlaneGeom g;
  float r0, r1, r2;   // K^-1 [x, y, 1]: the facing test's left-hand side
  bool z_ok;
};
__device__ __forceinline__ UniGeom uni_geom(const float* __restrict__ Hm, const float* __restrict__ Ki, const CoordNorm& cn,
                                            int x, int y) {
  UniGeom u;
  const float fx = (float)x, fy = (float)y;
  u.g.p0 = hrow_dot(Hm[0], Hm[1], Hm[2], fx, fy);
  u.g.p1 = hrow_dot(Hm[3], Hm[4], Hm[5], fx, fy);
  const float z = hrow_dot(Hm[6], Hm[7], Hm[8], fx, fy);
  u.r0 = hrow_dot(Ki[0], Ki[1], Ki[2], fx, fy);
  u.r1 = hrow_dot(Ki[3], Ki[4], Ki[5], fx, fy);
  u.r2 = hrow_dot(Ki[6], Ki[7], Ki[8], fx, fy);
  u.z_ok = (z > kZMin);
  u.g.z_clamped = (z < kZMin);
  u.g.zc = u.g.z_clamped ? kZMin : z;
  if (cn.fast) {
    u.g.ix = normalise_roundtrip_rcp(u.g.p0 / u.g.zc, cn.Wm1, cn.rcpW);
    u.g.iy = normalise_roundtrip_rcp(u.g.p1 / u.g.zc, cn.Hm1, cn.rcpH);
  } else {
    u.g.ix = normalise_roundtrip(u.g.p0 / u.g.zc, cn.Wm1);
    u.g.iy = normalise_roundtrip(u.g.p1 / u.g.zc, cn.Hm1);
  }
  return u;
}
__device__ __forceinline__ bool uni_mask(const UniGeom& u, const float* __restrict__ Rn) {
  return (facing_dot(u.r0, u.r1, u.r2, Rn[0], Rn[1], Rn[2]) > 0.0f) && u.z_ok;   // layers.py:223-225, same operation order
}
// The plane loops fetch plane n+1's normal (three scalar loads) while plane n is reduced: loaded where it is used, every
// iteration began with a wait for the scalar cache.
struct RnAhead {
  float q0, q1, q2;
  __device__ __forceinline__ void fetch(const float* __restrict__ Rn, int n) { q0 = Rn[n * 3]; q1 = Rn[n * 3 + 1]; q2 = Rn[n * 3 + 2]; }
  __device__ __forceinline__ bool mask(const UniGeom& u) const { return (facing_dot(u.r0, u.r1, u.r2, q0, q1, q2) > 0.0f) && u.z_ok; }
};

// One image's [N,H,W] block as a buffer resource: a tap load is then "descriptor + the lane's 32-bit tap offset + the plane's
// offset in an SGPR" — no per-lane 64-bit address arithmetic (with flat loads the plane loop spent 16 v_lshl_add_u64 per
// plane on its eight addresses: a quarter of its VALU instructions, and these kernels are VALU-paced: r04 PMC,
// SQ_ACTIVE_INST_VALU = 0.63 of the kernel's cycles).  The offsets are clamped into the plane (tap_kernel), so the range
// check never fires; the host checks that 2 N H W floats fit 32-bit byte offsets.
typedef __amdgpu_buffer_rsrc_t URsrc;
__device__ __forceinline__ URsrc image_rsrc(const float* base, unsigned bytes) {   // `base` must be workgroup-uniform
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ubuf(URsrc r, unsigned lane_off, unsigned plane_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)plane_off, 0));
}
__device__ __forceinline__ float usample_k(URsrc r, const TapK& k, unsigned po) {
  return ubuf(r, k.o00, po) * k.w00 + ubuf(r, k.o01, po) * k.w01 + ubuf(r, k.o10, po) * k.w10 + ubuf(r, k.o11, po) * k.w11;
}
__device__ __forceinline__ float usample_vg_k(URsrc r, const TapK& k, unsigned po, float& dx, float& dy) {
  const float nw = ubuf(r, k.o00, po), ne = ubuf(r, k.o01, po), sw = ubuf(r, k.o10, po), se = ubuf(r, k.o11, po);
  dx = nw * k.x00 + ne * k.x01 + sw * k.x10 + se * k.x11;
  dy = nw * k.y00 + ne * k.y01 + sw * k.y10 + se * k.y11;
  return nw * k.w00 + ne * k.w01 + sw * k.w10 + se * k.w11;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------------------------
// PAIR: two target views of the same source image (trainer.py:532, the novel frames -1 / +1) in ONE launch.  The grid holds
// every tile twice; workgroup ids 8 apart land on the same XCD (ids are dealt round-robin to the 8 XCDs) right after each
// other, so slot 2k of an XCD serves tile k for view A and slot 2k+1 the same tile for view B: the second workgroup finds
// the logits / sigma lines the first one pulled in in that XCD's L2 (pose_net rotations move the footprint by a few
// pixels).  Measured with both halves on the same view (8x49x192x640): forward 2 x 0.140 -> 0.221 ms, pass 1 2 x 0.191 ->
// 0.326 ms.  The arithmetic per view is exactly the single-view kernel's.
struct UniViewB {   // what differs for view B
  const float* tgt; const float* plane; const float* plane_aux; const float* inv_K3; const float* dists;
  float* ph_mean; float* rgb_rec; float* ph_map; float* stash;
};
struct PairSlot { int tile, view; };
__device__ __forceinline__ PairSlot pair_slot(int bx, int nblk2) {   // nblk2 = 2 x tiles
  const int nblk = nblk2 >> 1, full = (nblk / kXcds) * kXcds;   // tiles in whole rounds of the XCDs
  PairSlot p;
  if (bx < 2 * full) {
    const int xcd = bx % kXcds, slot = bx / kXcds;
    p.view = slot & 1;
    p.tile = xcd_banded((slot >> 1) * kXcds + xcd, nblk);
  } else {   // the remainder: neighbours in dispatch order
    p.view = (bx - 2 * full) & 1;
    p.tile = full + ((bx - 2 * full) >> 1);
  }
  return p;
}

template <bool MIX, bool RENDER = false, bool PAIR = false>
__global__ __launch_bounds__(kBlock) void uniform_fwd_kernel(SweepArgs a, float* __restrict__ rgb_rec,
                                                             float* __restrict__ ph_map, float* __restrict__ stash, UniViewB vb) {
  const int HW = a.H * a.W;
  int tile = 0;
  if (PAIR) {
    const PairSlot ps = pair_slot(blockIdx.x, gridDim.x);
    tile = ps.tile;
    if (ps.view) {
      a.tgt = vb.tgt; a.plane = vb.plane; a.plane_aux = vb.plane_aux; a.inv_K3 = vb.inv_K3; a.dists = vb.dists;
      a.ph_mean = vb.ph_mean; rgb_rec = vb.rgb_rec; ph_map = vb.ph_map; stash = vb.stash;
    }
  } else {
    tile = xcd_banded(blockIdx.x, gridDim.x);
  }
  const int pix = tile * kBlock + threadIdx.x;
  const int b = blockIdx.y;
  float ph_val = 0.0f;
  if (pix < HW) {
    const int y = pix / a.W, x = pix - y * a.W;
    const bool automask = a.flags & PD_AUTOMASK;
    const float* srcb = a.src + (long)b * 3 * HW;
    const float t0 = a.tgt[((long)b * 3 + 0) * HW + pix];
    const float t1 = a.tgt[((long)b * 3 + 1) * HW + pix];
    const float t2 = a.tgt[((long)b * 3 + 2) * HW + pix];
    float ea = 0.0f;
    if (automask) ea = fabsf(srcb[pix] - t0) + fabsf(srcb[HW + pix] - t1) + fabsf(srcb[2 * HW + pix] - t2);
    const CoordNorm cn = make_coord_norm(a.W, a.H);
    const UniGeom u = uni_geom(a.plane + (long)b * kUniH, a.inv_K3 + (long)b * 9, cn, x, y);
    const TapK t = tap_kernel(make_tap(u.g.ix, u.g.iy, a.W, a.H), a.W, a.H);
    const float s0 = sample_k(srcb, t), s1 = sample_k(srcb + HW, t), s2 = sample_k(srcb + 2 * HW, t);
    FwdAcc acc;
    RenderState rs;
    constexpr bool render = RENDER;   // alpha compositing over the planes instead of the softmax (a template flag: as a
                                      // run-time one it cost the softmax path 10 % — registers and branches in the plane loop)
    const float* Rn = a.plane_aux + (long)b * a.N * 3;
    const unsigned image_bytes = (unsigned)(a.N * HW) * 4u, plane_bytes = (unsigned)HW * 4u;
    const URsrc rl = image_rsrc(a.logits + (long)b * a.N * HW, image_bytes);
    const URsrc rsg = image_rsrc(MIX ? a.sigma + (long)b * a.N * HW : a.logits, MIX ? image_bytes : 0u);
    unsigned po = 0;
    RnAhead rn;
    rn.fetch(Rn, 0);
    for (int n = 0; n < a.N; ++n, po += plane_bytes) {
      const bool mk = rn.mask(u);
      rn.fetch(Rn, min(n + 1, a.N - 1));
      float l = 0.0f, s = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
      if (mk) {   // a masked plane samples as all-zero features (trainer.py:580)
        l = usample_k(rl, t, po);
        if (MIX) s = usample_k(rsg, t, po);
        c0 = s0; c1 = s1; c2 = s2;
      }
      if (render) {
        const bool last = (n == a.N - 1);
        const float dist = last ? 0.0f : a.dists[((long)b * (a.N - 1) + n) * HW + pix];
        mixture_accumulate<MIX>(acc, render_prob(rs, render_alpha(l, dist, last)), s, c0, c1, c2, t0, t1, t2, ea, automask);
      } else {
        fwd_accumulate<MIX>(acc, l, s, c0, c1, c2, t0, t1, t2, ea, automask);
      }
    }
    const FwdResult r = fwd_finish<MIX>(acc, t0, t1, t2, ea, automask, !render);
    float* st = stash + (long)b * a.stash_k * HW + pix;
    st[0] = r.lse2; st[HW] = r.Sn; st[2 * HW] = r.mx; st[3 * HW] = r.sel;
    rgb_rec[((long)b * 3 + 0) * HW + pix] = r.r0;
    rgb_rec[((long)b * 3 + 1) * HW + pix] = r.r1;
    rgb_rec[((long)b * 3 + 2) * HW + pix] = r.r2;
    ph_map[(long)b * HW + pix] = r.ph;
    ph_val = r.ph;
  }
  if (a.ph_mean) {
    __shared__ float wsum[kBlock / kWave];
    const float v = wave_sum(ph_val);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tt = 0.0f;
      for (int w = 0; w < kBlock / kWave; ++w) tt += wsum[w];
      unsafeAtomicAdd(a.ph_mean, tt * a.inv_numel);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, pass 1: per-plane gradients of the SAMPLED logit / sigma at every target pixel of image b
// ---------------------------------------------------------------------------------------------------------------
// Homography gradient.  `plane` is [B,4,3,3]: slice 0 is THE homography; slices 1..3 are the homographies of three
// virtual planes with n/d = e_0, e_1, e_2 (equal to slice 0 in value when t = 0).  Their gradients are defined as the
// plane sums weighted with tw[b][n][j] = n_n[j] / d_n:  g[0] = sum_n G_n,  g[1+j] = sum_n G_n tw[n][j].  Autograd through
// f(R + t e_j^T) then yields exactly dL/dt = sum_n Q_n n_n / d_n of the per-plane formulation (layers.py:216) although
// only one matrix per image is ever formed (pd_plane_sweep_uniform.hip header; ops.plane_sweep_homography).
constexpr int kUniG = 4;

template <bool MIX, bool RENDER = false>
__global__ __launch_bounds__(kBlock) void uniform_bwd_pass1_kernel(SweepArgs a, BwdOut o, int b0, float* __restrict__ tmp,
                                                                   float* __restrict__ partials, const float* __restrict__ tw,
                                                                   const int* __restrict__ run_flag) {
  if (run_flag && *run_flag == 0) return;   // the fused kernel served the whole launch
  __shared__ float red[kUniG * 9];
  const int HW = a.H * a.W, N = a.N;
  const int ntiles = gridDim.x;
  const int tile = xcd_banded(blockIdx.x, gridDim.x);
  const int by = blockIdx.y;
  const int pix = tile * kBlock + threadIdx.x;
  const int b = b0 + by;
  // (g_l, g_s) of a pixel-plane side by side: one 8-byte store here, one 8-byte load per list entry in pass 2 (pass 2
  // is paced by its number of memory instructions: 0.386 -> see DESIGN.md with the two tensors apart)
  float* __restrict__ tmp_b = tmp + (long)by * 2 * N * HW;
  if (threadIdx.x < kUniG * 9) red[threadIdx.x] = 0.0f;
  __syncthreads();
  const bool want_plane = (o.g_plane != nullptr);
  float gixw[kUniG], giyw[kUniG];
#pragma unroll
  for (int j = 0; j < kUniG; ++j) gixw[j] = giyw[j] = 0.0f;
  float gp_scale0 = 0.0f, p0 = 0.0f, p1 = 0.0f, fxx = 0.0f, fyy = 0.0f;
  bool zcl = false;
  if (pix < HW) {
    const int y = pix / a.W, x = pix - y * a.W;
    const float* srcb = a.src + (long)b * 3 * HW;
    const PixelCtx c = make_pixel_ctx<MIX>(a, o, b, pix, HW);
    const CoordNorm cn = make_coord_norm(a.W, a.H);
    const UniGeom u = uni_geom(a.plane + (long)b * kUniH, a.inv_K3 + (long)b * 9, cn, x, y);
    const TapK t = tap_kernel(make_tap(u.g.ix, u.g.iy, a.W, a.H), a.W, a.H);
    float d0x, d0y, d1x, d1y, d2x, d2y;
    const float s0 = sample_vg_k(srcb, t, d0x, d0y), s1 = sample_vg_k(srcb + HW, t, d1x, d1y),
                s2 = sample_vg_k(srcb + 2 * HW, t, d2x, d2y);
    const float* Rn = a.plane_aux + (long)b * N * 3;
    const float* twb = tw ? tw + (long)b * N * 3 : nullptr;
    constexpr bool render = RENDER;
    const float Rtot = MIX ? -c.A * c.mx : c.gdotr;   // sum_k p_k dL/dp_k in closed form
    float T = 1.0f, prefix = 0.0f;
    const unsigned image_bytes = (unsigned)(N * HW) * 4u, plane_bytes = (unsigned)HW * 4u;
    const URsrc rl = image_rsrc(a.logits + (long)b * N * HW, image_bytes);
    const URsrc rsg = image_rsrc(MIX ? a.sigma + (long)b * N * HW : a.logits, MIX ? image_bytes : 0u);
    const URsrc rtmp = image_rsrc(tmp_b, image_bytes * (MIX ? 2u : 1u));   // the scratch of this image: [N][HW] float2 / float
    const unsigned tmp_lane = (unsigned)pix * (MIX ? 8u : 4u), tmp_plane = plane_bytes * (MIX ? 2u : 1u);
    unsigned po = 0, pt = 0;
    RnAhead rn;
    rn.fetch(Rn, 0);
    for (int n = 0; n < N; ++n, po += plane_bytes, pt += tmp_plane) {
      float g_l = 0.0f, g_s = 0.0f;
      if (render && o.g_dists && n < N - 1) o.g_dists[((long)b * (N - 1) + n) * HW + pix] = 0.0f;   // (masked planes keep this)
      const bool mk = rn.mask(u);
      rn.fetch(Rn, min(n + 1, N - 1));
      if (mk) {
        float dlx = 0.0f, dly = 0.0f, dsx = 0.0f, dsy = 0.0f;
        float l, s = 0.0f;
        if (want_plane) {
          l = usample_vg_k(rl, t, po, dlx, dly);
          if (MIX) s = usample_vg_k(rsg, t, po, dsx, dsy);
        } else {
          l = usample_k(rl, t, po);
          if (MIX) s = usample_k(rsg, t, po);
        }
        PlaneGrad pg;
        if (render) {   // d prob_k / d alpha_n for k >= n through the transmittance (trainer.py:584-591)
          const bool last = (n == N - 1);
          const float dist = last ? 0.0f : a.dists[((long)b * (N - 1) + n) * HW + pix];
          const float alpha = render_alpha(l, dist, last);
          const float pn = alpha * T;
          pg = plane_grad_p<MIX>(c, pn, s, s0, s1, s2);
          prefix += pg.g_l * pn;
          const float keep = 1.0f - alpha + 1e-10f;
          const float g_alpha = pg.g_l * T - (Rtot - prefix) / keep;
          const float da = 1.0f - alpha;
          pg.g_l = (!last && l > 0.0f) ? g_alpha * dist * da : 0.0f;
          if (o.g_dists && !last) o.g_dists[((long)b * (N - 1) + n) * HW + pix] = g_alpha * fmaxf(l, 0.0f) * da;
          T *= keep;
        } else {
          pg = plane_grad<MIX>(c, l, s, s0, s1, s2);
        }
        g_l = pg.g_l; g_s = pg.g_s;
        if (want_plane) {
          const float gx = pg.g_l * dlx + pg.g_s * dsx + pg.gc0 * d0x + pg.gc1 * d1x + pg.gc2 * d2x;
          const float gy = pg.g_l * dly + pg.g_s * dsy + pg.gc0 * d0y + pg.gc1 * d1y + pg.gc2 * d2y;
          gixw[0] += gx; giyw[0] += gy;
          if (twb) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { const float w = twb[n * 3 + j]; gixw[1 + j] += gx * w; giyw[1 + j] += gy * w; }
          }
        }
      }
      if (MIX) {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        u2v v; v.x = __builtin_bit_cast(unsigned, g_l); v.y = __builtin_bit_cast(unsigned, g_s);
        __builtin_amdgcn_raw_buffer_store_b64(v, rtmp, (int)tmp_lane, (int)pt, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g_l), rtmp, (int)tmp_lane, (int)pt, 0);
      }
    }
    float inv_z = fast_rcp(u.g.zc);
    inv_z = fmaf(fmaf(-u.g.zc, inv_z, 1.0f), inv_z, inv_z);
    gp_scale0 = inv_z; p0 = u.g.p0; p1 = u.g.p1; zcl = u.g.z_clamped; fxx = (float)x; fyy = (float)y;
  }
  if (want_plane) {   // every lane takes part in the reductions (lanes past the image carry zeros)
    const float gscale_x = (float)(a.W - 1) / 2 * 2.0f / (float)(a.W - 1), gscale_y = (float)(a.H - 1) / 2 * 2.0f / (float)(a.H - 1);
    const int ln = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int j = 0; j < kUniG; ++j) {
      const float gp0 = gixw[j] * gscale_x * gp_scale0, gp1 = giyw[j] * gscale_y * gp_scale0;
      const float gz = zcl ? 0.0f : -(gp0 * p0 + gp1 * p1) * gp_scale0;
      const float gk[9] = {gp0 * fxx, gp0 * fyy, gp0, gp1 * fxx, gp1 * fyy, gp1, gz * fxx, gz * fyy, gz};
#pragma unroll
      for (int k = 0; k + 1 < 9; k += 2) {
        const float v = half_wave_sums_hi(gk[k], gk[k + 1]);
        if ((ln & 31) == 31) lds_add(&red[j * 9 + k + (ln >> 5)], v);
      }
      const float v8 = wave_sum_hi(gk[8]);
      if (ln == kWave - 1) lds_add(&red[j * 9 + 8], v8);
    }
    __syncthreads();
    if (threadIdx.x < kUniG * 9)
      partials[((long)b * ntiles + tile) * (kUniG * 9) + threadIdx.x] = red[threadIdx.x];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, pass 2: gather per SOURCE pixel
// ---------------------------------------------------------------------------------------------------------------
struct UniPrep {      // per image: forward homography (source -> target) from an fp64 adjugate
  float Hs[9];
  float ok;
  float pad[2];
};
__global__ void uniform_prep_kernel(const float* __restrict__ H_t2s, UniPrep* __restrict__ prep, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const float* h = H_t2s + (long)i * kUniH;
  const double a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], k = h[7], l = h[8];
  const double A = e * l - f * k, Bc = -(d * l - f * g), C = d * k - e * g;
  const double det = a * A + b * Bc + c * C, inv = 1.0 / det;
  UniPrep p;
  p.Hs[0] = (float)(A * inv);  p.Hs[1] = (float)(-(b * l - c * k) * inv); p.Hs[2] = (float)((b * f - c * e) * inv);
  p.Hs[3] = (float)(Bc * inv); p.Hs[4] = (float)((a * l - c * g) * inv);  p.Hs[5] = (float)(-(a * f - c * d) * inv);
  p.Hs[6] = (float)(C * inv);  p.Hs[7] = (float)(-(a * k - b * g) * inv); p.Hs[8] = (float)((a * e - b * d) * inv);
  bool ok = (det == det) && fabs(det) > 1e-30 && fabs(inv) < 1e30;
  for (int j = 0; j < 9; ++j) ok = ok && (fabsf(p.Hs[j]) < 1e30f) && (p.Hs[j] == p.Hs[j]);
  p.ok = ok ? 1.0f : 0.0f;
  p.pad[0] = p.pad[1] = 0.0f;   // pad[0] of image 0 doubles as the overflow flag (an int 0)
  prep[i] = p;
}

// A sample reaches source pixel s iff it lies in the open square s +- 1; the square is grown by 1/32 pixel before it is
// mapped and the mapped box by another 1/32: the forward chain's fp32 noise and the fp64-adjugate inverse's error are both
// below 1e-3 pixel, and every candidate is confirmed with the exact forward coordinates anyway.
constexpr float kUniReach = 1.03125f, kUniSlop = 0.03125f;

// Candidate window of source pixel (sx, sy) in the target view: bounding box of the pre-image of (sx-1, sx+1) x (sy-1, sy+1)
struct UniWindow { int x0, x1, y0, y1; };
__device__ __forceinline__ UniWindow uni_window(const UniPrep& p, int sx, int sy, int W, int H) {
  UniWindow w;
  w.x0 = 0; w.y0 = 0; w.x1 = W - 1; w.y1 = H - 1;
  // A homography that is not finite / not invertible (a diverged pose net) has no meaningful adjoint: give such an image an
  // EMPTY window (its gradients are garbage either way) instead of the whole-image scan, which at 192 x 640 would keep the
  // device busy for seconds per image.  The whole-image fallback below is for valid maps whose line at infinity crosses
  // the tile (rotations beyond ~50 degrees at KITTI's field of view: nothing a pose net scaled by 0.01 produces).
  if (p.ok == 0.0f) { w.x1 = -1; w.y1 = -1; return w; }
  float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f, wmin = 3.0e38f, wmax = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float x = (float)sx + ((k & 1) ? kUniReach : -kUniReach), y = (float)sy + ((k & 2) ? kUniReach : -kUniReach);
    const float uu = p.Hs[0] * x + p.Hs[1] * y + p.Hs[2], vv = p.Hs[3] * x + p.Hs[4] * y + p.Hs[5];
    const float ww = p.Hs[6] * x + p.Hs[7] * y + p.Hs[8];
    wmin = fminf(wmin, ww); wmax = fmaxf(wmax, ww);
    const float r = 1.0f / ww;
    xmin = fminf(xmin, uu * r); xmax = fmaxf(xmax, uu * r);
    ymin = fminf(ymin, vv * r); ymax = fmaxf(ymax, vv * r);
  }
  const float wabs = fmaxf(fabsf(wmin), fabsf(wmax));
  if (!(wmin * wmax > 0.0f) || !(fminf(fabsf(wmin), fabsf(wmax)) > 1e-3f * wabs)) return w;   // near the line at infinity
  if (!(xmin == xmin) || !(xmax == xmax) || !(ymin == ymin) || !(ymax == ymax)) return w;
  w.x0 = (int)fminf(fmaxf(floorf(xmin - kUniSlop), 0.0f), (float)W);
  w.y0 = (int)fminf(fmaxf(floorf(ymin - kUniSlop), 0.0f), (float)H);
  w.x1 = (int)fminf(fmaxf(ceilf(xmax + kUniSlop), -1.0f), (float)(W - 1));
  w.y1 = (int)fminf(fmaxf(ceilf(ymax + kUniSlop), -1.0f), (float)(H - 1));
  return w;
}

// OVERFLOW = false: the main kernel; source pixels with more than kUniK contributors are left alone.
// OVERFLOW = true:  the follow-up kernel; it repeats the scan, returns at once for everybody else and re-scans the
//                   candidates per plane for those pixels (strong minification only).  Kept out of the main kernel: with the
//                   re-scan loop inside it the main kernel ran 0.99 instead of 0.58 ms at 8x49x192x640.
struct PairLS { float l, s; };
template <bool MIX>
__device__ __forceinline__ PairLS load_ls(const float* __restrict__ tmp_b, long i) {
  PairLS r;
  if (MIX) {
    const float2 v = reinterpret_cast<const float2*>(tmp_b)[i];
    r.l = v.x; r.s = v.y;
  } else {
    r.l = tmp_b[i]; r.s = 0.0f;
  }
  return r;
}

template <bool MIX, bool OVERFLOW>
__global__ __launch_bounds__(kBlock) void uniform_bwd_pass2_kernel(SweepArgs a, int b0, const float* __restrict__ tmp,
                                                                   const UniPrep* __restrict__ prep,
                                                                   float* __restrict__ g_logits, float* __restrict__ g_sigma,
                                                                   int* __restrict__ overflow_flag, const int* __restrict__ run_flag,
                                                                   int accumulate, int list_limit = kUniK) {
  // list_limit (OVERFLOW only): the register slots of the kernel this one follows (kUniK, or kPairK after the pair kernel)
  if (run_flag && *run_flag == 0) return;        // the fused kernel served the whole launch
  const int HW = a.H * a.W, N = a.N, W = a.W, H = a.H;
  const int spix = xcd_banded(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  const int b = b0 + blockIdx.y;
  if (OVERFLOW && *overflow_flag == 0) return;   // nobody asked for the re-scan: the usual case, no second scan
  const float* __restrict__ tmp_b = tmp + (long)blockIdx.y * 2 * N * HW;
  if (spix >= HW) return;
  const int sy = spix / W, sx = spix - sy * W;
  const CoordNorm cn = make_coord_norm(W, H);
  const float* Hm = a.plane + (long)b * kUniH;
  const float* Ki = a.inv_K3 + (long)b * 9;
  const UniWindow win = uni_window(prep[b], sx, sy, W, H);
  int idx[kUniK];
  float wgt[kUniK];
#pragma unroll
  for (int k = 0; k < kUniK; ++k) { idx[k] = 0; wgt[k] = 0.0f; }
  int cnt = 0;
  for (int ty = win.y0; ty <= win.y1; ++ty)
    for (int tx = win.x0; tx <= win.x1; ++tx) {
      const UniGeom u = uni_geom(Hm, Ki, cn, tx, ty);
      const float w = tap_weight_on(u.g.ix, u.g.iy, sx, sy);
      if (w != 0.0f) {
        // (static indices only: a dynamic index would move the arrays to scratch memory)
#pragma unroll
        for (int k = 0; k < kUniK; ++k)
          if (k == cnt) { idx[k] = ty * W + tx; wgt[k] = w; }
        ++cnt;
      }
    }
  float* gl = g_logits ? g_logits + (long)b * N * HW + spix : nullptr;
  float* gs = (MIX && g_sigma) ? g_sigma + (long)b * N * HW + spix : nullptr;
  if (OVERFLOW) {
    if (cnt <= list_limit) return;
    for (int n = 0; n < N; ++n) {
      float accl = 0.0f, accs = 0.0f;
      for (int ty = win.y0; ty <= win.y1; ++ty)
        for (int tx = win.x0; tx <= win.x1; ++tx) {
          const UniGeom u = uni_geom(Hm, Ki, cn, tx, ty);
          const float w = tap_weight_on(u.g.ix, u.g.iy, sx, sy);
          if (w != 0.0f) {
            const PairLS v = load_ls<MIX>(tmp_b, (long)n * HW + ty * W + tx);
            accl += w * v.l;
            accs += w * v.s;
          }
        }
      if (accumulate) {   // PD_BWD_ACCUMULATE: another target view's gradient is already there
        if (gl) accl += gl[(long)n * HW];
        if (gs) accs += gs[(long)n * HW];
      }
      if (gl) gl[(long)n * HW] = accl;
      if (gs) gs[(long)n * HW] = accs;
    }
    return;
  }
  if (cnt > kUniK) { atomicOr(overflow_flag, 1); return; }   // the follow-up kernel's
  // Four entries for everybody (a near-isometric map gives four contributors), the rest under a wave-uniform bound:
  // kmax = the largest list in the wave, so a wave pays for its longest lane only.  (Unused entries: weight 0, index 0.)
  int kmax = cnt;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, kWave));
  kmax = __builtin_amdgcn_readfirstlane(kmax);
#pragma unroll 2
  for (int n = 0; n < N; ++n) {
    const long base = (long)n * HW;
    float accl = 0.0f, accs = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const PairLS v = load_ls<MIX>(tmp_b, base + idx[k]);
      accl += wgt[k] * v.l;
      accs += wgt[k] * v.s;
    }
#pragma unroll
    for (int k = 4; k < kUniK; ++k) {
      if (k < kmax) {   // wave-uniform
        const PairLS v = load_ls<MIX>(tmp_b, base + idx[k]);
        accl += wgt[k] * v.l;
        accs += wgt[k] * v.s;
      }
    }
    if (accumulate) {
      if (gl) accl += gl[(long)n * HW];
      if (gs) accs += gs[(long)n * HW];
    }
    if (gl) gl[(long)n * HW] = accl;
    if (gs) gs[(long)n * HW] = accs;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, pass 2 with the scratch staged through LDS
// ---------------------------------------------------------------------------------------------------------------
// The gather above issues one 8-byte load per list entry, plane and lane (4-6 of them) and is paced by that instruction
// count.  Neighbouring source pixels share most of their contributors, so here a workgroup owns a 32 x 16 tile of source
// pixels, copies the tile's pre-image box of the scratch (~35 x 20 target pixels for pose_net rotations) to LDS with
// coalesced loads — kStageP planes at a time, the next group's loads in flight while the current one is reduced — and
// the list entries become ds_read_b64s.  A box that does not fit kStageBox pixels (strong rotation / minification)
// makes the workgroup take the direct gather of the kernel above (workgroup-uniform branch, same lists).
// Measured at 8x49x192x640: 0.329 -> 0.215 ms.  The same staging for the TARGET-side kernels (forward, pass 1: logits and
// sigma boxes in LDS, taps as ds_reads) was built and measured SLOWER — forward 0.140 -> 0.199 ms, pass 1 0.188 ->
// 0.277 ms: those kernels are bound by their arithmetic, and eight ds_reads + a barrier per plane pair come on top — so
// they keep their direct gathers.
#ifndef PD_STAGE_P
#define PD_STAGE_P 2
#endif
#ifndef PD_STAGE_H
#define PD_STAGE_H 16   // 32 x 16 source tiles, 512 threads: the tile's pre-image box holds 1.37x the tile's scratch elements, a 32 x 8
#endif                 // tile's 1.64x — and that overlap is re-read through the L1s (pass 2 0.215 -> 0.185 ms, pair 0.420 -> 0.407)
constexpr int kStageW = 32, kStageH = PD_STAGE_H, kStageP = PD_STAGE_P, kStageBox = 1024;
constexpr int kStageThreads = kStageW * kStageH;   // one thread per source pixel of the tile
constexpr int kStagePre = (kStageBox * kStageP + kStageThreads - 1) / kStageThreads;   // staged elements per thread and group

template <bool MIX>
__global__ __launch_bounds__(kStageThreads) void uniform_bwd_pass2_staged_kernel(SweepArgs a, int b0, const float* __restrict__ tmp,
                                                                          const UniPrep* __restrict__ prep,
                                                                          float* __restrict__ g_logits, float* __restrict__ g_sigma,
                                                                          int* __restrict__ overflow_flag, int tiles_x, int accumulate) {
  typedef typename std::conditional<MIX, float2, float>::type Elem;
  __shared__ Elem buf[2][kStageP][kStageBox];
  const int HW = a.H * a.W, N = a.N, W = a.W, H = a.H;
  const int by = blockIdx.y;
  const int b = b0 + by, tid = threadIdx.x;
  const int blk = xcd_banded(blockIdx.x, gridDim.x);
  const int tyi = blk / tiles_x, txi = blk - tyi * tiles_x;
  const int xs0 = txi * kStageW, ys0 = tyi * kStageH;
  const int tw_ = min(kStageW, W - xs0), th = min(kStageH, H - ys0);
  const Elem* __restrict__ tmp_b = reinterpret_cast<const Elem*>(tmp + (long)by * 2 * N * HW);
  const CoordNorm cn = make_coord_norm(W, H);
  const float* Hm = a.plane + (long)b * kUniH;
  const float* Ki = a.inv_K3 + (long)b * 9;
  const UniPrep pr = prep[b];
  int bx0, bx1, by0, by1;   // box of the tile: union of its corner pixels' windows (the pre-image of a convex region is convex)
  {
    const UniWindow w0 = uni_window(pr, xs0, ys0, W, H), w1 = uni_window(pr, xs0 + tw_ - 1, ys0 + th - 1, W, H);
    const UniWindow w2 = uni_window(pr, xs0 + tw_ - 1, ys0, W, H), w3 = uni_window(pr, xs0, ys0 + th - 1, W, H);
    bx0 = min(min(w0.x0, w1.x0), min(w2.x0, w3.x0)); bx1 = max(max(w0.x1, w1.x1), max(w2.x1, w3.x1));
    by0 = min(min(w0.y0, w1.y0), min(w2.y0, w3.y0)); by1 = max(max(w0.y1, w1.y1), max(w2.y1, w3.y1));
  }
  const int bw = max(bx1 - bx0 + 1, 0), bh = max(by1 - by0 + 1, 0);
  const int npx = bw * bh;
  const bool staged = npx <= kStageBox;          // workgroup-uniform
  const int lx = tid & (kStageW - 1), ly = tid >> 5;
  const bool has_s = lx < tw_ && ly < th;
  const int sx = xs0 + lx, sy = ys0 + ly;
  int idx[kUniK];
  float wgt[kUniK];
#pragma unroll
  for (int k = 0; k < kUniK; ++k) { idx[k] = 0; wgt[k] = 0.0f; }
  int cnt = 0;
  if (has_s) {
    const UniWindow win = uni_window(pr, sx, sy, W, H);
    for (int ty = win.y0; ty <= win.y1; ++ty)
      for (int tx = win.x0; tx <= win.x1; ++tx) {
        const UniGeom u = uni_geom(Hm, Ki, cn, tx, ty);
        const float w = tap_weight_on(u.g.ix, u.g.iy, sx, sy);
        if (w != 0.0f) {
          // a confirmed contributor lies in the box by construction; the clamp only guards the LDS index
          const int slot = staged ? min(max((ty - by0) * bw + (tx - bx0), 0), kStageBox - 1) : ty * W + tx;
#pragma unroll
          for (int k = 0; k < kUniK; ++k)
            if (k == cnt) { idx[k] = slot; wgt[k] = w; }
          ++cnt;
        }
      }
  }
  bool live = has_s;
  if (cnt > kUniK) { atomicOr(overflow_flag, 1); live = false; cnt = 0; }   // the follow-up kernel's pixel
  int kmax = cnt;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, kWave));
  kmax = __builtin_amdgcn_readfirstlane(kmax);
  // Entries a lane does not use have weight 0 and slot 0.  Slot 0 of the staged box holds a real value as soon as ANY
  // lane of the workgroup has a contributor (contributors lie in the box, so the box is not empty); a wave without a
  // single contributor must not read LDS at all — the box may be empty and the buffer uninitialised, and 0 * garbage is
  // NaN when the garbage is (seen once in ~40 runs of the minification test on fresh boxes).
  const int kuse = kmax == 0 ? 0 : max(kmax, 4);   // four entries for everybody (the regular case), the rest wave-bounded
  float* gl = (live && g_logits) ? g_logits + (long)b * N * HW + (long)sy * W + sx : nullptr;
  float* gs = (live && MIX && g_sigma) ? g_sigma + (long)b * N * HW + (long)sy * W + sx : nullptr;

  if (!staged) {   // the direct gather (idx = absolute pixel index)
    for (int n = 0; n < N; ++n) {
      const long base = (long)n * HW;
      float accl = 0.0f, accs = 0.0f;
#pragma unroll
      for (int k = 0; k < kUniK; ++k) {
        if (k < kmax) {
          const PairLS v = load_ls<MIX>(reinterpret_cast<const float*>(tmp_b), base + idx[k]);
          accl += wgt[k] * v.l;
          accs += wgt[k] * v.s;
        }
      }
      if (accumulate) {
        if (gl) accl += gl[base];
        if (gs) accs += gs[base];
      }
      if (gl) gl[base] = accl;
      if (gs) gs[base] = accs;
    }
    return;
  }

  // staging map: element e of a group = (plane p, box slot q); thread t copies e = t, t + 256, ...  The map is the same for
  // every group, so its divisions are done once.
  const int per_group = npx * kStageP;
  int e_pl[kStagePre], e_rc[kStagePre], e_q[kStagePre];   // plane within the group (-1: none), pixel offset, box slot
#pragma unroll
  for (int j = 0; j < kStagePre; ++j) {
    const int e = tid + j * kStageThreads;
    const int pp = e / max(npx, 1), q = e - pp * npx;
    const int ry = q / max(bw, 1), rx = q - ry * bw;
    e_pl[j] = e < per_group ? pp : -1;
    e_rc[j] = (by0 + ry) * W + (bx0 + rx);
    e_q[j] = q;
  }
  Elem pre[kStagePre];
  auto issue = [&](int n0) {
#pragma unroll
    for (int j = 0; j < kStagePre; ++j) {
      Elem v = Elem();
      if (e_pl[j] >= 0) v = tmp_b[(long)min(n0 + e_pl[j], N - 1) * HW + e_rc[j]];
      pre[j] = v;
    }
  };
  auto park = [&](int which) {
#pragma unroll
    for (int j = 0; j < kStagePre; ++j)
      if (e_pl[j] >= 0) buf[which][e_pl[j]][e_q[j]] = pre[j];
  };
  // PD_BWD_ACCUMULATE: the values already in g_logits / g_sigma are fetched one group ahead, with the staging loads (read
  // at the point of use they put a full memory round trip between every plane's gather and its store: 0.218 -> 0.405 ms)
  float nextl[kStageP], nexts[kStageP];
  auto fetch_old = [&](int n0) {
#pragma unroll
    for (int p = 0; p < kStageP; ++p) {
      nextl[p] = 0.0f; nexts[p] = 0.0f;
      if (accumulate && n0 + p < N) {
        if (gl) nextl[p] = gl[(long)(n0 + p) * HW];
        if (gs) nexts[p] = gs[(long)(n0 + p) * HW];
      }
    }
  };
  issue(0);
  fetch_old(0);
  park(0);
  int which = 0;
  for (int n0 = 0; n0 < N; n0 += kStageP, which ^= 1) {
    __syncthreads();                               // buf[which] complete; buf[which ^ 1] no longer read by anybody
    const bool more = n0 + kStageP < N;
    float oldl[kStageP], olds[kStageP];
#pragma unroll
    for (int p = 0; p < kStageP; ++p) { oldl[p] = nextl[p]; olds[p] = nexts[p]; }
    if (more) {                                    // in flight while this group is reduced
      issue(n0 + kStageP);
      fetch_old(n0 + kStageP);
    }
#pragma unroll
    for (int p = 0; p < kStageP; ++p) {
      const int n = n0 + p;
      if (n < N) {
        const long base = (long)n * HW;
        float accl = oldl[p], accs = olds[p];
#pragma unroll
        for (int k = 0; k < kUniK; ++k) {
          if (k < kuse) {
            const Elem v = buf[which][p][idx[k]];
            if constexpr (MIX) { accl += wgt[k] * v.x; accs += wgt[k] * v.y; }
            else accl += wgt[k] * v;
          }
        }
        if (gl) gl[base] = accl;
        if (gs) gs[base] = accs;
      }
    }
    if (more) park(which ^ 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, pass 2 of TWO target views in one kernel (pd_uniform_gather_pair)
// ---------------------------------------------------------------------------------------------------------------
// The reference's mono training sweeps two novel frames (-1, +1) over the same logits / sigma (trainer.py:532), so their
// gradients land in the same tensors: as two passes the second one reads what the first one wrote (385 MB at
// 8x49x192x640) and both write the full tensors.  Here one workgroup owns the 32 x 16 source tile for BOTH views: two
// gather lists per source pixel, the two scratch boxes of a plane staged side by side (one plane per step and view, so
// the LDS footprint stays that of the single-view kernel), one store per gradient element.  A source pixel whose list
// overflows in a view takes nothing from that view here; that view's follow-up kernel (uniform_bwd_pass2_kernel<.., true>,
// accumulate) adds it.  Boxes that do not fit make the workgroup gather both views directly.
constexpr int kPairViews = 2, kPairK = 10;      // list entries kept per view (12 in the single-view kernel: 142 VGPRs here)
constexpr int kPairPre = kStageBox / kStageThreads;    // staged elements per thread, view and plane
struct PairArgs {
  int N, H, W;
  const float* plane[kPairViews];
  const float* inv_K3[kPairViews];
  const float* tmp[kPairViews];
  const UniPrep* prep[kPairViews];
  int* overflow[kPairViews];
};

template <bool MIX>
__global__ __launch_bounds__(kStageThreads) void uniform_bwd_pass2_pair_kernel(PairArgs pa, float* __restrict__ g_logits,
                                                                        float* __restrict__ g_sigma, int tiles_x, int accumulate) {
  typedef typename std::conditional<MIX, float2, float>::type Elem;
  __shared__ Elem buf[2][kPairViews][kStageBox];
  const int H = pa.H, W = pa.W, N = pa.N, HW = H * W;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int blk = xcd_banded(blockIdx.x, gridDim.x);
  const int tyi = blk / tiles_x, txi = blk - tyi * tiles_x;
  const int xs0 = txi * kStageW, ys0 = tyi * kStageH;
  const int tw_ = min(kStageW, W - xs0), th = min(kStageH, H - ys0);
  const CoordNorm cn = make_coord_norm(W, H);
  const int lx = tid & (kStageW - 1), ly = tid >> 5;
  const bool has_s = lx < tw_ && ly < th;
  const int sx = xs0 + lx, sy = ys0 + ly;
  int bx0[kPairViews], by0[kPairViews], bw[kPairViews], npx[kPairViews];
  int idx[kPairViews][kPairK];
  float wgt[kPairViews][kPairK];
  int kuse[kPairViews];
  const Elem* tmp_b[kPairViews];
  bool staged = true;
#pragma unroll
  for (int v = 0; v < kPairViews; ++v) {
    const UniPrep pr = pa.prep[v][b];
    const UniWindow w0 = uni_window(pr, xs0, ys0, W, H), w1 = uni_window(pr, xs0 + tw_ - 1, ys0 + th - 1, W, H);
    const UniWindow w2 = uni_window(pr, xs0 + tw_ - 1, ys0, W, H), w3 = uni_window(pr, xs0, ys0 + th - 1, W, H);
    bx0[v] = min(min(w0.x0, w1.x0), min(w2.x0, w3.x0));
    by0[v] = min(min(w0.y0, w1.y0), min(w2.y0, w3.y0));
    const int bx1 = max(max(w0.x1, w1.x1), max(w2.x1, w3.x1)), by1 = max(max(w0.y1, w1.y1), max(w2.y1, w3.y1));
    bw[v] = max(bx1 - bx0[v] + 1, 0);
    npx[v] = bw[v] * max(by1 - by0[v] + 1, 0);
    staged = staged && npx[v] <= kStageBox;
    tmp_b[v] = reinterpret_cast<const Elem*>(pa.tmp[v] + (long)b * 2 * N * HW);
  }
#pragma unroll
  for (int v = 0; v < kPairViews; ++v) {
#pragma unroll
    for (int k = 0; k < kPairK; ++k) { idx[v][k] = 0; wgt[v][k] = 0.0f; }
    int cnt = 0;
    if (has_s) {
      const float* Hm = pa.plane[v] + (long)b * kUniH;
      const float* Ki = pa.inv_K3[v] + (long)b * 9;
      const UniWindow win = uni_window(pa.prep[v][b], sx, sy, W, H);
      for (int ty = win.y0; ty <= win.y1; ++ty)
        for (int tx = win.x0; tx <= win.x1; ++tx) {
          const UniGeom u = uni_geom(Hm, Ki, cn, tx, ty);
          const float w = tap_weight_on(u.g.ix, u.g.iy, sx, sy);
          if (w != 0.0f) {
            const int slot = staged ? min(max((ty - by0[v]) * bw[v] + (tx - bx0[v]), 0), kStageBox - 1) : ty * W + tx;
#pragma unroll
            for (int k = 0; k < kPairK; ++k)
              if (k == cnt) { idx[v][k] = slot; wgt[v][k] = w; }
            ++cnt;
          }
        }
    }
    if (cnt > kPairK) { atomicOr(pa.overflow[v], 1); cnt = 0;   // this view's follow-up kernel adds the pixel's share
#pragma unroll
      for (int k = 0; k < kPairK; ++k) { idx[v][k] = 0; wgt[v][k] = 0.0f; } }
    int kmax = cnt;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, kWave));
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    kuse[v] = kmax == 0 ? 0 : max(kmax, 4);   // (a wave without contributors must not read the possibly empty box)
  }
  const long s_off = (long)b * N * HW + (long)sy * W + sx;
  float* gl = (has_s && g_logits) ? g_logits + s_off : nullptr;
  float* gs = (has_s && MIX && g_sigma) ? g_sigma + s_off : nullptr;

  if (!staged) {   // direct gathers (idx = absolute pixel index)
    for (int n = 0; n < N; ++n) {
      const long base = (long)n * HW;
      float accl = 0.0f, accs = 0.0f;
      if (accumulate) {
        if (gl) accl = gl[base];
        if (gs) accs = gs[base];
      }
#pragma unroll
      for (int v = 0; v < kPairViews; ++v)
#pragma unroll
        for (int k = 0; k < kPairK; ++k)
          if (k < kuse[v]) {
            const PairLS e = load_ls<MIX>(reinterpret_cast<const float*>(tmp_b[v]), base + idx[v][k]);
            accl += wgt[v][k] * e.l;
            accs += wgt[v][k] * e.s;
          }
      if (gl) gl[base] = accl;
      if (gs) gs[base] = accs;
    }
    return;
  }

  // staging map (the same for every plane): element e = tid, tid + 256, ... of each view's box
  int e_rc[kPairViews][kPairPre];
#pragma unroll
  for (int v = 0; v < kPairViews; ++v)
#pragma unroll
    for (int j = 0; j < kPairPre; ++j) {
      const int q = tid + j * kStageThreads;
      const int ry = q / max(bw[v], 1), rx = q - ry * bw[v];
      e_rc[v][j] = q < npx[v] ? (by0[v] + ry) * W + (bx0[v] + rx) : -1;
    }
  // (Requesting the loads two steps ahead instead of one — two register sets, 145 VGPRs — made the kernel slower, 0.43 ->
  // 0.49 ms: it is not waiting for latency but moving 1.64x the scratch through the L1s, the boxes' overlap.)
  Elem pre[kPairViews][kPairPre];
  auto issue = [&](int n) {
#pragma unroll
    for (int v = 0; v < kPairViews; ++v)
#pragma unroll
      for (int j = 0; j < kPairPre; ++j) {
        Elem e = Elem();
        if (e_rc[v][j] >= 0) e = tmp_b[v][(long)n * HW + e_rc[v][j]];
        pre[v][j] = e;
      }
  };
  auto park = [&](int which) {
#pragma unroll
    for (int v = 0; v < kPairViews; ++v)
#pragma unroll
      for (int j = 0; j < kPairPre; ++j)
        if (e_rc[v][j] >= 0) buf[which][v][tid + j * kStageThreads] = pre[v][j];
  };
  float nextl = 0.0f, nexts = 0.0f;
  auto fetch_old = [&](int n) {
    nextl = 0.0f; nexts = 0.0f;
    if (accumulate) {
      if (gl) nextl = gl[(long)n * HW];
      if (gs) nexts = gs[(long)n * HW];
    }
  };
  issue(0);
  fetch_old(0);
  park(0);
  for (int n = 0; n < N; ++n) {
    const int which = n & 1;
    __syncthreads();                               // buf[which] complete; buf[which ^ 1] no longer read by anybody
    float accl = nextl, accs = nexts;
    const bool more = n + 1 < N;
    if (more) {                                    // in flight while this plane is reduced
      issue(n + 1);
      fetch_old(n + 1);
    }
#pragma unroll
    for (int v = 0; v < kPairViews; ++v)
#pragma unroll
      for (int k = 0; k < kPairK; ++k)
        if (k < kuse[v]) {
          const Elem e = buf[which][v][idx[v][k]];
          if constexpr (MIX) { accl += wgt[v][k] * e.x; accs += wgt[v][k] * e.y; }
          else accl += wgt[v][k] * e;
        }
    const long base = (long)n * HW;
    if (gl) gl[base] = accl;
    if (gs) gs[base] = accs;
    if (more) park(which ^ 1);
  }
}

#ifdef PD_EXPERIMENTS   // the one-kernel form (LDS hand-over, no scratch): measured slower; lives in scripts/experiments/
#include "pd_plane_sweep_uniform_fused.inc"
#endif


// partial sums of the kernels that ran: the fused kernel's unless it raised the flag
__global__ void uniform_reduce_kernel(const float* __restrict__ part_fused, int nblk_fused, const float* __restrict__ part_two,
                                      int nblk_two, const int* __restrict__ irregular_flag, float* __restrict__ out, int M) {
  const int j = blockIdx.x, b = blockIdx.y;
  const bool two = (*irregular_flag != 0) || part_fused == nullptr;
  const float* p = (two ? part_two + (long)b * nblk_two * M : part_fused + (long)b * nblk_fused * M) + j;
  const int nblk = two ? nblk_two : nblk_fused;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < nblk; i += kWave) acc += p[(long)i * M];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * M + j] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static size_t ualign4(size_t floats) { return (floats + 3) & ~(size_t)3; }

// images per launch of the two backward passes: enough workgroups to fill the chip, a scratch that still fits the
// 256 MB memory-side cache (2 x N x H x W floats per image)
static int uniform_chunk(const pd_sweep_desc* d) {
  if (const int c = switches().uni_chunk) return c < d->B ? c : d->B;   // PD_UNI_CHUNK, read once
  const size_t per_image = (size_t)2 * d->N * d->H * d->W * sizeof(float);
  (void)per_image;   // measured at 8x49x192x640: 1 / 2 / 4 / 8 images per launch -> 2.16 / 1.83 / 1.42 / 0.99 ms: parallelism
  return d->B;        // beats cache residency, so the whole batch goes in one launch per pass
}

// workspace: 2 x [B][nblk][4*9] partial sums (two-pass / fused) | UniPrep[B] | scratch [chunk][2][N][H][W]
size_t uniform_bwd_workspace_floats(const pd_sweep_desc* d) {
  const size_t nblk = (size_t)ceil_div(d->H * d->W, kBlock);   // (>= the fused kernel's tile count: 256 <= 13 * 60)
  return 2 * ualign4((size_t)d->B * nblk * kUniG * 9) + ualign4((size_t)d->B * (sizeof(UniPrep) / sizeof(float))) +
         (size_t)uniform_chunk(d) * 2 * d->N * d->H * d->W + 8;
}

template <bool PAIR>
static int uniform_fwd_launch(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash,
                              const UniViewB& vb, hipStream_t stream) {
  dim3 grid((PAIR ? 2 : 1) * ceil_div(d->H * d->W, kBlock), d->B);
  const bool mix = (d->flags & PD_MIXTURE) != 0, render = (d->flags & PD_RENDER_PROB) != 0;
  if (mix) { if (render) uniform_fwd_kernel<true, true, PAIR><<<grid, kBlock, 0, stream>>>(a, rgb_rec, ph_map, stash, vb);
             else        uniform_fwd_kernel<true, false, PAIR><<<grid, kBlock, 0, stream>>>(a, rgb_rec, ph_map, stash, vb); }
  else     { if (render) uniform_fwd_kernel<false, true, PAIR><<<grid, kBlock, 0, stream>>>(a, rgb_rec, ph_map, stash, vb);
             else        uniform_fwd_kernel<false, false, PAIR><<<grid, kBlock, 0, stream>>>(a, rgb_rec, ph_map, stash, vb); }
  return check_launch("uniform_fwd_kernel");
}
int uniform_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash, hipStream_t stream) {
  return uniform_fwd_launch<false>(d, a, rgb_rec, ph_map, stash, UniViewB{}, stream);
}
// two views of the same src / logits / sigma in one launch (pd_uniform_fwd_pair)
int uniform_fwd_pair(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash,
                     const SweepArgs& b, float* rgb_rec_b, float* ph_map_b, float* stash_b, hipStream_t stream) {
  UniViewB vb{b.tgt, b.plane, b.plane_aux, b.inv_K3, b.dists, b.ph_mean, rgb_rec_b, ph_map_b, stash_b};
  return uniform_fwd_launch<true>(d, a, rgb_rec, ph_map, stash, vb, stream);
}

// where the backward keeps its pieces inside a workspace
struct UniformWs { float* part_two; float* part_fused; UniPrep* prep; float* tmp; int* overflow; int* irregular; };
static UniformWs uniform_ws(const pd_sweep_desc* d, float* workspace) {
  const int nblk = ceil_div(d->H * d->W, kBlock);
  const uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15;
  UniformWs w;
  w.part_two = reinterpret_cast<float*>(base);
  w.part_fused = w.part_two + ualign4((size_t)d->B * nblk * kUniG * 9);
  w.prep = reinterpret_cast<UniPrep*>(w.part_fused + ualign4((size_t)d->B * nblk * kUniG * 9));
  w.tmp = reinterpret_cast<float*>(w.prep) + ualign4((size_t)d->B * (sizeof(UniPrep) / sizeof(float)));
  w.overflow = reinterpret_cast<int*>(&w.prep[0].pad[0]);    // both written 0 by uniform_prep_kernel
  w.irregular = reinterpret_cast<int*>(&w.prep[0].pad[1]);
  return w;
}

// `tw` = a.padding_mask slot: [B][N][3] translation weights n/d, or NULL (no translation gradient wanted)
int uniform_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, float* workspace, hipStream_t stream) {
  const int HW = d->H * d->W, nblk = ceil_div(HW, kBlock);
  const bool mix = (d->flags & PD_MIXTURE) != 0;
  const UniformWs w = uniform_ws(d, workspace);
  float* part_two = w.part_two;
  float* part_fused = w.part_fused;
  UniPrep* prep = w.prep;
  float* tmp = w.tmp;
  const float* tw = a.padding_mask;
  SweepArgs ak = a;
  ak.padding_mask = nullptr;
  uniform_prep_kernel<<<ceil_div(d->B, 64), 64, 0, stream>>>(a.plane, prep, d->B);
  int rc = check_launch("uniform_prep_kernel");
  int* overflow = w.overflow;
  int* irregular = w.irregular;
  (void)part_fused;
  // Measured at 8x49x192x640 (pose_net-like rotations): two-pass 0.27 + 0.38 = 0.65 ms, fused 0.75 ms — sixteen waves
  // meeting at a barrier 49 times cost more than the 770 MB the scratch tensor moves.  The fused kernel stays opt-in.
  const int accumulate = (d->flags & PD_BWD_ACCUMULATE) ? 1 : 0;
  const bool render = (d->flags & PD_RENDER_PROB) != 0;
  const bool defer = (d->flags & PD_BWD_DEFER_GATHER) != 0;   // pass 2 is pd_uniform_gather_pair's
#ifdef PD_EXPERIMENTS
  const bool fused = getenv("PD_UNI_FUSED") != nullptr && !accumulate && !render && !defer;   // experiments build only
#else
  const bool fused = false;
#endif
#ifdef PD_EXPERIMENTS
  const int tiles_x = ceil_div(d->W, kFuseC), ntiles = tiles_x * ceil_div(d->H, kFuseR);
#else
  const int ntiles = 0;
#endif
#ifdef PD_EXPERIMENTS
  if (!rc && fused) {   // regular case: per-plane gradients handed over through LDS
    dim3 grid(ntiles, d->B);
    if (mix) uniform_bwd_fused_kernel<true><<<grid, kFuseThreads, 0, stream>>>(ak, o, prep, part_fused, tw, tiles_x, irregular);
    else     uniform_bwd_fused_kernel<false><<<grid, kFuseThreads, 0, stream>>>(ak, o, prep, part_fused, tw, tiles_x, irregular);
    rc = check_launch("uniform_bwd_fused_kernel");
  }
#endif
  const int* run_flag = fused ? irregular : nullptr;   // the two-pass kernels return at once unless the fused one gave up
  const int chunk = uniform_chunk(d);
  if (defer && chunk < d->B) {
    set_error("PD_BWD_DEFER_GATHER needs the whole batch's scratch (PD_UNI_CHUNK is set)");
    return PD_ERR_UNSUPPORTED;
  }
  for (int b0 = 0; b0 < d->B && !rc; b0 += chunk) {
    const int nb = (d->B - b0) < chunk ? (d->B - b0) : chunk;
    dim3 grid(nblk, nb);
    const int stiles_x = ceil_div(d->W, kStageW);
    dim3 sgrid(stiles_x * ceil_div(d->H, kStageH), nb);
    // pass 2 with the scratch staged through LDS unless the fused kernel is selected (its run_flag protocol belongs to
    // the direct-gather kernel) or PD_IMPL_UNIFORM_DIRECT asks for the direct gather (cross-check)
    const bool staged = !fused && d->impl != PD_IMPL_UNIFORM_DIRECT;
    if (mix) {
      if (render) uniform_bwd_pass1_kernel<true, true><<<grid, kBlock, 0, stream>>>(ak, o, b0, tmp, part_two, tw, run_flag);
      else        uniform_bwd_pass1_kernel<true, false><<<grid, kBlock, 0, stream>>>(ak, o, b0, tmp, part_two, tw, run_flag);
      if (defer) { /* the caller gathers: pd_uniform_gather_pair */ }
      else {
      if (staged) uniform_bwd_pass2_staged_kernel<true><<<sgrid, kStageThreads, 0, stream>>>(ak, b0, tmp, prep, o.g_logits, o.g_sigma, overflow, stiles_x, accumulate);
      else uniform_bwd_pass2_kernel<true, false><<<grid, kBlock, 0, stream>>>(ak, b0, tmp, prep, o.g_logits, o.g_sigma, overflow, run_flag, accumulate);
      uniform_bwd_pass2_kernel<true, true><<<grid, kBlock, 0, stream>>>(ak, b0, tmp, prep, o.g_logits, o.g_sigma, overflow, run_flag, accumulate);
      }
    } else {
      if (render) uniform_bwd_pass1_kernel<false, true><<<grid, kBlock, 0, stream>>>(ak, o, b0, tmp, part_two, tw, run_flag);
      else        uniform_bwd_pass1_kernel<false, false><<<grid, kBlock, 0, stream>>>(ak, o, b0, tmp, part_two, tw, run_flag);
      if (defer) { /* the caller gathers: pd_uniform_gather_pair */ }
      else {
      if (staged) uniform_bwd_pass2_staged_kernel<false><<<sgrid, kStageThreads, 0, stream>>>(ak, b0, tmp, prep, o.g_logits, nullptr, overflow, stiles_x, accumulate);
      else uniform_bwd_pass2_kernel<false, false><<<grid, kBlock, 0, stream>>>(ak, b0, tmp, prep, o.g_logits, nullptr, overflow, run_flag, accumulate);
      uniform_bwd_pass2_kernel<false, true><<<grid, kBlock, 0, stream>>>(ak, b0, tmp, prep, o.g_logits, nullptr, overflow, run_flag, accumulate);
      }
    }
    rc = check_launch("uniform_bwd_pass kernels");
  }
  if (rc || !o.g_plane) return rc;
  uniform_reduce_kernel<<<dim3(kUniG * 9, d->B), kWave, 0, stream>>>(fused ? part_fused : nullptr, ntiles, part_two, nblk,
                                                                     irregular, o.g_plane, kUniG * 9);
  return check_launch("uniform_reduce_kernel");
}

int uniform_gather_pair(const pd_sweep_desc* d, const float* plane_a, const float* inv_K3_a, float* workspace_a,
                        const float* plane_b, const float* inv_K3_b, float* workspace_b, float* g_logits, float* g_sigma,
                        hipStream_t stream) {
  const bool mix = (d->flags & PD_MIXTURE) != 0;
  const int accumulate = (d->flags & PD_BWD_ACCUMULATE) ? 1 : 0;
  const UniformWs wa = uniform_ws(d, workspace_a), wb = uniform_ws(d, workspace_b);
  PairArgs pa;
  pa.N = d->N; pa.H = d->H; pa.W = d->W;
  pa.plane[0] = plane_a; pa.plane[1] = plane_b;
  pa.inv_K3[0] = inv_K3_a; pa.inv_K3[1] = inv_K3_b;
  pa.tmp[0] = wa.tmp; pa.tmp[1] = wb.tmp;
  pa.prep[0] = wa.prep; pa.prep[1] = wb.prep;
  pa.overflow[0] = wa.overflow; pa.overflow[1] = wb.overflow;
  const int stiles_x = ceil_div(d->W, kStageW);
  const dim3 sgrid(stiles_x * ceil_div(d->H, kStageH), d->B), grid(ceil_div(d->H * d->W, kBlock), d->B);
  if (mix) uniform_bwd_pass2_pair_kernel<true><<<sgrid, kStageThreads, 0, stream>>>(pa, g_logits, g_sigma, stiles_x, accumulate);
  else     uniform_bwd_pass2_pair_kernel<false><<<sgrid, kStageThreads, 0, stream>>>(pa, g_logits, nullptr, stiles_x, accumulate);
  // source pixels with more than kPairK contributors in a view: that view's follow-up adds them (returns at once otherwise)
  for (int v = 0; v < 2; ++v) {
    SweepArgs a{};
    a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W; a.flags = d->flags;
    a.plane = v ? plane_b : plane_a; a.inv_K3 = v ? inv_K3_b : inv_K3_a;
    const UniformWs& w = v ? wb : wa;
    if (mix) uniform_bwd_pass2_kernel<true, true><<<grid, kBlock, 0, stream>>>(a, 0, w.tmp, w.prep, g_logits, g_sigma, w.overflow, nullptr, 1, kPairK);
    else     uniform_bwd_pass2_kernel<false, true><<<grid, kBlock, 0, stream>>>(a, 0, w.tmp, w.prep, g_logits, nullptr, w.overflow, nullptr, 1, kPairK);
  }
  return check_launch("uniform_bwd_pass2_pair_kernel");
}

// The whole backward of two plane-uniform views of one source image (pd_uniform_bwd_pair): the first passes, the pair
// gather, the homography gradients' reductions.
int uniform_bwd_pair(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& oa, float* workspace_a, const SweepArgs& b,
                     const BwdOut& ob, float* workspace_b, float* g_logits, float* g_sigma, hipStream_t stream) {
  const int nblk = ceil_div(d->H * d->W, kBlock);
  const bool mix = (d->flags & PD_MIXTURE) != 0, render = (d->flags & PD_RENDER_PROB) != 0;
  const UniformWs wa = uniform_ws(d, workspace_a), wb = uniform_ws(d, workspace_b);
  uniform_prep_kernel<<<ceil_div(d->B, 64), 64, 0, stream>>>(a.plane, wa.prep, d->B);
  uniform_prep_kernel<<<ceil_div(d->B, 64), 64, 0, stream>>>(b.plane, wb.prep, d->B);
  int rc = check_launch("uniform_prep_kernel");
  if (rc) return rc;
  // (both first passes in ONE launch, as the forward does it, was measured: 0.400 ms against 2 x 0.193 — they are paced by
  // their scratch stores, 770 MB for the two views, not by the logits / sigma reads the views could share)
  for (int v = 0; v < 2 && !rc; ++v) {
    const SweepArgs& s = v ? b : a;
    const BwdOut& o = v ? ob : oa;
    const UniformWs& w = v ? wb : wa;
    SweepArgs ak = s;
    ak.padding_mask = nullptr;
    const dim3 grid(nblk, d->B);
    if (mix) { if (render) uniform_bwd_pass1_kernel<true, true><<<grid, kBlock, 0, stream>>>(ak, o, 0, w.tmp, w.part_two, s.padding_mask, nullptr);
               else        uniform_bwd_pass1_kernel<true, false><<<grid, kBlock, 0, stream>>>(ak, o, 0, w.tmp, w.part_two, s.padding_mask, nullptr); }
    else     { if (render) uniform_bwd_pass1_kernel<false, true><<<grid, kBlock, 0, stream>>>(ak, o, 0, w.tmp, w.part_two, s.padding_mask, nullptr);
               else        uniform_bwd_pass1_kernel<false, false><<<grid, kBlock, 0, stream>>>(ak, o, 0, w.tmp, w.part_two, s.padding_mask, nullptr); }
    rc = check_launch("uniform_bwd_pass1_kernel");
  }
  if (rc) return rc;
  if (g_logits) {
    rc = uniform_gather_pair(d, a.plane, a.inv_K3, workspace_a, b.plane, b.inv_K3, workspace_b, g_logits, g_sigma, stream);
    if (rc) return rc;
  }
  if (oa.g_plane) uniform_reduce_kernel<<<dim3(kUniG * 9, d->B), kWave, 0, stream>>>(nullptr, 0, wa.part_two, nblk, wa.irregular, oa.g_plane, kUniG * 9);
  if (ob.g_plane) uniform_reduce_kernel<<<dim3(kUniG * 9, d->B), kWave, 0, stream>>>(nullptr, 0, wb.part_two, nblk, wb.irregular, ob.g_plane, kUniG * 9);
  return check_launch("uniform_reduce_kernel");
}

}  // namespace pd
