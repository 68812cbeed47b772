// General backward of the fused plane sweep WITHOUT atomics: homography_warp (reference trainer.py:556-560 +
// layers.py:206-234 + the autograd of trainer.py:567-603, 728-742), any pose, any plane normals.
//
// The adjoint of F.grid_sample's bilinear gather is a scatter: target pixel t adds g(t) * w to the four source pixels
// under its sampling position.  The first general backward (pd_plane_sweep.hip) did that with global_atomic_add_f32
// into zero-filled gradients: ~106 M lane-atomics per launch at 8x49x192x640, bound by the L2's atomic unit (0.73 ms
// stereo / 1.21 ms mono pose, 0.14 / 0.08 of the HBM roofline) plus 385 MB of zero-fill per step.
//
// Here the DESTINATION is owned.  A workgroup owns a TR x TC tile of source pixels of image b and walks the planes:
//   1. per plane it maps the tile (grown by the bilinear footprint) through the plane's FORWARD homography
//      H_s2t = inverse(H_t2s) — computed once per (b, n) by `tile_prep_kernel` in fp64 — and takes the bounding box of
//      the image: every target pixel whose sample can touch the tile lies in that box (a homography maps a convex
//      region that stays on one side of the line at infinity to a convex region; boxes that straddle it fall back to
//      the whole image).  A margin of a quarter pixel in source space and of the box rounding in target space covers
//      the fp32 noise of the reference's coordinate chain and of the inverse (both < 1e-2 px);
//   2. the threads sweep the target pixels of the box: per pixel the packed per-target-pixel context (softmax
//      statistics, upstream gradients: `pack_ctx_kernel`, 48 bytes), the bit-exact sampling position of the forward,
//      the five-channel bilinear samples with their derivatives and the closed-form per-plane gradients (pd_sweep.h);
//      the four tap contributions go to the tile's LDS accumulators with ds_add_f32 when the tap lies in the tile and
//      are dropped otherwise — the neighbouring tile's workgroup computes that target pixel again (the price of
//      ownership: (TR+3)(TC+3)/(TR*TC) = 1.24 x the samples at 16 x 64);
//   3. after a barrier the tile is written to g_logits / g_sigma with plain coalesced stores — every element of the
//      gradients exactly once, no zero-fill, no read-modify-write — and the accumulators are cleared (two LDS buffers
//      alternate, so one barrier per plane suffices).
// The gradient of the homography entries is accumulated per thread over the pixels whose (clamped) top-left tap lies
// in the tile — a unique owner per target pixel and plane — reduced per workgroup and plane, and finished by the
// deterministic second-stage reduction the other kernels use.
#ifdef PD_EXPERIMENTS   // measured slower than the default kernels: built with -DPD_EXPERIMENTS only (scripts/build_variants.sh)
#include "pd_sweep_geom.h"

namespace pd {

#ifndef PD_TILE_ABL
#define PD_TILE_ABL 0   // diagnostics: 1 no LDS adds, 2 no context loads, 4 no flush, 8 no geometry early-outs
#endif
constexpr int kTileR = 16, kTileC = 64;   // source tile (rows x columns); TC = one wave of columns
constexpr int kTileThreads = 256;
constexpr float kSrcMargin = 0.25f;       // source-space growth of the tile before it is mapped to the target view

struct TilePrep {       // per (b, n), written by tile_prep_kernel
  float Hs[9];          // forward homography source -> target (inverse of H_t2s), fp32 from an fp64 adjugate
  float ok;             // 1: finite and well-conditioned enough to trust the bounding boxes; 0: scan the whole image
  float pad[2];
};

__global__ void tile_prep_kernel(const float* __restrict__ H_t2s, TilePrep* __restrict__ prep, int M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float* h = H_t2s + (long)i * 9;
  const double a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], k = h[7], l = h[8];
  const double A = e * l - f * k, B = -(d * l - f * g), C = d * k - e * g;
  const double det = a * A + b * B + c * C;
  TilePrep p;
  const double inv = 1.0 / det;
  p.Hs[0] = (float)(A * inv);  p.Hs[1] = (float)(-(b * l - c * k) * inv); p.Hs[2] = (float)((b * f - c * e) * inv);
  p.Hs[3] = (float)(B * inv);  p.Hs[4] = (float)((a * l - c * g) * inv);  p.Hs[5] = (float)(-(a * f - c * d) * inv);
  p.Hs[6] = (float)(C * inv);  p.Hs[7] = (float)(-(a * k - b * g) * inv); p.Hs[8] = (float)((a * e - b * d) * inv);
  bool ok = (det == det) && fabs(det) > 1e-30 && fabs(inv) < 1e30;
  for (int j = 0; j < 9; ++j) ok = ok && (fabsf(p.Hs[j]) < 1e30f) && (p.Hs[j] == p.Hs[j]);
  p.ok = ok ? 1.0f : 0.0f;
  p.pad[0] = p.pad[1] = 0.0f;
  prep[i] = p;
}

// The per-target-pixel context of the backward (pd_sweep.h PixelCtx), packed as three float4 planes [B][3][HW]:
//   {t0, t1, t2, lse2}  {invS, mx, A, gdotr}  {gr0, gr1, gr2, -}
template <bool MIX>
__global__ __launch_bounds__(kBlock) void pack_ctx_kernel(SweepArgs a, BwdOut o, float4* __restrict__ ctx) {
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= HW) return;
  const PixelCtx c = make_pixel_ctx<MIX>(a, o, b, pix, HW);
  float4* q = ctx + (long)b * 3 * HW + pix;
  q[0] = make_float4(c.t0, c.t1, c.t2, c.lse2);
  q[HW] = make_float4(c.invS, c.mx, c.A, c.gdotr);
  q[2 * HW] = make_float4(c.gr0, c.gr1, c.gr2, 0.0f);
}

// Target-view bounding box of everything that can sample into the source rectangle [sx0, sx1] x [sy0, sy1]
// (continuous coordinates), clamped to the image.  Returns false when the box is empty.
__device__ __forceinline__ bool target_box(const TilePrep& p, float sx0, float sy0, float sx1, float sy1, int W, int H,
                                           int& tx0, int& ty0, int& tx1, int& ty1) {
  tx0 = 0; ty0 = 0; tx1 = W - 1; ty1 = H - 1;
  if (p.ok == 0.0f) return true;
  float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
  float wmin = 3.0e38f, wmax = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float x = (k & 1) ? sx1 : sx0, y = (k & 2) ? sy1 : sy0;
    const float u = p.Hs[0] * x + p.Hs[1] * y + p.Hs[2];
    const float v = p.Hs[3] * x + p.Hs[4] * y + p.Hs[5];
    const float w = p.Hs[6] * x + p.Hs[7] * y + p.Hs[8];
    wmin = fminf(wmin, w); wmax = fmaxf(wmax, w);
    const float r = 1.0f / w;
    xmin = fminf(xmin, u * r); xmax = fmaxf(xmax, u * r);
    ymin = fminf(ymin, v * r); ymax = fmaxf(ymax, v * r);
  }
  // the rectangle must stay clear of the line that maps to infinity (w = 0); H_s2t is only defined up to scale, so
  // "one sign, not close to zero relative to its size" is the test.  Otherwise: the whole image.
  const float wabs = fmaxf(fabsf(wmin), fabsf(wmax));
  if (!(wmin * wmax > 0.0f) || !(fminf(fabsf(wmin), fabsf(wmax)) > 1e-3f * wabs)) return true;
  if (!(xmin == xmin) || !(xmax == xmax) || !(ymin == ymin) || !(ymax == ymax)) return true;
  const float fx0 = floorf(xmin - 0.25f), fx1 = ceilf(xmax + 0.25f), fy0 = floorf(ymin - 0.25f), fy1 = ceilf(ymax + 0.25f);
  tx0 = (int)fminf(fmaxf(fx0, 0.0f), (float)W);
  ty0 = (int)fminf(fmaxf(fy0, 0.0f), (float)H);
  tx1 = (int)fminf(fmaxf(fx1, -1.0f), (float)(W - 1));
  ty1 = (int)fminf(fmaxf(fy1, -1.0f), (float)(H - 1));
  return tx1 >= tx0 && ty1 >= ty0;
}

struct TileOut {
  float* g_logits;
  float* g_sigma;
  float* partials;   // [B][ntiles][N][9] or NULL
};

template <bool MIX>
__global__ __launch_bounds__(kTileThreads) void sweep_bwd_tile_kernel(SweepArgs a, TileOut o, const float4* __restrict__ ctx,
                                                                      const TilePrep* __restrict__ prep, int tiles_x,
                                                                      int planes_per_chunk) {
  // LDS: two alternating tile accumulators [2][2 tensors][TR*TC] and two alternating sets of the 9 homography sums
  __shared__ float acc[2][2][kTileR * kTileC];
  __shared__ float red[2][12];
  const int HW = a.H * a.W, W = a.W, H = a.H, N = a.N;
  const int b = blockIdx.y;
  const int tile = blockIdx.x, tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int ys0 = tyi * kTileR, xs0 = txi * kTileC;
  const int th = min(kTileR, H - ys0), tw = min(kTileC, W - xs0);
  const int n_lo = blockIdx.z * planes_per_chunk, n_hi = min(N, n_lo + planes_per_chunk);
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * 2 * kTileR * kTileC; i += kTileThreads) (&acc[0][0][0])[i] = 0.0f;
  if (tid < 24) (&red[0][0])[tid] = 0.0f;
  __syncthreads();

  const float* srcb = a.src + (long)b * 3 * HW;
  const float4* ctxb = ctx + (long)b * 3 * HW;
  const CoordNorm cn = make_coord_norm(W, H);
  const float gscale_x = (float)(W - 1) / 2 * 2.0f / (float)(W - 1), gscale_y = (float)(H - 1) / 2 * 2.0f / (float)(H - 1);
  const bool want_plane = (o.partials != nullptr);
  const int lane = tid & (kWave - 1);
  // the source rectangle whose samples touch the tile: ix in (xs0 - 1, xs0 + tw), iy in (ys0 - 1, ys0 + th)
  const float sx0 = (float)(xs0 - 1) - kSrcMargin, sx1 = (float)(xs0 + tw) + kSrcMargin;
  const float sy0 = (float)(ys0 - 1) - kSrcMargin, sy1 = (float)(ys0 + th) + kSrcMargin;

  for (int n = n_lo; n < n_hi; ++n) {
    const int p = (n - n_lo) & 1;
    float* accL = acc[p][0];
    float* accS = acc[p][1];
    int tx0, ty0, tx1, ty1;
    const bool any = target_box(prep[(long)b * N + n], sx0, sy0, sx1, sy1, W, H, tx0, ty0, tx1, ty1);
    float gk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gk[k] = 0.0f;
    const long pl = ((long)b * N + n) * HW;
    if (any) {
      const int bw = tx1 - tx0 + 1, bh = ty1 - ty0 + 1, npx = bw * bh;
      const float rbw = 1.0f / (float)bw;
      for (int idx = tid; idx < npx; idx += kTileThreads) {
        int row = (int)(((float)idx + 0.5f) * rbw);       // idx / bw with a fix-up (bw is uniform, idx < 2^23)
        if (row * bw > idx) --row;
        if ((row + 1) * bw <= idx) ++row;
        const int x = tx0 + (idx - row * bw), y = ty0 + row;
        const int pix = y * W + x;
        bool mk;
        const PlaneGeom g = plane_coords<PD_WARP_HOMOGRAPHY>(a, cn, b, n, x, y, 0.0f, mk);
        if (!mk) continue;   // a masked plane samples as zeros (trainer.py:580): no gradient reaches the source
        const Tap t = make_tap(g.ix, g.iy, W, H);
        // taps of this sample inside the tile?  (tile pixels are image pixels, so no separate validity test)
        const int lx = t.x0 - xs0, ly = t.y0 - ys0;
        const bool cx0 = (unsigned)lx < (unsigned)tw, cx1 = (unsigned)(lx + 1) < (unsigned)tw;
        const bool cy0 = (unsigned)ly < (unsigned)th, cy1 = (unsigned)(ly + 1) < (unsigned)th;
        // owner of the pixel's homography gradient: the tile holding its (clamped) top-left tap
        const int ox = min(max(t.x0, 0), W - 1) - xs0, oy = min(max(t.y0, 0), H - 1) - ys0;
        const bool owner = want_plane && (unsigned)ox < (unsigned)tw && (unsigned)oy < (unsigned)th;
        if (!((cx0 || cx1) && (cy0 || cy1)) && !owner) continue;
        float4 q0, q1, q2;
        if (PD_TILE_ABL & 2) { q0 = make_float4(0.1f, 0.2f, 0.3f, 1.0f); q1 = make_float4(1.0f, 0.5f, 0.25f, 0.1f); q2 = q0; }
        else { q0 = ctxb[pix]; q1 = ctxb[HW + pix]; q2 = ctxb[2 * HW + pix]; }
        PixelCtx c;
        c.t0 = q0.x; c.t1 = q0.y; c.t2 = q0.z; c.lse2 = q0.w;
        c.invS = q1.x; c.mx = q1.y; c.A = q1.z; c.gdotr = q1.w;
        c.gr0 = q2.x; c.gr1 = q2.y; c.gr2 = q2.z;
        const TapK tk = tap_kernel(t, W, H);
        float dlx, dly, dsx = 0, dsy = 0, d0x, d0y, d1x, d1y, d2x, d2y;
        const float l = sample_vg_k(a.logits + pl, tk, dlx, dly);
        const float c0 = sample_vg_k(srcb, tk, d0x, d0y);
        const float c1 = sample_vg_k(srcb + HW, tk, d1x, d1y);
        const float c2 = sample_vg_k(srcb + 2 * HW, tk, d2x, d2y);
        const float s = MIX ? sample_vg_k(a.sigma + pl, tk, dsx, dsy) : 0.0f;
        const PlaneGrad pg = plane_grad<MIX>(c, l, s, c0, c1, c2);
        {  // adjoint of the bilinear gather into the tile
          const float w00 = t.wx0 * t.wy0, w01 = t.wx1 * t.wy0, w10 = t.wx0 * t.wy1, w11 = t.wx1 * t.wy1;
          const int e00 = ly * kTileC + lx;
          if (PD_TILE_ABL & 1) { if (pg.g_l * w00 + pg.g_s * w11 + w01 + w10 == 123.456f) accL[0] = 1.0f; } else {
          if (cx0 && cy0) { lds_add(accL + e00, pg.g_l * w00); if (MIX) lds_add(accS + e00, pg.g_s * w00); }
          if (cx1 && cy0) { lds_add(accL + e00 + 1, pg.g_l * w01); if (MIX) lds_add(accS + e00 + 1, pg.g_s * w01); }
          if (cx0 && cy1) { lds_add(accL + e00 + kTileC, pg.g_l * w10); if (MIX) lds_add(accS + e00 + kTileC, pg.g_s * w10); }
          if (cx1 && cy1) { lds_add(accL + e00 + kTileC + 1, pg.g_l * w11); if (MIX) lds_add(accS + e00 + kTileC + 1, pg.g_s * w11); }
          }
        }
        if (owner) {
          const float gix = pg.g_l * dlx + pg.g_s * dsx + pg.gc0 * d0x + pg.gc1 * d1x + pg.gc2 * d2x;
          const float giy = pg.g_l * dly + pg.g_s * dsy + pg.gc0 * d0y + pg.gc1 * d1y + pg.gc2 * d2y;
          const float gpx = gix * gscale_x, gpy = giy * gscale_y;
          float inv_z = fast_rcp(g.zc);
          inv_z = fmaf(fmaf(-g.zc, inv_z, 1.0f), inv_z, inv_z);
          const float gp0 = gpx * inv_z, gp1 = gpy * inv_z;
          const float gz = g.z_clamped ? 0.0f : -(gp0 * g.p0 + gp1 * g.p1) * inv_z;
          const float fx = (float)x, fy = (float)y;
          gk[0] += gp0 * fx; gk[1] += gp0 * fy; gk[2] += gp0;
          gk[3] += gp1 * fx; gk[4] += gp1 * fy; gk[5] += gp1;
          gk[6] += gz * fx;  gk[7] += gz * fy;  gk[8] += gz;
        }
      }
    }
    if (want_plane) {  // workgroup totals of the homography gradient for this plane
#pragma unroll
      for (int k = 0; k + 1 < 9; k += 2) {
        const float v = half_wave_sums_hi(gk[k], gk[k + 1]);
        if ((lane & 31) == 31) lds_add(&red[p][k + (lane >> 5)], v);
      }
      const float v8 = wave_sum_hi(gk[8]);
      if (lane == kWave - 1) lds_add(&red[p][8], v8);
    }
    __syncthreads();
    // flush: the tile's rows go out as they are (plain stores), then the accumulators are cleared for plane n + 2
    {
      float* gl = o.g_logits ? o.g_logits + pl + (long)ys0 * W + xs0 : nullptr;
      float* gs = (MIX && o.g_sigma) ? o.g_sigma + pl + (long)ys0 * W + xs0 : nullptr;
      if (((tw | W) & 3) == 0) {   // rows are 16-byte aligned: one float4 per lane
        for (int e = tid * 4; e < th * kTileC; e += kTileThreads * 4) {
          const int ry = e / kTileC, rx = e - ry * kTileC;   // kTileC is a power of two
          if (rx < tw) {
            const float4 vl = *reinterpret_cast<const float4*>(accL + e);
            if (gl) *reinterpret_cast<float4*>(gl + (long)ry * W + rx) = vl;
            *reinterpret_cast<float4*>(accL + e) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MIX) {
              const float4 vs = *reinterpret_cast<const float4*>(accS + e);
              if (gs) *reinterpret_cast<float4*>(gs + (long)ry * W + rx) = vs;
              *reinterpret_cast<float4*>(accS + e) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
      } else {
        for (int e = tid; e < th * kTileC; e += kTileThreads) {
          const int ry = e / kTileC, rx = e - ry * kTileC;
          if (rx < tw) {
            if (gl) gl[(long)ry * W + rx] = accL[e];
            accL[e] = 0.0f;
            if (MIX) {
              if (gs) gs[(long)ry * W + rx] = accS[e];
              accS[e] = 0.0f;
            }
          }
        }
      }
      if (want_plane && tid < 9) {
        o.partials[(((long)b * gridDim.x + tile) * N + n) * 9 + tid] = red[p][tid];
        red[p][tid] = 0.0f;
      }
    }
  }
}

bool tile_bwd_applicable(const pd_sweep_desc* d) {
  return d->mode == PD_WARP_HOMOGRAPHY && !(d->flags & PD_RENDER_PROB) && d->impl == PD_IMPL_TILE;
}

static int tile_count(const pd_sweep_desc* d) { return ceil_div(d->H, kTileR) * ceil_div(d->W, kTileC); }

// workspace: [B][ntiles][N][9] partial sums | TilePrep [B*N] | ctx float4 [B][3][HW]   (16-byte aligned pieces)
static size_t align4(size_t floats) { return (floats + 3) & ~(size_t)3; }
size_t tile_bwd_workspace_floats(const pd_sweep_desc* d) {
  const size_t part = align4((size_t)d->B * tile_count(d) * d->N * 9);
  const size_t prep = align4((size_t)d->B * d->N * (sizeof(TilePrep) / sizeof(float)));
  const size_t ctx = (size_t)d->B * 3 * d->H * d->W * 4;
  return part + prep + ctx + 4;
}

int tile_bwd(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, float* workspace, hipStream_t stream) {
  const int HW = d->H * d->W, ntiles = tile_count(d);
  const bool mix = (d->flags & PD_MIXTURE) != 0;
  // carve the workspace (the caller's pointer comes from a tensor allocation: at least 16-byte aligned; keep it so)
  uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15;
  float* partials = reinterpret_cast<float*>(base);
  TilePrep* prep = reinterpret_cast<TilePrep*>(partials + align4((size_t)d->B * ntiles * d->N * 9));
  float4* ctx = reinterpret_cast<float4*>(reinterpret_cast<float*>(prep) +
                                          align4((size_t)d->B * d->N * (sizeof(TilePrep) / sizeof(float))));
  const int M = d->B * d->N;
  tile_prep_kernel<<<ceil_div(M, 128), 128, 0, stream>>>(a.plane, prep, M);
  int rc = check_launch("tile_prep_kernel");
  if (rc) return rc;
  dim3 pgrid(ceil_div(HW, kBlock), d->B);
  if (mix) pack_ctx_kernel<true><<<pgrid, kBlock, 0, stream>>>(a, o, ctx);
  else     pack_ctx_kernel<false><<<pgrid, kBlock, 0, stream>>>(a, o, ctx);
  rc = check_launch("pack_ctx_kernel");
  if (rc) return rc;
  // plane chunks: enough workgroups to fill 256 CUs a few times over, few enough that the per-workgroup prologue
  // (zeroing 16 KB of LDS) stays small next to the plane loop
  int chunks = 1;
  while ((long)ntiles * d->B * chunks < 2048 && chunks * 4 <= d->N) ++chunks;
  const int ppc = ceil_div(d->N, chunks);
  chunks = ceil_div(d->N, ppc);
  TileOut to;
  to.g_logits = o.g_logits; to.g_sigma = mix ? o.g_sigma : nullptr; to.partials = o.g_plane ? partials : nullptr;
  dim3 grid(ntiles, d->B, chunks);
  const int tiles_x = ceil_div(d->W, kTileC);
  if (mix) sweep_bwd_tile_kernel<true><<<grid, kTileThreads, 0, stream>>>(a, to, ctx, prep, tiles_x, ppc);
  else     sweep_bwd_tile_kernel<false><<<grid, kTileThreads, 0, stream>>>(a, to, ctx, prep, tiles_x, ppc);
  rc = check_launch("sweep_bwd_tile_kernel");
  if (rc || !o.g_plane) return rc;
  return reduce_partials(partials, o.g_plane, ntiles, d->N * 9, d->B, stream);
}

}  // namespace pd

#endif  // PD_EXPERIMENTS
