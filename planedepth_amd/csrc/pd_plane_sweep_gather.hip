// homography_warp with ONE HOMOGRAPHY PER PLANE (6-DoF poses: --use_colmap, reference trainer.py:397-398; any pose with
// a translation, layers.py:206-219) — the backward without atomics.
//
// The general backward (pd_plane_sweep.hip) scatters every sample's gradient into its four taps with atomics and is
// bound by the L2's atomic unit (NOTEBOOK.md 3.4.6: 1.2 ms at 8x49x192x640, 0.42 ms of it without the scatter).  Here the
// adjoint of the bilinear gather is turned round, as in the plane-uniform kernels (pd_plane_sweep_uniform.hip), but per
// plane:
//   pass 1 = sweep_bwd_kernel<.., TOSCRATCH> (pd_plane_sweep.hip): the target-anchored closed-form gradients, written
//            side by side as (g_l, g_s) to a scratch [B][N][H*W] with coalesced stores; the homography gradient as before;
//   pass 2 (this file, one thread per SOURCE pixel, all planes): the target pixels whose bilinear footprint covers the
//            source pixel are the integer points in the pre-image of the open square s +- 1 under plane n's homography —
//            located with the inverse matrix (fp64 adjugate, gather_prep_kernel), every candidate CONFIRMED with the
//            forward's own coordinate chain (plane_coords, bit for bit), which also yields torch's bilinear weight —
//            g[n][s] = sum_t w(t, s) * scratch[n][t], one plain coalesced store per element: no zero-fill, no atomics,
//            deterministic.  PD_BWD_ACCUMULATE adds to what is there.
// Planes whose map is not a moderate, orientation-preserving deformation over the whole image (line at infinity near
// the view, minification beyond ~3x: nothing a pose produces, but a diverged pose net may) are flagged by the prepare
// kernel, written as zeros by pass 2 and served by gather_fixup_kernel with atomics out of the same scratch — the same
// numbers as the general kernel, at its speed, for those planes only.  A plane whose matrix is not finite / not
// invertible gets no gradient from pass 2 or the fix-up's scatter other than what its finite samples give.
#include <type_traits>

#include "pd_sweep_geom.h"

namespace pd {

struct GatherPrep {   // per (image, plane)
  float Hs[9];        // source -> target, from an fp64 adjugate of H_t2s
  float regular;      // 1: pass 2 gathers this plane; 0: the fix-up kernel scatters it
  float pad[2];
};
constexpr int kGatherPrepFloats = sizeof(GatherPrep) / sizeof(float);

// A sample reaches source pixel s iff it lies in the open square s +- 1.  The square is grown by 1/256 pixel before it is
// mapped and the mapped box by another 1/256: the fp32 noise of the forward chain and of the inverse matrix are ~1e-4
// pixel each at 640 columns (3e-4 at 2048), and every candidate is confirmed with the exact forward coordinates anyway.
// The margins are kept that tight because they decide how many lanes walk 3 instead of 2 candidates per axis (a wave pays
// for its longest lane): 1/32 + 1/32 made that 12.5 % of the lanes per axis, these make it 0.8 %.
constexpr float kGatherReach = 1.00390625f, kGatherSlop = 0.00390625f;
constexpr int kGatherSpan = 16;      // columns / rows of a window pass 2 walks at most (regular planes stay below it)
constexpr int kGatherSpanOk = 7;     // what the prepare kernel accepts on its 5 x 5 sample of the image
constexpr float kGatherWRatio = 0.7f;  // min |w| / max |w| of the inverse map's denominator over the (grown) image

// A thread of pass 2 owns a 2 x 2 block of source pixels (sx, sy) .. (sx+1, sy+1): the exact coordinates of a candidate
// are the expensive part (two IEEE divisions, the normalisation round trip) and serve all four pixels, and a wave pays
// for its largest window — 9..16 candidates for four pixels instead of 4..9 for one.
// Pre-image box of the block's footprint (sx - reach, sx + 1 + reach) x (sy - reach, sy + 1 + reach): integer candidates
// [x0, x1] x [y0, y1], NOT clamped to the image
constexpr float kGatherHalf = 0.5f + kGatherReach;
struct GatherWindow { float x0, x1, y0, y1; };
__device__ __forceinline__ GatherWindow gather_window(const float* __restrict__ Hs, float sx, float sy) {
  const float cx = sx + 0.5f, cy = sy + 0.5f;
  const float u = Hs[0] * cx + Hs[1] * cy + Hs[2], v = Hs[3] * cx + Hs[4] * cy + Hs[5], w = Hs[6] * cx + Hs[7] * cy + Hs[8];
  float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float ex = (k & 1) ? kGatherHalf : -kGatherHalf, ey = (k & 2) ? kGatherHalf : -kGatherHalf;
    const float r = fast_rcp(w + ex * Hs[6] + ey * Hs[7]);   // (1 ulp: far inside the margins)
    const float x = (u + ex * Hs[0] + ey * Hs[1]) * r, y = (v + ex * Hs[3] + ey * Hs[4]) * r;
    xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
    ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
  }
  GatherWindow g;
  g.x0 = ceilf(xmin - kGatherSlop); g.x1 = floorf(xmax + kGatherSlop);
  g.y0 = ceilf(ymin - kGatherSlop); g.y1 = floorf(ymax + kGatherSlop);
  return g;
}

// flags[0]: some plane of the launch is irregular (the fix-up kernel has work); flags[1]: windows pass 2 cut at kGatherSpan
// (the prepare kernel's bound on the window growth between its samples rules that out for the planes it accepts; the
// flag stays as the tests' tripwire: pd_debug_gather_flags)
__global__ void gather_prep_kernel(const float* __restrict__ H_t2s, GatherPrep* __restrict__ prep, int* __restrict__ flags,
                                   int BN, int W, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BN) return;
  const float* h = H_t2s + (long)i * 9;
  const double a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], k = h[7], l = h[8];
  const double A = e * l - f * k, Bc = -(d * l - f * g), C = d * k - e * g;
  const double det = a * A + b * Bc + c * C, inv = 1.0 / det;
  GatherPrep p;
  p.Hs[0] = (float)(A * inv);  p.Hs[1] = (float)(-(b * l - c * k) * inv); p.Hs[2] = (float)((b * f - c * e) * inv);
  p.Hs[3] = (float)(Bc * inv); p.Hs[4] = (float)((a * l - c * g) * inv);  p.Hs[5] = (float)(-(a * f - c * d) * inv);
  p.Hs[6] = (float)(C * inv);  p.Hs[7] = (float)(-(a * k - b * g) * inv); p.Hs[8] = (float)((a * e - b * d) * inv);
  bool ok = (det == det) && fabs(det) > 1e-30 && fabs(inv) < 1e30;
  for (int j = 0; j < 9; ++j) ok = ok && (fabsf(p.Hs[j]) < 1e30f) && (p.Hs[j] == p.Hs[j]);
  float wspread = 1.0f;   // max |w| / min |w| over the grown image
  if (ok) {   // the denominator is affine in the source position: its extremes over the grown image are at the corners
    float wmin = 3.0e38f, wmax = -3.0e38f;
    for (int q = 0; q < 4; ++q) {
      const float x = (q & 1) ? (float)W + kGatherReach : -kGatherReach, y = (q & 2) ? (float)H + kGatherReach : -kGatherReach;
      const float w = p.Hs[6] * x + p.Hs[7] * y + p.Hs[8];
      wmin = fminf(wmin, w); wmax = fmaxf(wmax, w);
    }
    ok = (wmin * wmax > 0.0f) && (fminf(fabsf(wmin), fabsf(wmax)) >= kGatherWRatio * fmaxf(fabsf(wmin), fabsf(wmax)));
    if (ok) wspread = fmaxf(fabsf(wmin), fabsf(wmax)) / fminf(fabsf(wmin), fabsf(wmax));
  }
  if (ok) {   // window sizes on a 5 x 5 sample of the source image
    for (int qy = 0; qy < 5 && ok; ++qy)
      for (int qx = 0; qx < 5 && ok; ++qx) {
        const GatherWindow w = gather_window(p.Hs, (float)(W - 1) * 0.25f * qx, (float)(H - 1) * 0.25f * qy);
        // The window of a block scales with the Jacobian of u = (Hs x)_xy / w(x): J = (N' - u w') / w with N = Hs_xy x affine
        // and u confined to the image (plus the window) wherever a target samples the plane, i.e. |J| ~ 1 / |w| times a
        // factor that varies like u w' / N' — bounded here by one more power of the spread.  A plane is regular only if that
        // bound keeps EVERY window inside the kGatherSpan columns / rows pass 2 walks (a cut window would silently drop
        // gradient terms, flags[1]): with the acceptance thresholds below, kGatherSpanOk / kGatherWRatio^2 = 7 / 0.49 =
        // 14.3 < kGatherSpan - 1, so the explicit test only bites if those constants are ever loosened.
        const float grow = wspread * wspread;
        const float sx = w.x1 - w.x0 + 1.0f, sy = w.y1 - w.y0 + 1.0f;
        ok = (sx <= (float)kGatherSpanOk) && (sy <= (float)kGatherSpanOk) &&
             (sx * grow <= (float)(kGatherSpan - 1)) && (sy * grow <= (float)(kGatherSpan - 1));   // (false for NaN)
      }
  }
  p.regular = ok ? 1.0f : 0.0f;
  p.pad[0] = p.pad[1] = 0.0f;
  prep[i] = p;
  if (!ok) atomicOr(&flags[0], 1);
}

__global__ void gather_clear_flags_kernel(int* __restrict__ flags) { flags[threadIdx.x] = 0; }

// ---------------------------------------------------------------------------------------------------------------
// pass 2
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGatherTileW = 64, kGatherTileH = 2 * (kBlock / 32);   // 32 x 8 threads of 2 x 2 source pixels

// bilinear weights of target sample (ix, iy) on the four pixels of the block at (sx, sy): torch's own expressions
struct BlockWeights { float w00, w01, w10, w11; };
__device__ __forceinline__ BlockWeights block_weights(float ix, float iy, float fsx, float fsy) {
  const float xf = floorf(ix), yf = floorf(iy);
  const float wl = (xf + 1.0f) - ix, wr = ix - xf, wu = (yf + 1.0f) - iy, wd = iy - yf;   // taps xf, xf+1 / yf, yf+1
  // column sx is tap xf (weight wl) or tap xf + 1 (weight wr); column sx + 1 likewise
  const float a0 = (xf == fsx) ? wl : ((xf + 1.0f == fsx) ? wr : 0.0f);
  const float a1 = (xf == fsx + 1.0f) ? wl : ((xf == fsx) ? wr : 0.0f);
  const float b0 = (yf == fsy) ? wu : ((yf + 1.0f == fsy) ? wd : 0.0f);
  const float b1 = (yf == fsy + 1.0f) ? wu : ((yf == fsy) ? wd : 0.0f);
  BlockWeights r;
  r.w00 = a0 * b0; r.w01 = a1 * b0; r.w10 = a0 * b1; r.w11 = a1 * b1;
  return r;
}

template <bool MIX>
__global__ __launch_bounds__(kBlock) void gather_bwd_pass2_kernel(SweepArgs a, const float* __restrict__ tmp,
                                                                  const GatherPrep* __restrict__ prep,
                                                                  float* __restrict__ g_logits, float* __restrict__ g_sigma,
                                                                  int* __restrict__ flags, int tiles_x, int accumulate) {
  typedef typename std::conditional<MIX, float2, float>::type Elem;
  const int HW = a.H * a.W, N = a.N, W = a.W, H = a.H;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int blk = xcd_banded(blockIdx.x, gridDim.x);
  const int tyi = blk / tiles_x, txi = blk - tyi * tiles_x;
  const int sx = txi * kGatherTileW + 2 * (tid & 31), sy = tyi * kGatherTileH + 2 * (tid >> 5);
  if (sx >= W || sy >= H) return;
  const bool right = sx + 1 < W, down = sy + 1 < H;
  const bool pair = right && ((W & 1) == 0);   // both pixels of a row as one aligned 8-byte access
  const CoordNorm cn = make_coord_norm(W, H);
  const Elem* __restrict__ tmp_b = reinterpret_cast<const Elem*>(tmp) + (long)b * N * HW;
  const long s_off = (long)b * N * HW + (long)sy * W + sx;
  float* gl = g_logits ? g_logits + s_off : nullptr;
  float* gs = (MIX && g_sigma) ? g_sigma + s_off : nullptr;
  const float fsx = (float)sx, fsy = (float)sy;
  auto put = [&](float* p, long base, float v00, float v01, float v10, float v11) {
    if (!p) return;
    if (pair) {
      *reinterpret_cast<float2*>(p + base) = make_float2(v00, v01);
      if (down) *reinterpret_cast<float2*>(p + base + W) = make_float2(v10, v11);
    } else {
      p[base] = v00;
      if (right) p[base + 1] = v01;
      if (down) { p[base + W] = v10; if (right) p[base + W + 1] = v11; }
    }
  };
  auto get = [&](const float* p, long base, float& v00, float& v01, float& v10, float& v11) {
    if (!p) return;
    v00 = p[base];
    if (right) v01 = p[base + 1];
    if (down) { v10 = p[base + W]; if (right) v11 = p[base + W + 1]; }
  };
  bool cut = false;
  for (int n = 0; n < N; ++n) {
    const GatherPrep* pr = prep + (long)b * N + n;   // workgroup-uniform
    const long base = (long)n * HW;
    if (pr->regular == 0.0f) {   // the fix-up kernel adds this plane's gradient with atomics
      if (!accumulate) { put(gl, base, 0.0f, 0.0f, 0.0f, 0.0f); put(gs, base, 0.0f, 0.0f, 0.0f, 0.0f); }
      continue;
    }
    float l00 = 0.0f, l01 = 0.0f, l10 = 0.0f, l11 = 0.0f, s00 = 0.0f, s01 = 0.0f, s10 = 0.0f, s11 = 0.0f;
    if (accumulate) {   // requested ahead of the window scan
      get(gl, base, l00, l01, l10, l11);
      get(gs, base, s00, s01, s10, s11);
    }
    const GatherWindow win = gather_window(pr->Hs, fsx, fsy);
    const int x0 = (int)fminf(fmaxf(win.x0, 0.0f), (float)W), y0 = (int)fminf(fmaxf(win.y0, 0.0f), (float)H);
    int x1 = (int)fminf(fmaxf(win.x1, -1.0f), (float)(W - 1)), y1 = (int)fminf(fmaxf(win.y1, -1.0f), (float)(H - 1));
    if (x1 - x0 >= kGatherSpan) { x1 = x0 + kGatherSpan - 1; cut = true; }
    if (y1 - y0 >= kGatherSpan) { y1 = y0 + kGatherSpan - 1; cut = true; }
    // One flat walk over the window: a wave runs as many steps as its largest window has points.  The scratch element of
    // a candidate is requested one step ahead and unconditionally (its address needs the window only): read where it is
    // used, behind the weight test, every step waited for a memory round trip of its own.
    const int cnt = (x1 >= x0 && y1 >= y0) ? (x1 - x0 + 1) * (y1 - y0 + 1) : 0;
    int tx = x0, ty = y0;
    Elem vnext = Elem();
    if (cnt > 0) vnext = tmp_b[base + y0 * W + x0];
    for (int j = 0; j < cnt; ++j) {
      const Elem v = vnext;
      int nx = tx + 1, ny = ty;
      if (nx > x1) { nx = x0; ++ny; }
      if (j + 1 < cnt) vnext = tmp_b[base + ny * W + nx];
      bool mk;
      const PlaneGeom g = plane_coords<PD_WARP_HOMOGRAPHY>(a, cn, b, n, tx, ty, 0.0f, mk);
      const BlockWeights w = block_weights(g.ix, g.iy, fsx, fsy);
      if ((w.w00 != 0.0f) | (w.w01 != 0.0f) | (w.w10 != 0.0f) | (w.w11 != 0.0f)) {   // (masked planes of a pixel are zeros in the scratch)
        float vl, vs = 0.0f;
        if constexpr (MIX) { vl = v.x; vs = v.y; } else vl = v;
        l00 += w.w00 * vl; l01 += w.w01 * vl; l10 += w.w10 * vl; l11 += w.w11 * vl;
        if (MIX) { s00 += w.w00 * vs; s01 += w.w01 * vs; s10 += w.w10 * vs; s11 += w.w11 * vs; }
      }
      tx = nx; ty = ny;
    }
    put(gl, base, l00, l01, l10, l11);
    put(gs, base, s00, s01, s10, s11);
  }
  if (cut) atomicOr(&flags[1], 1);
}

// ---------------------------------------------------------------------------------------------------------------
// pass 2 with the scratch staged through LDS
// ---------------------------------------------------------------------------------------------------------------
// In the kernel above every candidate costs a scattered 8-byte gather (0.21 of its 0.565 ms at 8x49x192x640).  The blocks
// of a workgroup's 64 x 16 tile share their candidates: the pre-image of the tile's footprint is a convex quadrilateral
// whose bounding box is that of the four CORNER blocks' windows.  Per plane the corner threads publish their windows,
// the workgroup copies the box of the scratch (~73 x 21 elements) to LDS with coalesced loads — requested one plane
// ahead, parked after the current plane's walk, one barrier per plane, two alternating buffers — and the candidates
// become ds_reads.  A box beyond the buffer (24 rows, 128 columns, 2304 elements: magnification > ~1.4) makes the
// workgroup walk that plane with the direct gathers (workgroup-uniform).
constexpr int kGStageRows = 24, kGStageCols = 128, kGStageMax = 2304;   // 2 x 18 KB: four workgroups per CU
constexpr int kGPre = (kGStageRows / (kBlock / kWave)) * (kGStageCols / kWave);   // staged elements per thread and plane

template <bool MIX>
__global__ __launch_bounds__(kBlock) void gather_bwd_pass2_staged_kernel(SweepArgs a, const float* __restrict__ tmp,
                                                                         const GatherPrep* __restrict__ prep,
                                                                         float* __restrict__ g_logits, float* __restrict__ g_sigma,
                                                                         int* __restrict__ flags, int tiles_x, int accumulate) {
  typedef typename std::conditional<MIX, float2, float>::type Elem;
  __shared__ Elem buf[2][kGStageMax];
  __shared__ int corner[2][4][4];   // per buffer: the four corner blocks' windows (x0, y0, x1, y1)
  const int HW = a.H * a.W, N = a.N, W = a.W, H = a.H;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int blk = xcd_banded(blockIdx.x, gridDim.x);
  const int tyi = blk / tiles_x, txi = blk - tyi * tiles_x;
  const int tile_x0 = txi * kGatherTileW, tile_y0 = tyi * kGatherTileH;
  const int lx = tid & 31, ly = tid >> 5, wv = tid >> 6, ln = tid & (kWave - 1);
  const int lx_last = min(31, (W - 1 - tile_x0) >> 1), ly_last = min(kBlock / 32 - 1, (H - 1 - tile_y0) >> 1);
  const bool has_s = lx <= lx_last && ly <= ly_last;
  const int sx = tile_x0 + 2 * lx, sy = tile_y0 + 2 * ly;
  const bool right = sx + 1 < W, down = sy + 1 < H;
  const bool pair = right && ((W & 1) == 0);
  const CoordNorm cn = make_coord_norm(W, H);
  const Elem* __restrict__ tmp_b = reinterpret_cast<const Elem*>(tmp) + (long)b * N * HW;
  const long s_off = (long)b * N * HW + (long)sy * W + sx;
  float* gl = (has_s && g_logits) ? g_logits + s_off : nullptr;
  float* gs = (has_s && MIX && g_sigma) ? g_sigma + s_off : nullptr;
  const float fsx = (float)sx, fsy = (float)sy;
  const GatherPrep* prep_b = prep + (long)b * N;
  auto put = [&](float* p, long base, float v00, float v01, float v10, float v11) {
    if (!p) return;
    if (pair) {
      *reinterpret_cast<float2*>(p + base) = make_float2(v00, v01);
      if (down) *reinterpret_cast<float2*>(p + base + W) = make_float2(v10, v11);
    } else {
      p[base] = v00;
      if (right) p[base + 1] = v01;
      if (down) { p[base + W] = v10; if (right) p[base + W + 1] = v11; }
    }
  };
  auto get = [&](const float* p, long base, float& v00, float& v01, float& v10, float& v11) {
    if (!p) return;
    v00 = p[base];
    if (right) v01 = p[base + 1];
    if (down) { v10 = p[base + W]; if (right) v11 = p[base + W + 1]; }
  };
  struct Win { int x0, y0, x1, y1; };
  bool cut = false;
  // window of this thread's block on plane n, clamped to the image (empty: x1 < x0); irregular planes: empty
  auto window_of = [&](int n) {
    Win w; w.x0 = 0; w.y0 = 0; w.x1 = -1; w.y1 = -1;
    if (n < N && prep_b[n].regular != 0.0f) {
      const GatherWindow g = gather_window(prep_b[n].Hs, fsx, fsy);
      w.x0 = (int)fminf(fmaxf(g.x0, 0.0f), (float)W); w.y0 = (int)fminf(fmaxf(g.y0, 0.0f), (float)H);
      w.x1 = (int)fminf(fmaxf(g.x1, -1.0f), (float)(W - 1)); w.y1 = (int)fminf(fmaxf(g.y1, -1.0f), (float)(H - 1));
      if (w.x1 - w.x0 >= kGatherSpan) { w.x1 = w.x0 + kGatherSpan - 1; cut = true; }
      if (w.y1 - w.y0 >= kGatherSpan) { w.y1 = w.y0 + kGatherSpan - 1; cut = true; }
    }
    return w;
  };
  auto publish = [&](int slot, const Win& w) {   // (a thread can be several corners of a tile that is one block wide / high)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lx == ((q & 1) ? lx_last : 0) && ly == ((q & 2) ? ly_last : 0)) {
        corner[slot][q][0] = w.x0; corner[slot][q][1] = w.y0; corner[slot][q][2] = w.x1; corner[slot][q][3] = w.y1;
      }
  };
  struct Box { int x0, y0, bw, bh; bool staged; };
  auto box_of = [&](int slot) {
    int x0 = corner[slot][0][0], y0 = corner[slot][0][1], x1 = corner[slot][0][2], y1 = corner[slot][0][3];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      x0 = min(x0, corner[slot][q][0]); y0 = min(y0, corner[slot][q][1]);
      x1 = max(x1, corner[slot][q][2]); y1 = max(y1, corner[slot][q][3]);
    }
    Box bx;
    bx.x0 = __builtin_amdgcn_readfirstlane(x0); bx.y0 = __builtin_amdgcn_readfirstlane(y0);
    bx.bw = __builtin_amdgcn_readfirstlane(max(x1 - x0 + 1, 0)); bx.bh = __builtin_amdgcn_readfirstlane(max(y1 - y0 + 1, 0));
    bx.staged = bx.bw <= kGStageCols && bx.bh <= kGStageRows && bx.bw * bx.bh <= kGStageMax;
    return bx;
  };
  Elem pre[kGPre];
  auto issue = [&](int n, const Box& bx) {   // thread (wave wv, lane ln): rows wv, wv + 4, ...; columns ln, ln + 64
    if (!bx.staged) return;
    const Elem* src = tmp_b + (long)n * HW + (long)bx.y0 * W + bx.x0;
#pragma unroll
    for (int k = 0; k < kGPre; ++k) {
      const int r = wv + (kBlock / kWave) * (k >> 1), c = ln + kWave * (k & 1);
      Elem v = Elem();
      if (r < bx.bh && c < bx.bw) v = src[(long)r * W + c];
      pre[k] = v;
    }
  };
  auto park = [&](int which, const Box& bx) {
    if (!bx.staged) return;
#pragma unroll
    for (int k = 0; k < kGPre; ++k) {
      const int r = wv + (kBlock / kWave) * (k >> 1), c = ln + kWave * (k & 1);
      if (r < bx.bh && c < bx.bw) buf[which][r * bx.bw + c] = pre[k];
    }
  };

  Win w_cur = has_s ? window_of(0) : window_of(N);
  publish(0, w_cur);
  __syncthreads();
  Box box_cur = box_of(0);
  issue(0, box_cur);
  park(0, box_cur);
  Win w_nxt = has_s ? window_of(1) : window_of(N);
  publish(1, w_nxt);
  for (int n = 0; n < N; ++n) {
    __syncthreads();   // buf[n & 1] complete, the corner windows of plane n + 1 visible; nobody reads buf[(n + 1) & 1] any more
    Box box_nxt; box_nxt.x0 = box_nxt.y0 = box_nxt.bw = box_nxt.bh = 0; box_nxt.staged = false;
    if (n + 1 < N) { box_nxt = box_of((n + 1) & 1); issue(n + 1, box_nxt); }
    const long base = (long)n * HW;
    if (prep_b[n].regular == 0.0f) {   // the fix-up kernel adds this plane's gradient with atomics
      if (!accumulate) { put(gl, base, 0.0f, 0.0f, 0.0f, 0.0f); put(gs, base, 0.0f, 0.0f, 0.0f, 0.0f); }
    } else {
      float l00 = 0.0f, l01 = 0.0f, l10 = 0.0f, l11 = 0.0f, s00 = 0.0f, s01 = 0.0f, s10 = 0.0f, s11 = 0.0f;
      if (accumulate) { get(gl, base, l00, l01, l10, l11); get(gs, base, s00, s01, s10, s11); }
      const int x0 = w_cur.x0, y0 = w_cur.y0, x1 = w_cur.x1, y1 = w_cur.y1;
      const int cnt = (x1 >= x0 && y1 >= y0) ? (x1 - x0 + 1) * (y1 - y0 + 1) : 0;
      int tx = x0, ty = y0;
      const Elem* lds = buf[n & 1];
      const int lbase = -box_cur.y0 * box_cur.bw - box_cur.x0;
      Elem vnext = Elem();
      if (!box_cur.staged && cnt > 0) vnext = tmp_b[base + y0 * W + x0];
      for (int j = 0; j < cnt; ++j) {
        Elem v;
        int nx = tx + 1, ny = ty;
        if (nx > x1) { nx = x0; ++ny; }
        if (box_cur.staged) {
          v = lds[lbase + ty * box_cur.bw + tx];
        } else {
          v = vnext;
          if (j + 1 < cnt) vnext = tmp_b[base + ny * W + nx];
        }
        bool mk;
        const PlaneGeom g = plane_coords<PD_WARP_HOMOGRAPHY>(a, cn, b, n, tx, ty, 0.0f, mk);
        const BlockWeights w = block_weights(g.ix, g.iy, fsx, fsy);
        if ((w.w00 != 0.0f) | (w.w01 != 0.0f) | (w.w10 != 0.0f) | (w.w11 != 0.0f)) {
          float vl, vs = 0.0f;
          if constexpr (MIX) { vl = v.x; vs = v.y; } else vl = v;
          l00 += w.w00 * vl; l01 += w.w01 * vl; l10 += w.w10 * vl; l11 += w.w11 * vl;
          if (MIX) { s00 += w.w00 * vs; s01 += w.w01 * vs; s10 += w.w10 * vs; s11 += w.w11 * vs; }
        }
        tx = nx; ty = ny;
      }
      put(gl, base, l00, l01, l10, l11);
      put(gs, base, s00, s01, s10, s11);
    }
    if (n + 1 < N) park((n + 1) & 1, box_nxt);
    w_cur = w_nxt; box_cur = box_nxt;
    if (n + 2 < N) {
      w_nxt = has_s ? window_of(n + 2) : window_of(N);
      publish(n & 1, w_nxt);
    }
  }
  if (cut) atomicOr(&flags[1], 1);
}

// ---------------------------------------------------------------------------------------------------------------
// irregular planes: scatter out of the scratch with atomics (target-anchored; returns at once when there are none)
// ---------------------------------------------------------------------------------------------------------------
template <bool MIX>
__global__ __launch_bounds__(kBlock) void gather_fixup_kernel(SweepArgs a, const float* __restrict__ tmp,
                                                              const GatherPrep* __restrict__ prep,
                                                              float* __restrict__ g_logits, float* __restrict__ g_sigma,
                                                              const int* __restrict__ flags) {
  if (flags[0] == 0) return;
  typedef typename std::conditional<MIX, float2, float>::type Elem;
  const int HW = a.H * a.W, N = a.N, W = a.W, H = a.H;
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= HW) return;
  const int y = pix / W, x = pix - y * W;
  const CoordNorm cn = make_coord_norm(W, H);
  const Elem* __restrict__ tmp_b = reinterpret_cast<const Elem*>(tmp) + (long)b * N * HW;
  for (int n = 0; n < N; ++n) {
    if (prep[(long)b * N + n].regular != 0.0f) continue;
    const Elem v = tmp_b[(long)n * HW + pix];
    float vl, vs = 0.0f;
    if constexpr (MIX) { vl = v.x; vs = v.y; } else vl = v;
    if (vl == 0.0f && vs == 0.0f) continue;   // masked or without gradient
    bool mk;
    const PlaneGeom g = plane_coords<PD_WARP_HOMOGRAPHY>(a, cn, b, n, x, y, 0.0f, mk);
    const Tap t = make_tap(g.ix, g.iy, W, H);
    const long pl = ((long)b * N + n) * HW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool right = k & 1, down = k & 2;
      const bool valid = (right ? t.vx1 : t.vx0) && (down ? t.vy1 : t.vy0);
      const float w = (right ? t.wx1 : t.wx0) * (down ? t.wy1 : t.wy0);
      if (valid && w != 0.0f) {
        const long e = pl + (long)(t.y0 + (down ? 1 : 0)) * W + (t.x0 + (right ? 1 : 0));
        if (g_logits && vl != 0.0f) unsafeAtomicAdd(g_logits + e, w * vl);
        if (MIX && g_sigma && vs != 0.0f) unsafeAtomicAdd(g_sigma + e, w * vs);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static size_t galign4(size_t floats) { return (floats + 3) & ~(size_t)3; }

bool gather_bwd_applicable(const pd_sweep_desc* d) {
  return d->mode == PD_WARP_HOMOGRAPHY && !(d->flags & PD_HOMO_UNIFORM) &&
         (d->impl == PD_IMPL_AUTO || d->impl == PD_IMPL_UNIFORM_DIRECT || d->impl == PD_IMPL_FAST_ROWS || d->impl == PD_IMPL_EXACT_ROWS);
}

// workspace: partial sums [B][nblk][N*9] | GatherPrep[B*N] | flags (4 ints) | scratch [B][N][H*W] (x2 with PD_MIXTURE)
size_t gather_bwd_workspace_floats(const pd_sweep_desc* d) {
  const size_t nblk = (size_t)ceil_div(d->H * d->W, kBlock);
  const size_t per = (d->flags & PD_MIXTURE) ? 2 : 1;
  return galign4((size_t)d->B * nblk * d->N * 9) + galign4((size_t)d->B * d->N * kGatherPrepFloats) + 4 +
         per * (size_t)d->B * d->N * d->H * d->W + 8;
}

GatherPlan gather_bwd_plan(const pd_sweep_desc* d, float* workspace) {
  GatherPlan gp;
  const uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15;
  gp.nblk = ceil_div(d->H * d->W, kBlock);
  gp.partials = reinterpret_cast<float*>(base);
  float* prep = gp.partials + galign4((size_t)d->B * gp.nblk * d->N * 9);
  gp.prep = prep;
  gp.flags = reinterpret_cast<int*>(prep + galign4((size_t)d->B * d->N * kGatherPrepFloats));
  gp.scratch = reinterpret_cast<float*>(gp.flags) + 4;
  return gp;
}

int gather_bwd_prepare(const pd_sweep_desc* d, const SweepArgs& a, const GatherPlan& gp, hipStream_t stream) {
  gather_clear_flags_kernel<<<1, 4, 0, stream>>>(gp.flags);
  const int BN = d->B * d->N;
  gather_prep_kernel<<<ceil_div(BN, 64), 64, 0, stream>>>(a.plane, reinterpret_cast<GatherPrep*>(gp.prep), gp.flags, BN, d->W, d->H);
  return check_launch("gather_prep_kernel");
}

int gather_bwd_finish(const pd_sweep_desc* d, const SweepArgs& a, const BwdOut& o, const GatherPlan& gp, hipStream_t stream) {
  const bool mix = (d->flags & PD_MIXTURE) != 0;
  const int accumulate = (d->flags & PD_BWD_ACCUMULATE) ? 1 : 0;
  const int tiles_x = ceil_div(d->W, kGatherTileW);
  const dim3 grid2(tiles_x * ceil_div(d->H, kGatherTileH), d->B), grid1(ceil_div(d->H * d->W, kBlock), d->B);
  const GatherPrep* prep = reinterpret_cast<const GatherPrep*>(gp.prep);
  const bool staged = d->impl != PD_IMPL_UNIFORM_DIRECT;   // (the direct-gather form: cross-check, and what large boxes fall back to)
  if (mix) {
    if (staged) gather_bwd_pass2_staged_kernel<true><<<grid2, kBlock, 0, stream>>>(a, gp.scratch, prep, o.g_logits, o.g_sigma, gp.flags, tiles_x, accumulate);
    else gather_bwd_pass2_kernel<true><<<grid2, kBlock, 0, stream>>>(a, gp.scratch, prep, o.g_logits, o.g_sigma, gp.flags, tiles_x, accumulate);
    gather_fixup_kernel<true><<<grid1, kBlock, 0, stream>>>(a, gp.scratch, prep, o.g_logits, o.g_sigma, gp.flags);
  } else {
    if (staged) gather_bwd_pass2_staged_kernel<false><<<grid2, kBlock, 0, stream>>>(a, gp.scratch, prep, o.g_logits, nullptr, gp.flags, tiles_x, accumulate);
    else gather_bwd_pass2_kernel<false><<<grid2, kBlock, 0, stream>>>(a, gp.scratch, prep, o.g_logits, nullptr, gp.flags, tiles_x, accumulate);
    gather_fixup_kernel<false><<<grid1, kBlock, 0, stream>>>(a, gp.scratch, prep, o.g_logits, nullptr, gp.flags);
  }
  return check_launch("gather_bwd_pass2_kernel / gather_fixup_kernel");
}

}  // namespace pd
