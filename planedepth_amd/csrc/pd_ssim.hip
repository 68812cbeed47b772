// SSIM (3x3 box window, reflection padding) and the 0.85*SSIM + 0.15*L1 reprojection loss, forward and backward.
// Replaces layers.py:276-306 (SSIM) and trainer.py:687-699 (compute_reprojection_loss): 2 reflection pads,
// 5 avg-pools and ~20 elementwise launches in the reference; one stencil kernel each way here.
//
// Backward structure: out(p) depends on the 3x3 window sums around p, so d out(p)/d x(q) for q in window(p) is
//   a(p) + b(p) * x(q) + c(p) * y(q)      with per-pixel coefficients a,b,c (derivatives w.r.t. mean, E[x^2], E[xy]).
// The gradient at q gathers those coefficients from every p whose (reflected) window contains q.  Reflection makes
// a border pixel appear more than once in a window; the gather loops over the 3x3 *offsets* of each neighbour p and
// counts every hit, which reproduces reflection_pad2d_backward + avg_pool2d_backward exactly.
#include "pd_common.h"

namespace pd {

constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

__device__ __forceinline__ int reflect(int i, int n) {  // ReflectionPad2d(1): -1 -> 1, n -> n-2
  return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
}

struct SsimStats {
  float mu_x, mu_y, sxx, syy, sxy;  // window means of x, y, x^2, y^2, xy
};

__device__ __forceinline__ SsimStats window_stats(const float* __restrict__ x, const float* __restrict__ y, int px,
                                                  int py, int H, int W) {
  float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = reflect(py + dy, H);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = reflect(px + dx, W);
      const float a = x[yy * W + xx], b = y[yy * W + xx];
      sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
    }
  }
  const float k = 1.0f / 9.0f;
  return {sx * k, sy * k, sxx * k, syy * k, sxy * k};
}

__device__ __forceinline__ float ssim_value(const SsimStats& s) {
  const float sigma_x = s.sxx - s.mu_x * s.mu_x, sigma_y = s.syy - s.mu_y * s.mu_y;
  const float sigma_xy = s.sxy - s.mu_x * s.mu_y;
  const float n = (2.0f * s.mu_x * s.mu_y + kC1) * (2.0f * sigma_xy + kC2);
  const float d = (s.mu_x * s.mu_x + s.mu_y * s.mu_y + kC1) * (sigma_x + sigma_y + kC2);
  return fminf(fmaxf((1.0f - n / d) * 0.5f, 0.0f), 1.0f);
}

// Coefficients of d out / d(window entries): g * [ (ax + bx*x + c*y) for x-entries, (ay + by*y + c*x) for y-entries ] / 9
struct SsimCoef {
  float ax, bx, ay, by, c;
};

__device__ __forceinline__ SsimCoef ssim_coef(const SsimStats& s, float g) {
  const float mx = s.mu_x, my = s.mu_y;
  const float sigma_x = s.sxx - mx * mx, sigma_y = s.syy - my * my, sigma_xy = s.sxy - mx * my;
  const float n1 = 2.0f * mx * my + kC1, n2 = 2.0f * sigma_xy + kC2;
  const float d1 = mx * mx + my * my + kC1, d2 = sigma_x + sigma_y + kC2;
  const float n = n1 * n2, d = d1 * d2;
  const float v = (1.0f - n / d) * 0.5f;
  SsimCoef k = {0, 0, 0, 0, 0};
  if (!(v >= 0.0f && v <= 1.0f)) return k;  // clamp(.,0,1) blocks the gradient outside [0,1] (inclusive inside)
  const float gn = -0.5f * g / d;           // d out / d n
  const float gd = 0.5f * g * n / (d * d);  // d out / d d
  const float g_n1 = gn * n2, g_n2 = gn * n1, g_d1 = gd * d2, g_d2 = gd * d1;
  // through sigma_x = E[x^2] - mu_x^2 etc.
  const float g_sxx = g_d2, g_syy = g_d2, g_sxy = 2.0f * g_n2;
  const float g_mx = g_n1 * 2.0f * my + g_d1 * 2.0f * mx - g_d2 * 2.0f * mx - g_sxy * my;
  const float g_my = g_n1 * 2.0f * mx + g_d1 * 2.0f * my - g_d2 * 2.0f * my - g_sxy * mx;
  const float k9 = 1.0f / 9.0f;
  k.ax = g_mx * k9; k.bx = 2.0f * g_sxx * k9;
  k.ay = g_my * k9; k.by = 2.0f * g_syy * k9;
  k.c = g_sxy * k9;
  return k;
}

// ---- tiled backward ----------------------------------------------------------------------------------------------
// One workgroup owns a 32x8 tile of one image plane.  The per-centre coefficients are computed ONCE per centre (tile +
// 1 ring, from an LDS copy of the inputs with a 2-ring halo filled through the reflection) and then gathered by the
// pixels — instead of every pixel recomputing the window statistics and coefficients of its nine centres from global
// memory (162 loads and two divisions x 9 per pixel: 141 us for 8x3x192x640; this form: see NOTEBOOK.md §3.3).
constexpr int kTileW = 32, kTileH = 8;
static_assert(kTileW * kTileH == kBlock, "one thread per tile pixel");
constexpr int kInW = kTileW + 4, kInH = kTileH + 4;     // inputs: 2-ring halo
constexpr int kCoW = kTileW + 2, kCoH = kTileH + 2;     // centres: 1-ring halo

struct SsimTile {
  float x[kInH][kInW], y[kInH][kInW];
  float ax[kCoH][kCoW], bx[kCoH][kCoW], c[kCoH][kCoW], ay[kCoH][kCoW], by[kCoH][kCoW];
};

__device__ __forceinline__ int reflect_clamped(int i, int n) { return min(max(reflect(i, n), 0), n - 1); }

// Gradient of sum_p g(p) * ssim_out(p) w.r.t. x(q) (and y(q)) for this thread's pixel q = (tile origin + thread).
// `g` = upstream gradient plane, `scale` multiplies it.  All threads of the workgroup must call (barriers inside).
template <bool WANT_Y>
__device__ __forceinline__ void ssim_grad_tile(SsimTile& t, int H, int W, const float* __restrict__ x,
                                               const float* __restrict__ y, const float* __restrict__ g, float scale,
                                               int tx0, int ty0, float& gx, float& gy) {
  for (int i = threadIdx.x; i < kInH * kInW; i += kBlock) {   // inputs, reflected at the image border
    const int r = i / kInW, cc = i - r * kInW;
    const int sy = reflect_clamped(ty0 - 2 + r, H), sx = reflect_clamped(tx0 - 2 + cc, W);
    t.x[r][cc] = x[sy * W + sx];
    t.y[r][cc] = y[sy * W + sx];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCoH * kCoW; i += kBlock) {   // coefficients of every centre that exists
    const int r = i / kCoW, cc = i - r * kCoW;
    const int py = ty0 - 1 + r, px = tx0 - 1 + cc;
    SsimCoef k = {0, 0, 0, 0, 0};
    if (py >= 0 && py < H && px >= 0 && px < W) {
      float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {   // same summation order as window_stats (row-major over the window)
          const float a = t.x[r + dy][cc + dx], b = t.y[r + dy][cc + dx];
          sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
      const float k9 = 1.0f / 9.0f;
      const SsimStats st = {sx * k9, sy * k9, sxx * k9, syy * k9, sxy * k9};
      k = ssim_coef(st, scale * g[py * W + px]);
    }
    t.ax[r][cc] = k.ax; t.bx[r][cc] = k.bx; t.c[r][cc] = k.c;
    if (WANT_Y) { t.ay[r][cc] = k.ay; t.by[r][cc] = k.by; }
  }
  __syncthreads();
  gx = gy = 0.0f;
  const int lx = threadIdx.x & (kTileW - 1), ly = threadIdx.x / kTileW;
  const int qx = tx0 + lx, qy = ty0 + ly;
  if (qx >= W || qy >= H) return;
  const float xq = t.x[ly + 2][lx + 2], yq = t.y[ly + 2][lx + 2];
  // every centre p within one pixel of q, times the number of its window offsets that land on q (reflection makes a
  // border pixel appear more than once in a window): reflection_pad2d_backward + avg_pool2d_backward, exactly
#pragma unroll
  for (int oy = -1; oy <= 1; ++oy) {
    const int py = qy + oy;
    if (py < 0 || py >= H) continue;
    int cy = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) cy += (reflect(py + dy, H) == qy);
#pragma unroll
    for (int ox = -1; ox <= 1; ++ox) {
      const int px = qx + ox;
      if (px < 0 || px >= W) continue;
      int cx = 0;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) cx += (reflect(px + dx, W) == qx);
      const float mult = (float)(cx * cy);
      const int r = ly + 1 + oy, cc = lx + 1 + ox;
      gx += mult * (t.ax[r][cc] + t.bx[r][cc] * xq + t.c[r][cc] * yq);
      if (WANT_Y) gy += mult * (t.ay[r][cc] + t.by[r][cc] * yq + t.c[r][cc] * xq);
    }
  }
}

__global__ __launch_bounds__(kBlock) void ssim_fwd_kernel(int H, int W, const float* __restrict__ x,
                                                          const float* __restrict__ y, float* __restrict__ out) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= H * W) return;
  const long img = (long)blockIdx.y * H * W;
  const int py = pix / W, px = pix - py * W;
  out[img + pix] = ssim_value(window_stats(x + img, y + img, px, py, H, W));
}

// g_x(q) = sum over p, over offsets (dy,dx) with reflect(p+offset) == q, of  ax(p) + bx(p)*x(q) + c(p)*y(q)
// grid (tiles_x, tiles_y, B*C)
template <bool WANT_Y>
__global__ __launch_bounds__(kBlock) void ssim_bwd_kernel(int H, int W, const float* __restrict__ x,
                                                          const float* __restrict__ y, const float* __restrict__ g_out,
                                                          float* __restrict__ g_x, float* __restrict__ g_y) {
  __shared__ SsimTile tile;
  const long img = (long)blockIdx.z * H * W;
  const int tx0 = blockIdx.x * kTileW, ty0 = blockIdx.y * kTileH;
  float gx, gy;
  ssim_grad_tile<WANT_Y>(tile, H, W, x + img, y + img, g_out + img, 1.0f, tx0, ty0, gx, gy);
  const int qx = tx0 + (threadIdx.x & (kTileW - 1)), qy = ty0 + threadIdx.x / kTileW;
  if (qx >= W || qy >= H) return;
  if (g_x) g_x[img + qy * W + qx] = gx;
  if (WANT_Y && g_y) g_y[img + qy * W + qx] = gy;
}

// loss[b,0,p] = use_ssim ? 0.85 * mean_c ssim + 0.15 * mean_c |t - p| : mean_c |t - p|     (trainer.py:690-697)
__global__ __launch_bounds__(kBlock) void reproj_fwd_kernel(int H, int W, int use_ssim, const float* __restrict__ pred,
                                                            const float* __restrict__ tgt, float* __restrict__ loss) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  const int HW = H * W;
  if (pix >= HW) return;
  const int b = blockIdx.y;
  const int py = pix / W, px = pix - py * W;
  float l1 = 0.0f, ss = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const long img = ((long)b * 3 + c) * HW;
    l1 += fabsf(tgt[img + pix] - pred[img + pix]);
    if (use_ssim) ss += ssim_value(window_stats(pred + img, tgt + img, px, py, H, W));
  }
  l1 /= 3.0f;
  loss[(long)b * HW + pix] = use_ssim ? 0.85f * (ss / 3.0f) + 0.15f * l1 : l1;
}

// grid (tiles_x, tiles_y, B*3)
template <bool WANT_T>
__global__ __launch_bounds__(kBlock) void reproj_bwd_kernel(int H, int W, int use_ssim, const float* __restrict__ pred,
                                                            const float* __restrict__ tgt,
                                                            const float* __restrict__ g_loss,
                                                            float* __restrict__ g_pred, float* __restrict__ g_tgt) {
  __shared__ SsimTile tile;
  const int HW = H * W;
  const int b = blockIdx.z / 3;
  const long img = (long)blockIdx.z * HW;   // plane (b, c) of [B,3,H,W]
  const float* gl = g_loss + (long)b * HW;
  const int tx0 = blockIdx.x * kTileW, ty0 = blockIdx.y * kTileH;
  const float wl1 = (use_ssim ? 0.15f : 1.0f) / 3.0f, wss = 0.85f / 3.0f;
  float gx = 0.0f, gy = 0.0f;
  if (use_ssim) ssim_grad_tile<WANT_T>(tile, H, W, pred + img, tgt + img, gl, wss, tx0, ty0, gx, gy);   // uniform branch
  const int qx = tx0 + (threadIdx.x & (kTileW - 1)), qy = ty0 + threadIdx.x / kTileW;
  if (qx >= W || qy >= H) return;
  const int pix = qy * W + qx;
  // L1 part: d|t - p|/dp = -sgn(t - p)
  const float l1 = -wl1 * gl[pix] * sgn(tgt[img + pix] - pred[img + pix]);
  g_pred[img + pix] = gx + l1;
  if (WANT_T) g_tgt[img + pix] = gy - l1;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_ssim_fwd(int B, int C, int H, int W, const float* x, const float* y, float* out, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2, "bad shape");
  PD_REQUIRE(x && y && out, "NULL pointer");
  ssim_fwd_kernel<<<dim3(ceil_div(H * W, kBlock), B * C), kBlock, 0, (hipStream_t)stream>>>(H, W, x, y, out);
  return check_launch("ssim_fwd_kernel");
}

extern "C" int pd_ssim_bwd(int B, int C, int H, int W, const float* x, const float* y, const float* g_out, float* g_x,
                           float* g_y, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && C > 0 && H >= 2 && W >= 2, "bad shape");
  PD_REQUIRE(x && y && g_out && (g_x || g_y), "NULL pointer");
  PD_REQUIRE((long)B * C <= 65535, "too many image planes for one launch");
  dim3 grid(ceil_div(W, kTileW), ceil_div(H, kTileH), B * C);
  if (g_y) ssim_bwd_kernel<true><<<grid, kBlock, 0, (hipStream_t)stream>>>(H, W, x, y, g_out, g_x, g_y);
  else     ssim_bwd_kernel<false><<<grid, kBlock, 0, (hipStream_t)stream>>>(H, W, x, y, g_out, g_x, g_y);
  return check_launch("ssim_bwd_kernel");
}

extern "C" int pd_reproj_loss_fwd(int B, int H, int W, int use_ssim, const float* pred, const float* target,
                                  float* loss, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && H >= 2 && W >= 2, "bad shape");
  PD_REQUIRE(pred && target && loss, "NULL pointer");
  reproj_fwd_kernel<<<dim3(ceil_div(H * W, kBlock), B), kBlock, 0, (hipStream_t)stream>>>(H, W, use_ssim, pred, target,
                                                                                           loss);
  return check_launch("reproj_fwd_kernel");
}

extern "C" int pd_reproj_loss_bwd(int B, int H, int W, int use_ssim, const float* pred, const float* target,
                                  const float* g_loss, float* g_pred, float* g_target, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && H >= 2 && W >= 2, "bad shape");
  PD_REQUIRE(pred && target && g_loss && g_pred, "NULL pointer");
  PD_REQUIRE((long)B * 3 <= 65535, "batch too large for one launch");
  dim3 grid(ceil_div(W, kTileW), ceil_div(H, kTileH), B * 3);
  if (g_target)
    reproj_bwd_kernel<true><<<grid, kBlock, 0, (hipStream_t)stream>>>(H, W, use_ssim, pred, target, g_loss, g_pred,
                                                                       g_target);
  else
    reproj_bwd_kernel<false><<<grid, kBlock, 0, (hipStream_t)stream>>>(H, W, use_ssim, pred, target, g_loss, g_pred,
                                                                        g_target);
  return check_launch("reproj_bwd_kernel");
}
