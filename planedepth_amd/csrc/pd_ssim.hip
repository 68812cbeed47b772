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

__global__ __launch_bounds__(kBlock) void ssim_fwd_kernel(int H, int W, const float* __restrict__ x,
                                                          const float* __restrict__ y, float* __restrict__ out) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= H * W) return;
  const long img = (long)blockIdx.y * H * W;
  const int py = pix / W, px = pix - py * W;
  out[img + pix] = ssim_value(window_stats(x + img, y + img, px, py, H, W));
}

// g_x(q) = sum over p, over offsets (dy,dx) with reflect(p+offset) == q, of  ax(p) + bx(p)*x(q) + c(p)*y(q)
template <bool WANT_Y>
__global__ __launch_bounds__(kBlock) void ssim_bwd_kernel(int H, int W, const float* __restrict__ x,
                                                          const float* __restrict__ y, const float* __restrict__ g_out,
                                                          float* __restrict__ g_x, float* __restrict__ g_y) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= H * W) return;
  const long img = (long)blockIdx.y * H * W;
  const int qy = pix / W, qx = pix - qy * W;
  const float xq = x[img + pix], yq = y[img + pix];
  float gx = 0.0f, gy = 0.0f;
  // candidate centres p within distance 2 (reflection can map p+offset from outside back onto q)
  for (int py = max(qy - 2, 0); py <= min(qy + 2, H - 1); ++py) {
    int cy = 0;
    for (int dy = -1; dy <= 1; ++dy) cy += (reflect(py + dy, H) == qy);
    if (!cy) continue;
    for (int px = max(qx - 2, 0); px <= min(qx + 2, W - 1); ++px) {
      int cx = 0;
      for (int dx = -1; dx <= 1; ++dx) cx += (reflect(px + dx, W) == qx);
      if (!cx) continue;
      const float g = g_out[img + py * W + px];
      const SsimCoef k = ssim_coef(window_stats(x + img, y + img, px, py, H, W), g);
      const float mult = (float)(cx * cy);
      gx += mult * (k.ax + k.bx * xq + k.c * yq);
      if (WANT_Y) gy += mult * (k.ay + k.by * yq + k.c * xq);
    }
  }
  if (g_x) g_x[img + pix] = gx;
  if (WANT_Y && g_y) g_y[img + pix] = gy;
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

template <bool WANT_T>
__global__ __launch_bounds__(kBlock) void reproj_bwd_kernel(int H, int W, int use_ssim, const float* __restrict__ pred,
                                                            const float* __restrict__ tgt,
                                                            const float* __restrict__ g_loss,
                                                            float* __restrict__ g_pred, float* __restrict__ g_tgt) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  const int HW = H * W;
  if (pix >= HW) return;
  const int b = blockIdx.y, c = blockIdx.z;
  const long img = ((long)b * 3 + c) * HW;
  const float* gl = g_loss + (long)b * HW;
  const int qy = pix / W, qx = pix - qy * W;
  const float xq = pred[img + pix], yq = tgt[img + pix];
  const float wl1 = (use_ssim ? 0.15f : 1.0f) / 3.0f, wss = 0.85f / 3.0f;
  // L1 part: d|t - p|/dp = -sgn(t - p)
  float gx = -wl1 * gl[pix] * sgn(yq - xq), gy = -gx;
  if (use_ssim) {
    for (int py = max(qy - 2, 0); py <= min(qy + 2, H - 1); ++py) {
      int cy = 0;
      for (int dy = -1; dy <= 1; ++dy) cy += (reflect(py + dy, H) == qy);
      if (!cy) continue;
      for (int px = max(qx - 2, 0); px <= min(qx + 2, W - 1); ++px) {
        int cx = 0;
        for (int dx = -1; dx <= 1; ++dx) cx += (reflect(px + dx, W) == qx);
        if (!cx) continue;
        const SsimCoef k = ssim_coef(window_stats(pred + img, tgt + img, px, py, H, W), wss * gl[py * W + px]);
        const float mult = (float)(cx * cy);
        gx += mult * (k.ax + k.bx * xq + k.c * yq);
        if (WANT_T) gy += mult * (k.ay + k.by * yq + k.c * xq);
      }
    }
  }
  g_pred[img + pix] = gx;
  if (WANT_T) g_tgt[img + pix] = gy;
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
  dim3 grid(ceil_div(H * W, kBlock), B * C);
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
  dim3 grid(ceil_div(H * W, kBlock), B, 3);
  if (g_target)
    reproj_bwd_kernel<true><<<grid, kBlock, 0, (hipStream_t)stream>>>(H, W, use_ssim, pred, target, g_loss, g_pred,
                                                                       g_target);
  else
    reproj_bwd_kernel<false><<<grid, kBlock, 0, (hipStream_t)stream>>>(H, W, use_ssim, pred, target, g_loss, g_pred,
                                                                        g_target);
  return check_launch("reproj_bwd_kernel");
}
