// Batch doubling of Trainer.add_flip_right_inputs (reference trainer.py:252-276): every image-like
// input x of the stereo pair becomes cat([x_own, flip(x_other, -1)], dim 0) — the mirrored right image is a valid left
// image.  The reference issues a flip (copy) and a cat (second copy) per tensor; here one kernel writes the doubled
// batch directly.  `negate_c0` covers the grid tensor, whose x-coordinate channel changes sign under the mirror
// (trainer.py:258-260).  Pure data movement: bit-exact.
#include <math.h>
#include "pd_common.h"

namespace pd {

__global__ __launch_bounds__(kBlock) void cat_flip_kernel(int C, int H, int W, const float* __restrict__ own,
                                                          const float* __restrict__ other, int negate_c0,
                                                          float* __restrict__ out, int B) {
  const long n = (long)C * H * W;                 // elements per image
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  const int b = blockIdx.y;                       // 0 .. 2B-1
  if (i >= n) return;
  float v;
  if (b < B) {
    v = own[(long)b * n + i];
  } else {
    const int x = (int)(i % W);
    const long row = i - x;
    v = other[(long)(b - B) * n + row + (W - 1 - x)];
    if (negate_c0 && i < (long)H * W) v = -v;     // channel 0 of the mirrored half
  }
  out[(long)b * n + i] = v;
}

// inputs["grid"] of the reference's data pipeline, generated on the device:
// RandomResizeCrop (datasets/pair_transforms.py:27-37) builds torch.linspace(-1, 1, full_w) x torch.linspace(-1, 1, full_h)
// over the RESIZED frame and crops the H x W window at (h0, w0); Resize (:63-68) is full = (H, W), h0 = w0 = 0.
// torch.linspace's scalar formula (ATen RangeFactories: step = (end - start) / (steps - 1) in the tensor's dtype;
// start + step * i below the midpoint, end - step * (steps - 1 - i) from it on).  ATen's CPU kernel evaluates it per
// vector of 8 / 16 lanes (base + j * step), so the reference's own grid differs in the last bit between host CPUs;
// this kernel is within one ulp of either (tests/test_gpu_parity.py).
__device__ __forceinline__ float linspace_m1_1(int i, int steps) {
#pragma clang fp contract(off)
  if (steps <= 1) return -1.0f;
  const float step = (1.0f - (-1.0f)) / (float)(steps - 1);
  const int half = steps / 2;
  return (i < half) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(steps - i - 1));
}

__global__ __launch_bounds__(kBlock) void crop_grid_kernel(int H, int W, const int* __restrict__ params,
                                                           float* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const int full_w = params[b * 4 + 0], full_h = params[b * 4 + 1], w0 = params[b * 4 + 2], h0 = params[b * 4 + 3];
  float* o = out + (long)b * 2 * H * W;
  o[i] = linspace_m1_1(w0 + x, full_w);
  o[(long)H * W + i] = linspace_m1_1(h0 + y, full_h);
}

// The decoders' disparity levels (networks/depth_decoder.py:147-152, networks/plade_net.py:280-285):
//   disp_layered[b,n] = disp_max * (disp_min / disp_max) ** (levels[b,n] / (no_levels - 1)),  distance = 0.1 * 0.58 * W / disp_layered
// with levels = arange(no_levels) + the learnt residual.  The reference spends ~5 elementwise launches forward and ~8 backward
// on these [B,N,1,1] tensors every step; one launch each way here (torch's operation order: the base rounded to fp32 once, the
// exponent's division, powf, the product).
__global__ void plane_levels_fwd_kernel(int M, float base, float disp_max, float inv_nm1_den, float dist_num,
                                        const float* __restrict__ levels, float* __restrict__ disp, float* __restrict__ distance) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= M) return;
  const float e = levels[i] / inv_nm1_den;
  const float d = disp_max * powf(base, e);
  disp[i] = d;
  if (distance) distance[i] = dist_num / d;
}
// d disp / d level = disp ln(base) / (no_levels - 1);  d distance / d disp = -distance / disp
__global__ void plane_levels_bwd_kernel(int M, float ln_base, float inv_nm1_den, float dist_num, const float* __restrict__ disp,
                                        const float* __restrict__ g_disp, const float* __restrict__ g_distance,
                                        float* __restrict__ g_levels) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= M) return;
  const float d = disp[i];
  float g = g_disp ? g_disp[i] : 0.0f;
  if (g_distance) g -= g_distance[i] * dist_num / (d * d);
  g_levels[i] = g * d * ln_base / inv_nm1_den;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_plane_levels_fwd(int M, int no_levels, float disp_min, float disp_max, float dist_num, const float* levels,
                                   float* disp, float* distance, pd_stream_t stream) {
  PD_REQUIRE(M > 0 && no_levels > 1 && disp_min > 0.0f && disp_max > 0.0f, "bad arguments");
  PD_REQUIRE(levels && disp, "NULL pointer");
  const float base = (float)((double)disp_min / (double)disp_max);   // the reference's Python-float quotient, rounded once
  plane_levels_fwd_kernel<<<ceil_div(M, kBlock), kBlock, 0, (hipStream_t)stream>>>(M, base, disp_max, (float)(no_levels - 1), dist_num,
                                                                                  levels, disp, distance);
  return check_launch("plane_levels_fwd_kernel");
}

extern "C" int pd_plane_levels_bwd(int M, int no_levels, float disp_min, float disp_max, float dist_num, const float* disp,
                                   const float* g_disp, const float* g_distance, float* g_levels, pd_stream_t stream) {
  PD_REQUIRE(M > 0 && no_levels > 1 && disp_min > 0.0f && disp_max > 0.0f, "bad arguments");
  PD_REQUIRE(disp && g_levels && (g_disp || g_distance), "NULL pointer");
  const float base = (float)((double)disp_min / (double)disp_max);
  plane_levels_bwd_kernel<<<ceil_div(M, kBlock), kBlock, 0, (hipStream_t)stream>>>(M, logf(base), (float)(no_levels - 1), dist_num, disp,
                                                                                  g_disp, g_distance, g_levels);
  return check_launch("plane_levels_bwd_kernel");
}

extern "C" int pd_crop_grid(int B, int H, int W, const int* params, float* grid, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && (long)H * W < (1L << 31), "bad shape");
  PD_REQUIRE(params && grid, "NULL pointer");
  crop_grid_kernel<<<dim3(ceil_div(H * W, kBlock), B), kBlock, 0, (hipStream_t)stream>>>(H, W, params, grid);
  return check_launch("crop_grid_kernel");
}

extern "C" int pd_cat_flip(int B, int C, int H, int W, const float* own, const float* other, int negate_c0, float* out,
                           pd_stream_t stream) {
  PD_REQUIRE(B > 0 && 2 * B <= 65535 && C > 0 && H > 0 && W > 0, "bad shape");
  PD_REQUIRE(own && other && out && out != own && out != other, "NULL or aliasing pointer");
  const long n = (long)C * H * W;
  PD_REQUIRE(n < (1L << 31), "image too large");
  cat_flip_kernel<<<dim3((unsigned)((n + kBlock - 1) / kBlock), 2 * B), kBlock, 0, (hipStream_t)stream>>>(
      C, H, W, own, other, negate_c0, out, B);
  return check_launch("cat_flip_kernel");
}
