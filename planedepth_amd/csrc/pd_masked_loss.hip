// Photometric loss under the occlusion mask `mask_novel` (reference trainer.py:724-742).
//
// The mask comes from Trainer.mirror_occlusion_mask / generate_post_process_disp, which process_batch runs AFTER
// pred_novel_images (trainer.py:342-349), so the sweep's forward cannot take it; what it changes is [B,3,H,W]-sized:
//   pred   = rgb_rec * mask + target * (1 - mask)                                   (:726, also fed to the perceptual net)
//   L1:      ph = mean_c |pred - target|, automask: min(ph, mean_c |source - target|) (:738-741), loss = ph.mean() (:742)
//   mixture: loss = (ph_map * mask).mean()   with ph_map the sweep's per-pixel NLL   (:736, :742)
// One kernel forward (blend + per-pixel loss + workgroup partial sums, finished by a one-wave reduce: deterministic),
// one backward — instead of 8-12 elementwise torch operators and their autograd nodes.
#include "pd_common.h"

namespace pd {

struct MaskedPix {
  float p[3];   // blended prediction
  float v;      // this pixel's loss value
  bool sel;     // L1: the reprojection loss (not the automask's identity loss) is the minimum; ties go to it (torch.min
                // over cat([ph, auto]) returns the first index)
};

__device__ __forceinline__ MaskedPix masked_l1_pixel(const float* __restrict__ rgb, const float* __restrict__ tgt,
                                                     const float* __restrict__ src, float m, long i, int HW) {
#pragma clang fp contract(off)
  MaskedPix r;
  float e = 0.0f, a = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = tgt[i + (long)c * HW];
    r.p[c] = rgb[i + (long)c * HW] * m + t * (1.0f - m);
    e += fabsf(r.p[c] - t);
    if (src) a += fabsf(src[i + (long)c * HW] - t);
  }
  e = e / 3.0f;
  a = a / 3.0f;
  r.sel = !src || e <= a;
  r.v = r.sel ? e : a;
  return r;
}

template <bool MIX>
__global__ __launch_bounds__(kBlock) void masked_photometric_fwd_kernel(int HW, const float* __restrict__ rgb,
                                                                        const float* __restrict__ tgt,
                                                                        const float* __restrict__ src,
                                                                        const float* __restrict__ mask,
                                                                        const float* __restrict__ ph_map,
                                                                        float* __restrict__ pred, float* __restrict__ partials) {
  __shared__ float red[kBlock / kWave];
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  float v = 0.0f;
  if (pix < HW) {
    const float m = mask ? mask[(long)b * HW + pix] : 1.0f;
    const long i = (long)b * 3 * HW + pix;
    const MaskedPix r = masked_l1_pixel(rgb, tgt, MIX ? nullptr : src, m, i, HW);
    if (pred) {
#pragma unroll
      for (int c = 0; c < 3; ++c) pred[i + (long)c * HW] = r.p[c];
    }
    v = MIX ? ph_map[(long)b * HW + pix] * m : r.v;
  }
  v = wave_sum(v);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s += red[w];
    partials[(long)b * gridDim.x + blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kWave) void masked_mean_kernel(const float* __restrict__ partials, int n, float inv_count,
                                                            float* __restrict__ mean) {
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += kWave) s += partials[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) mean[0] = s * inv_count;
}

template <bool MIX>
__global__ __launch_bounds__(kBlock) void masked_photometric_bwd_kernel(int HW, float inv_count, const float* __restrict__ rgb,
                                                                        const float* __restrict__ tgt,
                                                                        const float* __restrict__ src,
                                                                        const float* __restrict__ mask,
                                                                        const float* __restrict__ g_mean,
                                                                        const float* __restrict__ g_pred,
                                                                        float* __restrict__ g_rgb, float* __restrict__ g_ph_map) {
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= HW) return;
  const float g = g_mean ? g_mean[0] * inv_count : 0.0f;
  const float m = mask ? mask[(long)b * HW + pix] : 1.0f;
  const long i = (long)b * 3 * HW + pix;
  if (MIX) {
    if (g_ph_map) g_ph_map[(long)b * HW + pix] = g * m;
    if (g_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) g_rgb[i + (long)c * HW] = g_pred ? g_pred[i + (long)c * HW] * m : 0.0f;
    }
    return;
  }
  if (!g_rgb) return;
  const MaskedPix r = masked_l1_pixel(rgb, tgt, src, m, i, HW);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float gp = g_pred ? g_pred[i + (long)c * HW] : 0.0f;
    if (r.sel) gp += g * sgn(r.p[c] - tgt[i + (long)c * HW]) / 3.0f;
    g_rgb[i + (long)c * HW] = gp * m;
  }
}

}  // namespace pd

using namespace pd;

extern "C" int pd_masked_photometric_fwd(int B, int H, int W, int mixture, const float* rgb_rec, const float* target,
                                         const float* source, const float* mask, const float* ph_map, float* pred,
                                         float* partials, float* mean, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && (long)H * W < (1L << 31), "bad shape");
  PD_REQUIRE(rgb_rec && target && partials && mean, "NULL pointer");
  PD_REQUIRE(!mixture || ph_map, "mixture mode needs the sweep's ph_map");
  const int HW = H * W, nblk = ceil_div(HW, kBlock);
  dim3 grid(nblk, B);
  hipStream_t s = (hipStream_t)stream;
  if (mixture) masked_photometric_fwd_kernel<true><<<grid, kBlock, 0, s>>>(HW, rgb_rec, target, source, mask, ph_map, pred, partials);
  else         masked_photometric_fwd_kernel<false><<<grid, kBlock, 0, s>>>(HW, rgb_rec, target, source, mask, ph_map, pred, partials);
  masked_mean_kernel<<<1, kWave, 0, s>>>(partials, nblk * B, 1.0f / ((float)B * (float)HW), mean);
  return check_launch("masked_photometric_fwd_kernel");
}

extern "C" int pd_masked_photometric_bwd(int B, int H, int W, int mixture, const float* rgb_rec, const float* target,
                                         const float* source, const float* mask, const float* g_mean, const float* g_pred,
                                         float* g_rgb_rec, float* g_ph_map, pd_stream_t stream) {
  PD_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && (long)H * W < (1L << 31), "bad shape");
  PD_REQUIRE(rgb_rec && target && (g_mean || g_pred) && (g_rgb_rec || g_ph_map), "NULL pointer");
  const int HW = H * W;
  dim3 grid(ceil_div(HW, kBlock), B);
  const float inv = 1.0f / ((float)B * (float)HW);
  hipStream_t s = (hipStream_t)stream;
  if (mixture) masked_photometric_bwd_kernel<true><<<grid, kBlock, 0, s>>>(HW, inv, rgb_rec, target, source, mask, g_mean, g_pred, g_rgb_rec, g_ph_map);
  else         masked_photometric_bwd_kernel<false><<<grid, kBlock, 0, s>>>(HW, inv, rgb_rec, target, source, mask, g_mean, g_pred, g_rgb_rec, g_ph_map);
  return check_launch("masked_photometric_bwd_kernel");
}
