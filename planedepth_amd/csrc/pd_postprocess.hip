// Warps of the self-distillation post-process, Trainer.generate_post_process_disp (reference trainer.py:421-466;
// SURVEY.md §8f rank 2).  The reference builds two [B*N,H,W,2] grids, calls F.grid_sample five times on [B*N,1,H,W]
// tensors, two softmaxes over the planes and three sum/clamp passes.  Both shapes it needs are fused here:
//   pd_warp_softmax   out[b,n,y,x] = softmax_n( planes[b,n] sampled at (x + s*d[b,n], y) )       (trainer.py:443-446, 451-453)
//   pd_warp_sum       out[b,0,y,x] = min(cap, sum_n planes[b,n] sampled at (x + s*d[b,n], y) )   (:447-449, 454-456, 463-465)
// with the reference's pixel -> [-1,1] -> pixel round trip (trainer.py:427-441 + grid_sample, align_corners=True,
// zeros padding) reproduced through pd_common.h's normalise_roundtrip, and PD_PP_FLIP_SRC reading the source planes
// mirrored along x (the `.flip(-1)` of trainer.py:451) without materialising the flipped tensor.  Forward only: the
// reference runs this under the fixed (no-grad) networks and detaches the result (:466).
#include "pd_common.h"

namespace pd {

struct WarpArgs {
  int N, H, W;
  int dense, flip;
  float sign;
  const float* planes;
  const float* disp;
};

// bilinear sample of one [H,W] plane with optionally mirrored columns
template <bool FLIP>
__device__ __forceinline__ float sample(const float* __restrict__ p, const Tap& t, int W) {
  const int c0 = FLIP ? (W - 1 - t.x0) : t.x0, c1 = FLIP ? (W - 2 - t.x0) : (t.x0 + 1);
  const float* r0 = p + (long)t.y0 * W;
  const float* r1 = r0 + W;
  const float nw = (t.vx0 && t.vy0) ? r0[c0] : 0.0f;
  const float ne = (t.vx1 && t.vy0) ? r0[c1] : 0.0f;
  const float sw = (t.vx0 && t.vy1) ? r1[c0] : 0.0f;
  const float se = (t.vx1 && t.vy1) ? r1[c1] : 0.0f;
  return nw * (t.wx0 * t.wy0) + ne * (t.wx1 * t.wy0) + sw * (t.wx0 * t.wy1) + se * (t.wx1 * t.wy1);
}

__device__ __forceinline__ Tap plane_tap(const WarpArgs& a, int b, int n, int x, int y, float iy) {
  const float d = a.dense ? a.disp[(((long)b * a.N + n) * a.H + y) * a.W + x] : a.disp[b * a.N + n];
  const float ix = normalise_roundtrip((float)x + a.sign * d, (float)(a.W - 1));
  return make_tap(ix, iy, a.W, a.H);
}

template <bool FLIP>
__global__ __launch_bounds__(kBlock) void warp_softmax_kernel(WarpArgs a, float* __restrict__ out) {
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= HW) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const float iy = normalise_roundtrip((float)y, (float)(a.H - 1));
  const float* pb = a.planes + (long)b * a.N * HW;
  float* ob = out + (long)b * a.N * HW + pix;
  // pass 1: sampled logits to `out`, running max / sum; pass 2: normalise this pixel's own N values in place
  float m = -INFINITY, Z = 0.0f;
  for (int n = 0; n < a.N; ++n) {
    const float l = sample<FLIP>(pb + (long)n * HW, plane_tap(a, b, n, x, y, iy), a.W);
    ob[(long)n * HW] = l;
    if (l > m) { Z *= __expf(m - l); m = l; }
    Z += __expf(l - m);
  }
  const float invZ = 1.0f / Z;
  for (int n = 0; n < a.N; ++n) ob[(long)n * HW] = __expf(ob[(long)n * HW] - m) * invZ;
}

template <bool FLIP>
__global__ __launch_bounds__(kBlock) void warp_sum_kernel(WarpArgs a, float cap, float* __restrict__ out) {
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= HW) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const float iy = normalise_roundtrip((float)y, (float)(a.H - 1));
  const float* pb = a.planes + (long)b * a.N * HW;
  float acc = 0.0f;
  for (int n = 0; n < a.N; ++n) acc += sample<FLIP>(pb + (long)n * HW, plane_tap(a, b, n, x, y, iy), a.W);
  out[(long)b * HW + pix] = fminf(acc, cap);   // o[o > 1] = 1
}

static int warp_args(WarpArgs& a, int B, int N, int H, int W, float sign, int flags, const float* planes,
                     const float* disp) {
  PD_REQUIRE(B > 0 && B <= 65535 && N > 0 && H > 0 && W > 1, "bad shape");
  PD_REQUIRE((long)H * W < (1L << 31), "image too large");
  PD_REQUIRE((flags & ~(PD_PP_DISP_DENSE | PD_PP_FLIP_SRC)) == 0, "unknown flags");
  PD_REQUIRE(planes && disp, "NULL pointer");
  a.N = N; a.H = H; a.W = W;
  a.dense = (flags & PD_PP_DISP_DENSE) != 0;
  a.flip = (flags & PD_PP_FLIP_SRC) != 0;
  a.sign = sign; a.planes = planes; a.disp = disp;
  return 0;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_warp_softmax(int B, int N, int H, int W, float sign, int flags, const float* planes,
                               const float* disp, float* out, pd_stream_t stream) {
  WarpArgs a;
  if (int rc = warp_args(a, B, N, H, W, sign, flags, planes, disp)) return rc;
  PD_REQUIRE(out && out != planes, "out must be a distinct buffer");
  dim3 grid(ceil_div(H * W, kBlock), B);
  if (a.flip) warp_softmax_kernel<true><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, out);
  else        warp_softmax_kernel<false><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, out);
  return check_launch("warp_softmax_kernel");
}

extern "C" int pd_warp_sum(int B, int N, int H, int W, float sign, int flags, const float* planes, const float* disp,
                           float cap, float* out, pd_stream_t stream) {
  WarpArgs a;
  if (int rc = warp_args(a, B, N, H, W, sign, flags, planes, disp)) return rc;
  PD_REQUIRE(out, "NULL output");
  dim3 grid(ceil_div(H * W, kBlock), B);
  if (a.flip) warp_sum_kernel<true><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, cap, out);
  else        warp_sum_kernel<false><<<grid, kBlock, 0, (hipStream_t)stream>>>(a, cap, out);
  return check_launch("warp_sum_kernel");
}
