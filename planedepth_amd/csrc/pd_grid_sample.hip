// F.grid_sample(mode="bilinear", padding_mode=zeros|border, align_corners=True) forward / backward, as the reference
// calls it outside the fused sweep: self-reconstruction (trainer.py:624-628, border) and the self-distillation
// post-process (trainer.py:444-463, zeros).  SURVEY.md row A5 gives the formula; torch's kernel is third-party.
#include "pd_common.h"

namespace pd {

// border mode clamps the COORDINATE to [0, size-1]; gradient w.r.t. the coordinate is zero where it was clamped
__device__ __forceinline__ float clip_coord(float v, float hi, float& dmul) {
  if (v <= 0.0f) { dmul = 0.0f; return 0.0f; }   // torch's clip_coordinates_set_grad: grad 0 at/below 0 ...
  if (v >= hi) { dmul = 0.0f; return hi; }       // ... and at/above size-1
  dmul = 1.0f;
  return v;
}

__global__ __launch_bounds__(kBlock) void grid_sample_fwd_kernel(int C, int Hi, int Wi, int HWo, int border,
                                                                 const float* __restrict__ in,
                                                                 const float* __restrict__ grid,
                                                                 float* __restrict__ out) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= HWo) return;
  const int m = blockIdx.y;
  const float2 g = reinterpret_cast<const float2*>(grid)[(long)m * HWo + pix];
  float ix = unnormalise(g.x, (float)(Wi - 1)), iy = unnormalise(g.y, (float)(Hi - 1));
  if (border) {
    float dm;
    ix = clip_coord(ix, (float)(Wi - 1), dm);
    iy = clip_coord(iy, (float)(Hi - 1), dm);
  }
  const Tap t = make_tap(ix, iy, Wi, Hi);
  for (int c = 0; c < C; ++c)
    out[((long)m * C + c) * HWo + pix] = bilinear(in + ((long)m * C + c) * Hi * Wi, t, Wi);
}

__global__ __launch_bounds__(kBlock) void grid_sample_bwd_kernel(int C, int Hi, int Wi, int HWo, int border,
                                                                 const float* __restrict__ in,
                                                                 const float* __restrict__ grid,
                                                                 const float* __restrict__ g_out,
                                                                 float* __restrict__ g_in, float* __restrict__ g_grid) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  const bool live = pix < HWo;   // no early return: the wave-cooperative scatter needs all 64 lanes
  const int m = blockIdx.y;
  const float2 g = live ? reinterpret_cast<const float2*>(grid)[(long)m * HWo + pix] : make_float2(0.0f, 0.0f);
  float mx = (float)(Wi - 1) / 2, my = (float)(Hi - 1) / 2;
  float ix = unnormalise(g.x, (float)(Wi - 1)), iy = unnormalise(g.y, (float)(Hi - 1));
  if (border) {
    float dmx, dmy;
    ix = clip_coord(ix, (float)(Wi - 1), dmx);
    iy = clip_coord(iy, (float)(Hi - 1), dmy);
    mx *= dmx;
    my *= dmy;
  }
  const Tap t = make_tap(ix, iy, Wi, Hi);
  // neighbouring output pixels usually sample neighbouring input pixels: lanes whose left column is the previous lane's
  // right column hand their left taps over (pd_common.h) — two atomics per lane and channel instead of four
  const ScatterPlan sp = plan_scatter(t, Wi, live);
  float gix = 0.0f, giy = 0.0f;
  for (int c = 0; c < C; ++c) {
    const long plane = ((long)m * C + c) * Hi * Wi;
    const float go = live ? g_out[((long)m * C + c) * HWo + pix] : 0.0f;
#ifndef PD_GS_NOSCATTER  // diagnostics: the kernel without its atomics
    if (g_in) bilinear_scatter_wave(g_in + plane, t, Wi, go, live, sp);
#endif
    if (g_grid && live) {
      float dx, dy;
      bilinear_grad(in + plane, t, Wi, dx, dy);
      gix += go * dx;
      giy += go * dy;
    }
  }
  if (g_grid && live) reinterpret_cast<float2*>(g_grid)[(long)m * HWo + pix] = make_float2(mx * gix, my * giy);
}

}  // namespace pd

using namespace pd;

extern "C" int pd_grid_sample_fwd(int M, int C, int Hi, int Wi, int Ho, int Wo, int padding_mode, const float* input,
                                  const float* grid, float* out, pd_stream_t stream) {
  PD_REQUIRE(M > 0 && M <= 65535 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bad shape");
  PD_REQUIRE(padding_mode == PD_PAD_ZEROS || padding_mode == PD_PAD_BORDER, "padding_mode must be zeros or border");
  PD_REQUIRE(input && grid && out, "NULL pointer");
  grid_sample_fwd_kernel<<<dim3(ceil_div(Ho * Wo, kBlock), M), kBlock, 0, (hipStream_t)stream>>>(
      C, Hi, Wi, Ho * Wo, padding_mode == PD_PAD_BORDER, input, grid, out);
  return check_launch("grid_sample_fwd_kernel");
}

extern "C" int pd_grid_sample_bwd(int M, int C, int Hi, int Wi, int Ho, int Wo, int padding_mode, const float* input,
                                  const float* grid, const float* g_out, float* g_input, float* g_grid,
                                  pd_stream_t stream) {
  PD_REQUIRE(M > 0 && M <= 65535 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bad shape");
  PD_REQUIRE(padding_mode == PD_PAD_ZEROS || padding_mode == PD_PAD_BORDER, "padding_mode must be zeros or border");
  PD_REQUIRE(input && grid && g_out && (g_input || g_grid), "NULL pointer");
  grid_sample_bwd_kernel<<<dim3(ceil_div(Ho * Wo, kBlock), M), kBlock, 0, (hipStream_t)stream>>>(
      C, Hi, Wi, Ho * Wo, padding_mode == PD_PAD_BORDER, input, grid, g_out, g_input, g_grid);
  return check_launch("grid_sample_bwd_kernel");
}
