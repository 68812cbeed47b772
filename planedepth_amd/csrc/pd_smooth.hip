// Edge-aware smoothness of a disparity image: get_smooth_loss_disp, reference layers.py:243-256 (called at
// trainer.py:768 on the crops disp[..., 0.2W:], color[..., 0.2W:]).
//   loss = mean_{x<W-1} |d(x) - d(x+1)| exp(-gamma mean_c |I(x) - I(x+1)|)  +  the same along y
// The reference runs ~14 ATen passes (slices, abs, mean over channels, exp, two means).  Here: one kernel reads every
// pixel once and adds block sums into a single float; the backward is one kernel too.  The operands may be crops of
// wider tensors (unit column stride, explicit row / channel / batch strides), so the 0.2W crop costs no copy.
#include "pd_common.h"

namespace pd {

struct SmoothArgs {
  int C, H, W;
  long d_sb, d_sh;          // disp strides (floats): batch, row
  long i_sb, i_sc, i_sh;    // image strides: batch, channel, row
  float gamma, inv_nx, inv_ny;
  const float* disp;
  const float* img;
};

// exp(-gamma * mean_c |I(p) - I(q)|) for the pixel pair at offsets p, q of image b
__device__ __forceinline__ float edge_weight(const SmoothArgs& a, const float* __restrict__ ib, long p, long q) {
  float s = 0.0f;
  for (int c = 0; c < a.C; ++c) s += fabsf(ib[c * a.i_sc + p] - ib[c * a.i_sc + q]);
  return __expf(-a.gamma * (s / (float)a.C));
}

#ifndef PD_SMOOTH_ROWS
#define PD_SMOOTH_ROWS 4
#endif
constexpr int kSmoothRows = PD_SMOOTH_ROWS;  // image rows per block of the forward; one atomic on the single output each
// (measured at 8x192x512: 8 rows per block with a flattened (row, column) index and its integer division per pixel
// 31 us; row loops with 1 / 2 / 4 / 8 / 16 rows per block 24.5 / 18.4 / 18.3 / 29.9 / 55 us — fewer rows means more
// atomics on the one output word, more rows fewer workgroups than CUs)

__global__ __launch_bounds__(kBlock) void smooth_fwd_kernel(SmoothArgs a, float* __restrict__ out) {
  __shared__ float red[kBlock / kWave];
  const int b = blockIdx.y, y0 = blockIdx.x * kSmoothRows, y1 = min(y0 + kSmoothRows, a.H);
  const float* db = a.disp + b * a.d_sb;
  const float* ib = a.img + b * a.i_sb;
  float v = 0.0f;
  for (int y = y0; y < y1; ++y) {
    const float* dr = db + y * a.d_sh;
    const long ir = y * a.i_sh;
    const bool down = y + 1 < a.H;
    for (int x = threadIdx.x; x < a.W; x += kBlock) {
      const float d = dr[x];
      if (x + 1 < a.W) v += fabsf(d - dr[x + 1]) * edge_weight(a, ib, ir + x, ir + x + 1) * a.inv_nx;
      if (down) v += fabsf(d - dr[a.d_sh + x]) * edge_weight(a, ib, ir + x, ir + x + a.i_sh) * a.inv_ny;
    }
  }
  v = wave_sum(v);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int i = 0; i < kBlock / kWave; ++i) s += red[i];
    unsafeAtomicAdd(out, s);
  }
}

// d loss / d disp at every pixel: its right/down pair and the left/up pair it is the second member of
__global__ __launch_bounds__(kBlock) void smooth_bwd_kernel(SmoothArgs a, const float* __restrict__ g_out,
                                                            float* __restrict__ g_disp) {
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= a.H * a.W) return;
  const int y = pix / a.W, x = pix - y * a.W;
  const float* db = a.disp + b * a.d_sb;
  const float* ib = a.img + b * a.i_sb;
  const float d = db[y * a.d_sh + x];
  const long ip = y * a.i_sh + x;
  float g = 0.0f;
  if (x + 1 < a.W) g += sgn(d - db[y * a.d_sh + x + 1]) * edge_weight(a, ib, ip, ip + 1) * a.inv_nx;
  if (x > 0)       g -= sgn(db[y * a.d_sh + x - 1] - d) * edge_weight(a, ib, ip - 1, ip) * a.inv_nx;
  if (y + 1 < a.H) g += sgn(d - db[(y + 1) * a.d_sh + x]) * edge_weight(a, ib, ip, ip + a.i_sh) * a.inv_ny;
  if (y > 0)       g -= sgn(db[(y - 1) * a.d_sh + x] - d) * edge_weight(a, ib, ip - a.i_sh, ip) * a.inv_ny;
  g_disp[(long)b * a.H * a.W + pix] = g * g_out[0];
}

// The same with the gradient written into the UNCROPPED tensor: rows of W + x_pad floats whose first x_pad columns (the
// part trainer.py:768 crops away) get their exact zero here — autograd then has no slice to undo (a zero-fill + a
// strided copy + three host-side operator calls per step).
__global__ __launch_bounds__(kBlock) void smooth_bwd_padded_kernel(SmoothArgs a, const float* __restrict__ g_out,
                                                                   float* __restrict__ g_disp, int x_pad) {
  const int Wf = a.W + x_pad;
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= a.H * Wf) return;
  const int y = pix / Wf, xf = pix - y * Wf, x = xf - x_pad;
  float g = 0.0f;
  if (x >= 0) {
    const float* db = a.disp + b * a.d_sb;
    const float* ib = a.img + b * a.i_sb;
    const float d = db[y * a.d_sh + x];
    const long ip = y * a.i_sh + x;
    if (x + 1 < a.W) g += sgn(d - db[y * a.d_sh + x + 1]) * edge_weight(a, ib, ip, ip + 1) * a.inv_nx;
    if (x > 0)       g -= sgn(db[y * a.d_sh + x - 1] - d) * edge_weight(a, ib, ip - 1, ip) * a.inv_nx;
    if (y + 1 < a.H) g += sgn(d - db[(y + 1) * a.d_sh + x]) * edge_weight(a, ib, ip, ip + a.i_sh) * a.inv_ny;
    if (y > 0)       g -= sgn(db[(y - 1) * a.d_sh + x] - d) * edge_weight(a, ib, ip - a.i_sh, ip) * a.inv_ny;
    g *= g_out[0];
  }
  g_disp[(long)b * a.H * Wf + pix] = g;
}

static int smooth_args(SmoothArgs& a, int B, int C, int H, int W, const float* disp, long d_sb, long d_sh,
                       const float* img, long i_sb, long i_sc, long i_sh, float gamma) {
  PD_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 1 && W > 1, "bad shape (needs H, W >= 2)");
  PD_REQUIRE((long)H * W < (1L << 31), "image too large");
  PD_REQUIRE(disp && img, "NULL pointer");
  a.C = C; a.H = H; a.W = W;
  a.d_sb = d_sb; a.d_sh = d_sh; a.i_sb = i_sb; a.i_sc = i_sc; a.i_sh = i_sh;
  a.gamma = gamma;
  a.inv_nx = 1.0f / ((float)B * (float)H * (float)(W - 1));   // .mean() over [B,1,H,W-1]
  a.inv_ny = 1.0f / ((float)B * (float)(H - 1) * (float)W);   // .mean() over [B,1,H-1,W]
  a.disp = disp; a.img = img;
  return 0;
}

}  // namespace pd

using namespace pd;

extern "C" int pd_smooth_loss_fwd(int B, int C, int H, int W, const float* disp, int64_t disp_stride_b, int64_t disp_stride_h,
                                  const float* img, int64_t img_stride_b, int64_t img_stride_c, int64_t img_stride_h, float gamma,
                                  float* out, pd_stream_t stream) {
  SmoothArgs a;
  if (int rc = smooth_args(a, B, C, H, W, disp, disp_stride_b, disp_stride_h, img, img_stride_b, img_stride_c,
                           img_stride_h, gamma)) return rc;
  PD_REQUIRE(out, "NULL output");
  if (hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return check_launch("hipMemsetAsync");
  smooth_fwd_kernel<<<dim3(ceil_div(H, kSmoothRows), B), kBlock, 0, (hipStream_t)stream>>>(a, out);
  return check_launch("smooth_fwd_kernel");
}

extern "C" int pd_smooth_loss_bwd(int B, int C, int H, int W, const float* disp, int64_t disp_stride_b, int64_t disp_stride_h,
                                  const float* img, int64_t img_stride_b, int64_t img_stride_c, int64_t img_stride_h, float gamma,
                                  const float* g_out, float* g_disp, pd_stream_t stream) {
  SmoothArgs a;
  if (int rc = smooth_args(a, B, C, H, W, disp, disp_stride_b, disp_stride_h, img, img_stride_b, img_stride_c,
                           img_stride_h, gamma)) return rc;
  PD_REQUIRE(g_out && g_disp, "NULL pointer");
  smooth_bwd_kernel<<<dim3(ceil_div(H * W, kBlock), B), kBlock, 0, (hipStream_t)stream>>>(a, g_out, g_disp);
  return check_launch("smooth_bwd_kernel");
}

extern "C" int pd_smooth_loss_bwd_padded(int B, int C, int H, int W, int x_pad, const float* disp, int64_t disp_stride_b,
                                         int64_t disp_stride_h, const float* img, int64_t img_stride_b, int64_t img_stride_c,
                                         int64_t img_stride_h, float gamma, const float* g_out, float* g_disp,
                                         pd_stream_t stream) {
  SmoothArgs a;
  if (int rc = smooth_args(a, B, C, H, W, disp, disp_stride_b, disp_stride_h, img, img_stride_b, img_stride_c,
                           img_stride_h, gamma)) return rc;
  PD_REQUIRE(g_out && g_disp && x_pad >= 0 && (long)H * (W + x_pad) < (1L << 31), "NULL pointer / bad padding");
  smooth_bwd_padded_kernel<<<dim3(ceil_div(H * (W + x_pad), kBlock), B), kBlock, 0, (hipStream_t)stream>>>(a, g_out, g_disp, x_pad);
  return check_launch("smooth_bwd_padded_kernel");
}
