// Batch doubling of Trainer.add_flip_right_inputs (reference trainer.py:252-276): every image-like
// input x of the stereo pair becomes cat([x_own, flip(x_other, -1)], dim 0) — the mirrored right image is a valid left
// image.  The reference issues a flip (copy) and a cat (second copy) per tensor; here one kernel writes the doubled
// batch directly.  `negate_c0` covers the grid tensor, whose x-coordinate channel changes sign under the mirror
// (trainer.py:258-260).  Pure data movement: bit-exact.
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

}  // namespace pd

using namespace pd;

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
