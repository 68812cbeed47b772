// Helpers shared by the fused decoder tails (pd_decoder_tail.hip: DepthDecoder, softmax; pd_plade_tail.hip: PladeNet,
// alpha compositing): sigma's sigmoid + clamp, and PX pixels per thread as one 16-byte access per tensor and plane.
#pragma once
#include <initializer_list>
#include <stdint.h>

#include "pd_common.h"

namespace pd {

constexpr float kTailSigmaMin = 0.01f, kTailSigmaMax = 1.0f;

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float clamp_sigma(float s) { return fminf(fmaxf(s, kTailSigmaMin), kTailSigmaMax); }

// PX pixels per thread: 4 (one 16-byte access per tensor and plane) when H*W is a multiple of 4 and every pointer is
// 16-byte aligned, else 1.  The arithmetic is per pixel either way.
template <int PX>
struct Px {
  float v[PX];
};
template <int PX>
__device__ __forceinline__ Px<PX> ldv(const float* __restrict__ p) {
  Px<PX> r;
  if (PX == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1 % PX] = t.y; r.v[2 % PX] = t.z; r.v[3 % PX] = t.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int PX>
__device__ __forceinline__ void stv(float* __restrict__ p, const Px<PX>& r) {
  if (PX == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % PX], r.v[2 % PX], r.v[3 % PX]);
  else p[0] = r.v[0];
}
template <int PX>
__device__ __forceinline__ Px<PX> splat(float x) {
  Px<PX> r;
#pragma unroll
  for (int j = 0; j < PX; ++j) r.v[j] = x;
  return r;
}

// 4 pixels per thread when every row of 4 is whole and 16-byte aligned in every tensor involved
static inline int tail_px(int H, int W, std::initializer_list<const void*> ptrs) {
  if (((long)H * W) % 4 != 0) return 1;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return 1;
  return 4;
}

}  // namespace pd
