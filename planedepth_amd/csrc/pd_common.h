// Shared device/host helpers for the planedepth_hip kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>

#include "planedepth_hip.h"

namespace pd {

constexpr int kWave = 64;    // CDNA wavefront width
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

// ---- host side -------------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Process-environment tuning / A-B switches.  They are read ONCE, when the library is first used (pd_version() or the
// first launch), never on the launch path: an entry point's behaviour cannot change between two calls of a process, and
// workspace sizing and launch always agree.  Kernel selection that tests need per call goes through pd_sweep_desc.impl.
struct Switches {
  bool no_rowpair;    // PD_NO_ROWPAIR=1: forward without row pairs
  bool pp_rows_off;   // PD_PP_ROWS=0: post-process kernels in per-pixel gather form
  bool pp_seg_off;    // PD_PP_SEG=0: post-process kernels without the segment form (one pixel per lane, planes sampled twice)
  bool pp_chain_off;  // PD_PP_CHAIN=0: pd_post_process through the single warps (the softmax's [B,N,H,W] intermediate in memory)
  int row_waves;      // PD_ROW_WAVES=n: waves per row workgroup of the row-shift kernels (0 = default)
  int uni_chunk;      // PD_UNI_CHUNK=n: images per launch of the plane-uniform backward passes (0 = whole batch)
  bool fwd_stream;    // PD_FWD_STREAM=0: the headline forward on the plane-group row-shift kernel instead of the segment-stream one
};
const Switches& switches();

// LDS a workgroup of the CURRENT device can be given (bytes: hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 160 KB on
// gfx950, which is also the answer when there is no device to ask).  Queried once per device ordinal and kept in an atomic
// slot of a per-device table: a process that drives several GPUs (or several threads on one) gets each device's own answer.
// The kernels' applicability tests use it instead of a constant, so a device or partition with less falls back to a kernel
// that fits instead of failing at launch.
size_t device_lds_bytes();
// Dynamic-LDS limit of one kernel instantiation, per device ordinal (0 = the 64 KB every kernel starts with).
constexpr int kMaxDevices = 64;
struct LdsGrant {
  std::atomic<size_t> granted[kMaxDevices];
  std::mutex slow;   // serialises the rare raise (first launch of a shape that needs more than was granted before)
};
// Raise the kernel's dynamic-LDS limit on the current device to `bytes` if that is above what was granted there before;
// PD_OK, or PD_ERR_UNSUPPORTED with the error text set.  After the first call for a shape: one relaxed atomic load.
int grant_dynamic_lds(const void* kernel, size_t bytes, LdsGrant* grant, const char* what);

#define PD_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      pd::set_error(__VA_ARGS__);    \
      return PD_ERR_ARG;             \
    }                                \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- device side -----------------------------------------------------------------------------------------------
// Workgroup i of a launch runs on XCD i % 8, and every XCD has its own L2.  The pixel-linear gather kernels read a
// footprint of two (or a few) source rows per target row, so rows dealt round-robin to the XCDs are fetched into several
// L2s; handing each XCD one contiguous band of the image instead lets the vertical neighbours share their lines.
// Measured at 8x49x192x640 (scripts/gpu_r2_band.sh): plane-uniform kernels fwd 0.161 -> 0.148 ms, bwd 0.538 -> 0.518 ms;
// the general per-plane kernels do NOT want it (disp_warp forward 0.233 -> 0.571 ms: their taps stay in the row, and
// a band per XCD takes the DRAM-page sharing between concurrently running neighbours away), so only the former use it.
#ifndef PD_XCD_BAND
#define PD_XCD_BAND 1
#endif
constexpr int kXcds = 8;
__device__ __forceinline__ int xcd_banded(int bx, int nblk) {
#if PD_XCD_BAND
  const int per = nblk / kXcds;   // blocks per band; a remainder keeps its place at the end
  if (bx < per * kXcds) return (bx % kXcds) * per + bx / kXcds;
#endif
  return bx;
}

// Bilinear footprint of one sample under F.grid_sample(align_corners=True) semantics: the four taps
// (x0,y0) (x0+1,y0) (x0,y0+1) (x0+1,y0+1) with torch's weights  nw=(x1-ix)(y1-iy), ne=(ix-x0)(y1-iy), ...
struct Tap {
  int x0, y0;
  float wx0, wx1, wy0, wy1;  // (x1-ix), (ix-x0), (y1-iy), (iy-y0)
  bool vx0, vx1, vy0, vy1;   // column x0 / x0+1 and row y0 / y0+1 inside the image
};

__device__ __forceinline__ Tap make_tap(float ix, float iy, int W, int H) {
  Tap t;
  const float xf = floorf(ix), yf = floorf(iy);
  const float xf1 = xf + 1.0f, yf1 = yf + 1.0f;
  t.wx0 = xf1 - ix;
  t.wx1 = ix - xf;
  t.wy0 = yf1 - iy;
  t.wy1 = iy - yf;
  // validity is decided on the float value (NaN / huge coordinates are simply "outside")
  t.vx0 = (xf >= 0.0f) && (xf <= (float)(W - 1));
  t.vx1 = (xf1 >= 0.0f) && (xf1 <= (float)(W - 1));
  t.vy0 = (yf >= 0.0f) && (yf <= (float)(H - 1));
  t.vy1 = (yf1 >= 0.0f) && (yf1 <= (float)(H - 1));
  t.x0 = (int)fminf(fmaxf(xf, -2.0f), (float)W);
  t.y0 = (int)fminf(fmaxf(yf, -2.0f), (float)H);
  return t;
}

// Sample one [H,W] plane.  Out-of-image taps contribute zero (padding_mode="zeros").
__device__ __forceinline__ float bilinear(const float* __restrict__ p, const Tap& t, int W) {
  const float* r0 = p + (long)t.y0 * W + t.x0;
  const float* r1 = r0 + W;
  const float nw = (t.vx0 && t.vy0) ? r0[0] : 0.0f;
  const float ne = (t.vx1 && t.vy0) ? r0[1] : 0.0f;
  const float sw = (t.vx0 && t.vy1) ? r1[0] : 0.0f;
  const float se = (t.vx1 && t.vy1) ? r1[1] : 0.0f;
  return nw * (t.wx0 * t.wy0) + ne * (t.wx1 * t.wy0) + sw * (t.wx0 * t.wy1) + se * (t.wx1 * t.wy1);
}

// d(sample)/d(ix), d(sample)/d(iy) in pixel units (torch's grid_sampler_2d_backward before the (size-1)/2 factor).
__device__ __forceinline__ void bilinear_grad(const float* __restrict__ p, const Tap& t, int W, float& dx, float& dy) {
  const float* r0 = p + (long)t.y0 * W + t.x0;
  const float* r1 = r0 + W;
  const float nw = (t.vx0 && t.vy0) ? r0[0] : 0.0f;
  const float ne = (t.vx1 && t.vy0) ? r0[1] : 0.0f;
  const float sw = (t.vx0 && t.vy1) ? r1[0] : 0.0f;
  const float se = (t.vx1 && t.vy1) ? r1[1] : 0.0f;
  dx = (ne - nw) * t.wy0 + (se - sw) * t.wy1;
  dy = (sw - nw) * t.wx0 + (se - ne) * t.wx1;
}

// Value and both derivatives from one set of loads.
__device__ __forceinline__ float bilinear_vg(const float* __restrict__ p, const Tap& t, int W, float& dx, float& dy) {
  const float* r0 = p + (long)t.y0 * W + t.x0;
  const float* r1 = r0 + W;
  const float nw = (t.vx0 && t.vy0) ? r0[0] : 0.0f;
  const float ne = (t.vx1 && t.vy0) ? r0[1] : 0.0f;
  const float sw = (t.vx0 && t.vy1) ? r1[0] : 0.0f;
  const float se = (t.vx1 && t.vy1) ? r1[1] : 0.0f;
  dx = (ne - nw) * t.wy0 + (se - sw) * t.wy1;
  dy = (sw - nw) * t.wx0 + (se - ne) * t.wx1;
  return nw * (t.wx0 * t.wy0) + ne * (t.wx1 * t.wy0) + sw * (t.wx0 * t.wy1) + se * (t.wx1 * t.wy1);
}

// The same footprint in "kernel form": clamped element offsets + coefficients that are zero for taps outside the image,
// shared by every tensor sampled at this position.  Sampling is then four unconditional loads and four FMAs per tensor
// (value), four more per derivative — no validity selects at the loads (the general sweep kernels are VALU-bound and
// sample five tensors per pixel and plane).  A zero coefficient times a finite border value is an exact zero, which is
// what padding_mode="zeros" asks for.
struct TapK {
  unsigned o00, o01, o10, o11;  // BYTE offsets of (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1) inside an [H,W] plane: unsigned
                                // 32-bit, so a load is "uniform base + zero-extended lane offset" (no 64-bit lane math)
  float w00, w01, w10, w11;    // value weights
  float x00, x01, x10, x11;    // d value / d ix coefficients
  float y00, y01, y10, y11;    // d value / d iy coefficients
};
__device__ __forceinline__ TapK tap_kernel(const Tap& t, int W, int H) {
  TapK k;
  const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
  const int ya = min(max(t.y0, 0), H - 1) * W, yb = min(max(t.y0 + 1, 0), H - 1) * W;
  k.o00 = (unsigned)(ya + xa) << 2; k.o01 = (unsigned)(ya + xb) << 2;
  k.o10 = (unsigned)(yb + xa) << 2; k.o11 = (unsigned)(yb + xb) << 2;
  const float m00 = (t.vx0 && t.vy0) ? 1.0f : 0.0f, m01 = (t.vx1 && t.vy0) ? 1.0f : 0.0f;
  const float m10 = (t.vx0 && t.vy1) ? 1.0f : 0.0f, m11 = (t.vx1 && t.vy1) ? 1.0f : 0.0f;
  k.w00 = t.wx0 * t.wy0 * m00; k.w01 = t.wx1 * t.wy0 * m01; k.w10 = t.wx0 * t.wy1 * m10; k.w11 = t.wx1 * t.wy1 * m11;
  k.x00 = -t.wy0 * m00; k.x01 = t.wy0 * m01; k.x10 = -t.wy1 * m10; k.x11 = t.wy1 * m11;
  k.y00 = -t.wx0 * m00; k.y01 = -t.wx1 * m01; k.y10 = t.wx0 * m10; k.y11 = t.wx1 * m11;
  return k;
}
__device__ __forceinline__ float at_byte(const float* __restrict__ p, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + byte_off);
}
__device__ __forceinline__ float sample_k(const float* __restrict__ p, const TapK& k) {
  return at_byte(p, k.o00) * k.w00 + at_byte(p, k.o01) * k.w01 + at_byte(p, k.o10) * k.w10 + at_byte(p, k.o11) * k.w11;
}
__device__ __forceinline__ float sample_vg_k(const float* __restrict__ p, const TapK& k, float& dx, float& dy) {
  const float nw = at_byte(p, k.o00), ne = at_byte(p, k.o01), sw = at_byte(p, k.o10), se = at_byte(p, k.o11);
  dx = nw * k.x00 + ne * k.x01 + sw * k.x10 + se * k.x11;
  dy = nw * k.y00 + ne * k.y01 + sw * k.y10 + se * k.y11;
  return nw * k.w00 + ne * k.w01 + sw * k.w10 + se * k.w11;
}

// Adjoint of `bilinear`: scatter g to the (valid) taps with hardware fp32 atomics (global_atomic_add_f32).
__device__ __forceinline__ void bilinear_scatter(float* __restrict__ p, const Tap& t, int W, float g) {
  float* r0 = p + (long)t.y0 * W + t.x0;
  float* r1 = r0 + W;
  if (t.vx0 && t.vy0) unsafeAtomicAdd(r0, g * (t.wx0 * t.wy0));
  if (t.vx1 && t.vy0) unsafeAtomicAdd(r0 + 1, g * (t.wx1 * t.wy0));
  if (t.vx0 && t.vy1) unsafeAtomicAdd(r1, g * (t.wx0 * t.wy1));
  if (t.vx1 && t.vy1) unsafeAtomicAdd(r1 + 1, g * (t.wx1 * t.wy1));
}

// Pixel -> normalised [-1,1] -> pixel, with the reference's exact op order (trainer.py:550-552 then grid_sample's
// un-normalisation).  The round trip is NOT the identity in fp32, so it is reproduced, not skipped.
// fp contraction is OFF here: fusing (a - 0.5) * 2 into fma(a, 2, -1) or (g + 1) * 0.5 into fma(g, 0.5, 0.5) changes
// the coordinate by an ulp, which is visible in the weights of taps that are almost out of view.
__device__ __forceinline__ float normalise(float px, float size_m1) {
#pragma clang fp contract(off)
  const float q = px / size_m1;
  const float h = q - 0.5f;
  return h * 2.0f;
}

__device__ __forceinline__ float unnormalise(float g, float size_m1) {
#pragma clang fp contract(off)
  const float s = g + 1.0f;
  const float h = s * 0.5f;
  return h * size_m1;
}

__device__ __forceinline__ float normalise_roundtrip(float px, float size_m1) {
  return unnormalise(normalise(px, size_m1), size_m1);
}

// Correctly rounded a / b from the correctly rounded reciprocal of b (Markstein's theorem; b = W-1 is an integer
// <= 2^24 and a is far from the over/underflow range, so no special cases arise).  Verified bit-for-bit against
// IEEE division over the whole coordinate range by tests/test_gpu_parity.py::test_fast_division_is_exact.
__device__ __forceinline__ float div_by(float a, float b, float rcp_b) {
  const float q0 = a * rcp_b;
  const float r = fmaf(-q0, b, a);
  return fmaf(r, rcp_b, q0);
}

__device__ __forceinline__ float refined_rcp(float b) {
  float y = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, y, 1.0f);
  return fmaf(e, y, y);
}

// normalise_roundtrip with the division by the (workgroup-uniform) size done through its refined reciprocal: the same
// bits as the IEEE division (Markstein) for every finite px; an infinite px is passed through as IEEE division would
// (the correction step alone would turn it into NaN).  (g + 1)/2 = fl(2h + 1)/2 = fl(h + 0.5): scaling by 2 commutes
// with rounding.  The IEEE expansion costs ~11 VALU instructions per coordinate, this 7.
__device__ __forceinline__ float normalise_roundtrip_rcp(float px, float size_m1, float rcp_size_m1) {
#pragma clang fp contract(off)
  const float q = div_by(px, size_m1, rcp_size_m1);
  const float h = q - 0.5f;
  const float hh = h + 0.5f;
  const float r = hh * size_m1;
  return (fabsf(px) == __builtin_inff()) ? px * size_m1 : r;   // +-inf stays +-inf (size_m1 > 0)
}

// One row of a 3x3 matrix times [x, y, 1] in the rounding order of torch's batched matmul on the host the golden vectors were
// captured on (layers.py:221 `torch.matmul(H_t2s, pix_coords_t)`; oneDNN / MKL sgemm walks k in order with fused multiply-adds):
// fl(fl(h1 * y + fl(h0 * x)) + h2).  Checked bit for bit against torch.matmul on the trainer fixtures' matrices (NOTEBOOK 11.3);
// written out so that the compiler's contraction choice (which product it fuses) cannot change it.
__device__ __forceinline__ float hrow_dot(float h0, float h1, float h2, float x, float y) {
#pragma clang fp contract(off)
  const float t0 = h0 * x;
  const float t1 = __builtin_fmaf(h1, y, t0);
  return t1 + h2;
}

__device__ __forceinline__ float hrow_dot4(float h0, float h1, float h2, float h3, float x, float y, float z, float w) {
#pragma clang fp contract(off)
  const float t0 = h0 * x;
  const float t1 = __builtin_fmaf(h1, y, t0);
  const float t2 = __builtin_fmaf(h2, z, t1);
  return __builtin_fmaf(h3, w, t2);
}

// The facing test's left-hand side (layers.py:223: `(matmul(inv_K, pix) * matmul(R, n)).sum(1)`): three products rounded on their
// own, added in index order — no contraction, so that a pixel on the horizon line falls on the reference's side of it.
__device__ __forceinline__ float facing_dot(float r0, float r1, float r2, float q0, float q1, float q2) {
#pragma clang fp contract(off)
  const float a = r0 * q0, b = r1 * q1, c = r2 * q2;
  return (a + b) + c;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// ---- DPP cross-lane helpers (GFX9 family: wave_shr / row_bcast controls exist on gfx950) ---------------------------
// They are VALU operand modifiers: no LDS traffic, unlike __shfl_* which lowers to ds_bpermute_b32.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_zero(float v) {  // lanes without a source (or masked rows) read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

// lane i receives lane i-1 across the whole wave; lane 0 receives 0
__device__ __forceinline__ float wave_shift_up1(float v) { return dpp_zero<0x138>(v); }
// lane i receives lane i+1 (wave_shl:1); lane 63 receives `fallback`
__device__ __forceinline__ int wave_from_next(int v, int fallback) {
  return __builtin_amdgcn_update_dpp(fallback, v, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ float wave_from_next(float v) { return dpp_zero<0x130>(v); }
// lane i receives lane i-1 (wave_shr:1); lane 0 receives `fallback`
__device__ __forceinline__ int wave_from_prev(int v, int fallback) {
  return __builtin_amdgcn_update_dpp(fallback, v, 0x138, 0xF, 0xF, false);
}

// ---- wave-cooperative adjoint of `bilinear` --------------------------------------------------------------------------
// The general backward is bound by the L2 atomic units (4 fp32 atomics per pixel, plane and tensor: measured at one
// atomic per clock and L2 channel).  Neighbouring target pixels usually land on neighbouring source pixels: where lane
// i+1's LEFT column is lane i's RIGHT column on the same source rows, lane i adds lane i+1's left-tap contributions to
// its own right-tap ones and lane i+1 skips its left atomics — two atomics per lane instead of four.  Both helpers
// must be called by all 64 lanes of the wave (`live` = this lane contributes at all).
struct ScatterPlan {
  bool take;   // absorb the next lane's left taps into my right taps
  bool taken;  // my left taps are absorbed by the previous lane
};
__device__ __forceinline__ ScatterPlan plan_scatter(const Tap& t, int W, bool live) {
  constexpr int kNone = -(1 << 30), kNone2 = -(1 << 29);  // never one apart from each other or from a real key
  // (source row, left column) as one integer; x0 is clamped to [-2, W], so a pitch of W + 4 keeps "keys one apart"
  // equivalent to "same row, adjacent column" (with pitch W, column -1 of row y+1 would follow column W-1 of row y)
  const int key = live ? t.y0 * (W + 4) + (t.x0 + 2) : kNone;
  const int give = (live && t.vx1) ? key : kNone2;                // my right column exists: I can absorb
  // The cross-lane reads come first, unconditionally: inside a short-circuit `&&` they would run under a partial
  // EXEC mask, and a disabled source lane reads as the fallback — the two sides of a pair would then disagree.
  const int next_key = wave_from_next(key, kNone);      // lane 63: kNone  -> never takes
  const int prev_give = wave_from_prev(give, kNone2);   // lane 0:  kNone2 -> never taken
  ScatterPlan p;
  p.take = (next_key == give + 1);
  p.taken = (prev_give + 1 == key);
  return p;
}
__device__ __forceinline__ void bilinear_scatter_wave(float* __restrict__ p, const Tap& t, int W, float g, bool live,
                                                      const ScatterPlan& sp) {
  const float cL0 = g * (t.wx0 * t.wy0), cR0 = g * (t.wx1 * t.wy0), cL1 = g * (t.wx0 * t.wy1), cR1 = g * (t.wx1 * t.wy1);
  const float nL0 = wave_from_next(cL0), nL1 = wave_from_next(cL1);   // the next lane's left taps (same addresses as my right)
  if (!live) return;
  float* r0 = p + (long)t.y0 * W + t.x0;
  float* r1 = r0 + W;
  // contributions that are exactly zero are skipped: rows whose vertical round trip is exact (3 of 4 in disp mode) have
  // wy1 == 0 and would otherwise spend half of their atomics on adding 0.0
  if (!sp.taken) {
    if (t.vx0 && t.vy0 && cL0 != 0.0f) unsafeAtomicAdd(r0, cL0);
    if (t.vx0 && t.vy1 && cL1 != 0.0f) unsafeAtomicAdd(r1, cL1);
  }
  const float a0 = sp.take ? cR0 + nL0 : cR0, a1 = sp.take ? cR1 + nL1 : cR1;
  if (t.vx1 && t.vy0 && a0 != 0.0f) unsafeAtomicAdd(r0 + 1, a0);
  if (t.vx1 && t.vy1 && a1 != 0.0f) unsafeAtomicAdd(r1 + 1, a1);
}

// Sum over the 64 lanes; the total is valid in lanes 48..63 (use lane 63).
__device__ __forceinline__ float wave_sum_hi(float v) {
  v += dpp_zero<0xB1>(v);         // quad_perm:[1,0,3,2]
  v += dpp_zero<0x4E>(v);         // quad_perm:[2,3,0,1]   -> quad sums
  v += dpp_zero<0x124>(v);        // row_ror:4
  v += dpp_zero<0x128>(v);        // row_ror:8             -> row (16-lane) sums in every lane
  v += dpp_zero<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_zero<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> row 3 holds the wave total
  return v;
}

// Sums of TWO values over the wave for the price of one reduction (gfx950 v_permlane32_swap): returns a's total in
// lane 31 and b's total in lane 63.
__device__ __forceinline__ float half_wave_sums_hi(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  float v = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // lanes 0-31: a_lo + a_hi, lanes 32-63: b_lo + b_hi
  v += dpp_zero<0xB1>(v);
  v += dpp_zero<0x4E>(v);
  v += dpp_zero<0x124>(v);
  v += dpp_zero<0x128>(v);
  v += dpp_zero<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3: lanes 31 / 63 hold the half-wave totals
  return v;
}

// LDS float add without a return value, kept out of the compiler's atomic optimiser (which wraps a uniform-address
// atomic in a readlane loop: ~20 instructions per call in the backward's plane loop).
__device__ __forceinline__ void lds_add(float* p, float v) {
  __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(p), v, 0, 0, false);
}

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
// sign(x) in two instructions: scale any normal |x| beyond 1, then clamp to [-1, 1] (v_med3_f32); sign(0) = 0 as in
// torch's abs backward.  Denormal differences (|x| < 2^-126) give a fraction instead of +-1: below fp32 data noise.
__device__ __forceinline__ float sgn_fast(float x) {
  return __builtin_amdgcn_fmed3f(x * 8.5070592e37f /* 2^126 */, -1.0f, 1.0f);
}

}  // namespace pd
