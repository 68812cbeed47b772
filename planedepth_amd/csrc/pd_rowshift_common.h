// Memory-access layer and row staging shared by the row-organised sweep kernels (pd_plane_sweep_rowshift.hip: one pixel
// per lane; pd_plane_sweep_rowquad.hip: four pixels per lane): dispatch order, buffer resources per (plane, row), tap
// loads, colour rows in LDS, per-plane shifts.
#pragma once
#include <stdlib.h>

#include "pd_rowgeom.h"

namespace pd {

constexpr int kMaxRowThreads = 1024;

// Tuning knobs (scripts/gpu_variants.sh builds variants with -D...): plane-group size, one-group-ahead prefetch and
// the occupancy the register allocator must leave room for (waves per SIMD), per kernel.
#ifndef PD_VARIANT
#define PD_VARIANT 0
#endif
#ifndef PD_FWD_U
#define PD_FWD_U 4
#endif
#ifndef PD_FWD_PF
#define PD_FWD_PF 1
#endif
#ifndef PD_FWD_OCC
#define PD_FWD_OCC 3  // (the single-row bodies alone fit 128 VGPRs = 4 waves per SIMD; the pair body needs ~147)
#endif
#ifndef PD_BWD_U
#define PD_BWD_U 2
#endif
#ifndef PD_PF_DEPTH
#define PD_PF_DEPTH 2  // groups in the software pipeline (2: one group ahead, colour taps prefetched too; 3: two ahead)
#endif
#ifndef PD_STORE_AUX
#define PD_STORE_AUX 0  // cache-policy bits of the gradient stores (experiments: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef PD_LOAD_AUX
#define PD_LOAD_AUX 0   // same for the tap loads
#endif
#ifndef PD_BWD_REVERSE
#define PD_BWD_REVERSE 1  // the backward walks the rows in the opposite order of the forward: what the forward read last is
#endif                    // what the backward reads first, and the other way round for the next step's forward — with the
                          // gradient stores streaming past the caches (PD_STREAM_STORE_AUX = nt) the 256 MB memory-side cache
                          // still holds those rows: step +2.4-2.8 % (forward 0.109 -> 0.105 ms, backward 0.178 -> 0.174), and
                          // the same inside the DDP training step.  (Round 1, write-back stores: no gain — the dirty gradient
                          // lines cycled the cache.  Forward bottom-up / backward top-down instead: +1 %, worse in the DDP step.)
#ifndef PD_BWD_HANDOVER
#define PD_BWD_HANDOVER 1  // lane 0 takes its left neighbour's hand-over out of LDS when it is already there
#endif
#ifndef PD_BWD_PF
#define PD_BWD_PF 1
#endif
#ifndef PD_TC_PREFETCH
#define PD_TC_PREFETCH 1  // colour taps (LDS) ride along with the prefetched plane group; 0: read when the group is reduced
#endif
#define PD_TC_IN_GROUP (PD_PF_DEPTH == 2 && PD_TC_PREFETCH)
#ifndef PD_BWD_PF_DEPTH
#define PD_BWD_PF_DEPTH PD_PF_DEPTH  // the backward's own pipeline depth (experiments)
#endif
#ifndef PD_BWD_OCC
#define PD_BWD_OCC 3
#endif
constexpr int kVariant = PD_VARIANT;
constexpr int kRowThreadsMax = 512;  // row workgroups use <= 8 waves (row_threads)

// Row handled by workgroup r of an image.  The dispatcher deals consecutive workgroups to the 8 XCDs round-robin; the
// banded mapping gives each XCD (its own L2) a contiguous band of rows instead of every 8th row.
// (image, row) of this workgroup: row-major over the images (all images' row 0, then row 1, ...).  Two effects:
// (1) the rows that need two source rows (inexact vertical round trip) cluster at small y, so they are dispatched
//     FIRST — longest jobs first instead of the last image's heavy rows starting in the last round (forward 0.148 ->
//     0.137 ms);
// (2) consecutive workgroups go to the 8 XCDs round-robin, so with B a multiple of 8 the rows y and y+1 of one image
//     (B workgroups apart) share an XCD and run at the same time: the second source row of an inexact row is its
//     neighbour's main row and mostly hits in that XCD's L2 (PMC at B = 8: backward HBM traffic 995 -> 909 MB).  Padding
//     B to a multiple of 8 to get this for every batch size was measured and rejected: the padding workgroups all land
//     on the same XCDs and leave them idle (B = 4: 4.1 k -> 2.4 k images/s).
// Variant 8 = the image-major order.
__device__ __forceinline__ int wg_image(int B, int H) {
  if (kVariant & 8) return blockIdx.y;
  return (int)((blockIdx.y * gridDim.x + blockIdx.x) % (unsigned)B);
}
__device__ __forceinline__ int wg_rowid(int B, int H) {
  if (kVariant & 8) return blockIdx.x;
  return (int)((blockIdx.y * gridDim.x + blockIdx.x) / (unsigned)B);
}
// PD_BWD_REVERSE: the backward walks the rows in the opposite order of the forward, so that what the forward touched last
// is still in the memory-side cache when autograd starts the backward right after it (see the switch above for the numbers).
__device__ __forceinline__ int bwd_rowid(int B, int H) {
  const int r = wg_rowid(B, H);
  return PD_BWD_REVERSE ? H - 1 - r : r;
}
__device__ __forceinline__ int block_row(int r, int H) {
  if ((kVariant & 1) && (H % 8 == 0)) return (r & 7) * (H >> 3) + (r >> 3);
  return r;
}

// The one-row bodies drop the multiplications by the vertical weight: they run only for rows whose single live source
// row is the row itself with weight exactly 1 (every row whose normalise -> un-normalise round trip is exact).  Anything
// else with one live row (weight 1 - eps at an image border) is rewritten as a two-row footprint with a zero second
// weight on the same row, which the two-row bodies handle in full generality.
// Vertical round-trip noise.  The reference's y -> normalise -> un-normalise chain returns y + e with |e| <= 6e-6 for a
// quarter of the rows (fp32 rounding; exact arithmetic gives y), which makes grid_sample blend in the NEXT source row
// with weight e.  (Forward values and per-pixel gradients always use that second row; only the ADJOINT's e-weighted
// term into the neighbouring row is dropped: pd_plane_sweep_rowshift.hip's header.)  Serving that second row exactly
// doubles the loads of those rows (measured at 8x49x192x640: forward
// 0.137 -> 0.104 ms, backward 0.31 -> 0.30 ms, HBM reads -20% without it).  It is served by default all the same:
// dropping it moves results by up to e * (range of the logits), measured 4e-5 (rgb_rec) .. 1e-4 (g_sigma) of the
// tensors' range on random inputs — the whole 1e-4 parity budget.  PD_IMPL_FAST_ROWS opts into dropping a second row
// whose weight is below 2^-16 (smooth network outputs make the difference far smaller than random test data does).
// row_eps: a second source row whose weight is BELOW it is dropped (0: none is — PD_IMPL_EXACT_ROWS).
__device__ __forceinline__ RowSel two_row_form(RowSel r, float row_eps) {
  if (row_eps > 0.0f && r.nrows == 2) {
    const bool a_main = r.wA >= r.wB;
    if ((a_main ? r.wB : r.wA) < row_eps) {
      r.nrows = 1;
      r.yA = a_main ? r.yA : r.yB;
      r.wA = 1.0f;
      r.wy_main = 1.0f;
      return r;
    }
  }
  if (r.nrows == 1 && (r.wA != 1.0f || r.wy_main != 1.0f)) {
    r.nrows = 2;
    r.yB = r.yA;
    r.wB = 0.0f;
  }
  return r;
}

// ---- memory access layer ------------------------------------------------------------------------------------------
// Row-sized buffer resources (SRD in SGPRs, built from workgroup-uniform values only) give three things at once:
//   * a 32-bit per-lane byte offset instead of 64-bit address arithmetic (the u64 adds were ~15% of all VALU cycles);
//   * hardware range checking: a tap left of column 0 (offset wraps to >= 2^31) or right of column W-1 reads as 0,
//     which IS grid_sample's padding_mode="zeros" — no validity compares, selects or clamped indices in the forward;
//   * loads without exec-mask branches, so a whole group's loads issue back to back.
typedef __amdgpu_buffer_rsrc_t Rsrc;

__device__ __forceinline__ Rsrc row_rsrc(const float* row, int W) {  // `row` must be wave-uniform
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, W * 4, 0x00020000);
}
// Same, for an address the compiler is known to keep in VGPRs (it then wraps every access in a waterfall loop): state
// the uniformity explicitly.  Not the default: where the address already lives in SGPRs this costs extra moves.
__device__ __forceinline__ Rsrc row_rsrc_uniform(const float* row, int W) {
  const uint64_t p = reinterpret_cast<uint64_t>(row);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), 0, W * 4, 0x00020000);
}
// Descriptor with an explicit extent: 0 bytes turns every access through it into a hardware no-op (how the backward
// skips a gradient nobody asked for without a branch per plane).
__device__ __forceinline__ Rsrc row_rsrc_bytes(const float* row, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(Rsrc r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store(Rsrc r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, PD_STORE_AUX);
}

// Offset of plane n inside one image's [N,H,W] block, in floats: a 32-bit product (the host checks N*H*W < 2^31) added
// to a per-workgroup 64-bit base — the full 64-bit (b*N + n)*HW product per plane and tensor was a third of the scalar
// instructions of the plane loop.
__device__ __forceinline__ const float* plane_ptr(const float* image_base, int n, int HW) {
  return image_base + (unsigned)(n * HW);
}
__device__ __forceinline__ float* plane_ptr(float* image_base, int n, int HW) {
  return image_base + (unsigned)(n * HW);
}

// The (up to) four taps of one scalar plane, loaded up-front.  Out-of-image taps come back as 0 from the hardware.
template <int NROWS>
struct Taps {
  float a0, a1, b0, b1;  // row A (x0, x0+1), row B (x0, x0+1)
};

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f buf_load2(Rsrc r, unsigned byte_off) {  // 8 bytes at any 4-byte-aligned offset
  return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, PD_LOAD_AUX));
}

// Both horizontal taps of a row come from ONE 8-byte load at column x0 (measured on gfx950, scripts/probes/buf_probe:
// unaligned 8/16-byte buffer loads work, the range check is per dword at the upper end, and an access that STARTS
// left of the row reads as all-zero).  The only column pair that needs help is x0 = -1, whose second tap (column 0) is
// inside the image: it is fetched as the first dword of a load at column 0.
struct TapPos {
  unsigned off;   // byte offset of the load
  bool edge;      // x0 == -1
};

__device__ __forceinline__ TapPos tap_pos(const ColTap& c) {
  TapPos p;
  p.edge = (c.x0 == -1);
  p.off = p.edge ? 0u : ((unsigned)c.x0 << 2);
  return p;
}

template <int NROWS>
__device__ __forceinline__ Taps<NROWS> load_taps(Rsrc rowA, Rsrc rowB, const TapPos& p) {
  Taps<NROWS> t;
  const v2f va = buf_load2(rowA, p.off);
  t.a0 = va.x; t.a1 = va.y;
  t.b0 = t.b1 = 0.0f;
  if (NROWS == 2) {
    const v2f vb = buf_load2(rowB, p.off);
    t.b0 = vb.x; t.b1 = vb.y;
  }
  return t;
}

// Apply the x0 = -1 fix-up after the data has arrived (kept out of the issue phase so it does not wait on the load).
template <int NROWS>
__device__ __forceinline__ void fix_edge(Taps<NROWS>& t, bool edge) {
  t.a1 = edge ? t.a0 : t.a1;
  t.a0 = edge ? 0.0f : t.a0;
  if (NROWS == 2) {
    t.b1 = edge ? t.b0 : t.b1;
    t.b0 = edge ? 0.0f : t.b0;
  }
}

// w0A/w1A(/w0B/w1B): horizontal weight x vertical weight, computed once per plane and shared by all five channels
struct TapW {
  float a0, a1, b0, b1;
};

template <int NROWS>
__device__ __forceinline__ TapW tap_weights(const ColTap& c, const RowSel& r, float live) {
  TapW w;
  const float w0 = c.w0 * live, w1 = c.w1 * live;  // live = 1, or 0 for a plane the padding mask removes
  w.a0 = (NROWS == 1) ? w0 : w0 * r.wA;           // one-row bodies run only where that row's weight is exactly 1
  w.a1 = (NROWS == 1) ? w1 : w1 * r.wA;
  w.b0 = w.b1 = 0.0f;
  if (NROWS == 2) {
    w.b0 = w0 * r.wB;
    w.b1 = w1 * r.wB;
  }
  return w;
}

template <int NROWS>
__device__ __forceinline__ float tap_value(const Taps<NROWS>& t, const TapW& w) {
  float v = t.a0 * w.a0 + t.a1 * w.a1;
  if (NROWS == 2) v += t.b0 * w.b0 + t.b1 * w.b1;
  return v;
}

// d value / d ix = (ne - nw) * wyA + (se - sw) * wyB   (out-of-image taps already read as zero)
template <int NROWS>
__device__ __forceinline__ float tap_dx(const Taps<NROWS>& t, const RowSel& r) {
  float d = t.a1 - t.a0;
  if (NROWS == 2) d = d * r.wA + (t.b1 - t.b0) * r.wB;
  return d;
}

// Colour rows in LDS: per live row W+4 float4 (r,g,b,-) with two zero guard cells on each side, so taps at
// x0 in [-2, W] need no validity handling either.  lds_off = byte offset of tap x0 in row A.
__device__ __forceinline__ unsigned colour_off(int x0, int W) {
  const int xc = min(max(x0, -2), W);  // v_med3_i32: far-out shifts land on a guard cell
  return (unsigned)(xc + 2) << 4;
}

// Raw colour taps of one plane (read from LDS in the issue phase, so their latency overlaps like the global loads')
template <int NROWS>
struct ColourTaps {
  float4 nw, ne, sw, se;
};

template <int NROWS>
__device__ __forceinline__ ColourTaps<NROWS> load_colour_taps(const char* __restrict__ lrgb, int W, unsigned off) {
  ColourTaps<NROWS> c;
  c.nw = *reinterpret_cast<const float4*>(lrgb + off);
  c.ne = *reinterpret_cast<const float4*>(lrgb + off + 16);
  if (NROWS == 2) {
    const unsigned rb = (unsigned)(W + 4) << 4;
    c.sw = *reinterpret_cast<const float4*>(lrgb + rb + off);
    c.se = *reinterpret_cast<const float4*>(lrgb + rb + off + 16);
  }
  return c;
}

template <int NROWS>
__device__ __forceinline__ void colour_values(const ColourTaps<NROWS>& t, const TapW& w, float& c0, float& c1,
                                              float& c2) {
  c0 = t.nw.x * w.a0 + t.ne.x * w.a1;
  c1 = t.nw.y * w.a0 + t.ne.y * w.a1;
  c2 = t.nw.z * w.a0 + t.ne.z * w.a1;
  if (NROWS == 2) {
    c0 += t.sw.x * w.b0 + t.se.x * w.b1;
    c1 += t.sw.y * w.b0 + t.se.y * w.b1;
    c2 += t.sw.z * w.b0 + t.se.z * w.b1;
  }
}

template <int NROWS>
__device__ __forceinline__ void colour_dx(const ColourTaps<NROWS>& t, const RowSel& r, float& d0, float& d1,
                                          float& d2) {
  d0 = t.ne.x - t.nw.x;
  d1 = t.ne.y - t.nw.y;
  d2 = t.ne.z - t.nw.z;
  if (NROWS == 2) {
    d0 = d0 * r.wA + (t.se.x - t.sw.x) * r.wB;
    d1 = d1 * r.wA + (t.se.y - t.sw.y) * r.wB;
    d2 = d2 * r.wA + (t.se.z - t.sw.z) * r.wB;
  }
}

// Shift of plane i along target row `yrow` of image b: sign * disparity clamped to +-(W+2) (beyond +-(W+1) nothing is in
// view either way).
__device__ __forceinline__ float staged_shift(const SweepArgs& a, int b, int i, int yrow) {
  const float lim = (float)(a.W + 2);
  const long di = (a.flags & PD_DISP_ROWS) ? ((long)b * a.N + i) * a.H + yrow : (long)b * a.N + i;
  const float sd = a.sign * a.plane[di];
  // PD_MASK_ROWS: a masked plane samples as all-zero features (trainer.py:580) — exactly what a plane shifted out
  // of view does (every tap is outside the row), so the row's mask value just overrides the shift
  const bool masked = a.mask_rows && a.mask_rows[((long)b * a.N + i) * a.H + yrow] == 0.0f;
  return (!masked && sd >= -lim && sd <= lim) ? sd : ((sd < 0.0f && !masked) ? -lim : lim);  // NaN -> +lim
}

// Stage the live source colour rows of image b into LDS as float4 (with zero guard cells), plus the per-plane shifts
// sdisp[n] (staged_shift).
template <int NROWS>
__device__ __forceinline__ void stage_row_constants(const SweepArgs& a, int b, const RowSel& r, float4* __restrict__ lrgb,
                                                    float* __restrict__ sdisp, int yrow) {
  const int W = a.W, HW = a.H * a.W, RS = W + 4;
  const float* srcb = a.src + (long)b * 3 * HW;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const float* p = srcb + (long)r.yA * W + x;
    lrgb[2 + x] = make_float4(p[0], p[HW], p[2 * HW], 0.0f);
    if (NROWS == 2) {
      const float* q = srcb + (long)r.yB * W + x;
      lrgb[RS + 2 + x] = make_float4(q[0], q[HW], q[2 * HW], 0.0f);
    }
  }
  if (threadIdx.x < 4) {
    const int g = (threadIdx.x < 2) ? threadIdx.x : W + threadIdx.x;  // cells 0,1 and W+2,W+3
    lrgb[g] = z;
    if (NROWS == 2) lrgb[RS + g] = z;
  }
  for (int i = threadIdx.x; i < a.N; i += blockDim.x) sdisp[i] = staged_shift(a, b, i, yrow);
}

// ---- shared by the source-ordered backward (pd_plane_sweep_rowstream.hip) and the segment-stream forward (pd_plane_sweep_fwdstream.hip) ----
// frac(s*d) closer than this to an integer: the plane takes the general path.  Worst-case error of the coordinate
// chain against exact arithmetic: fl(x + sd) <= ulp(2W)/2, the division, the two additions and the product by W-1
// each <= ulp(.)/2 scaled by W-1 — 5.4e-7 * W in total (3.1e-4 at W = 640); the threshold keeps a factor of 2.4-3
// up to W = 4096 (NOTEBOOK.md 3.6.3).
__device__ __forceinline__ float irregular_tol(int W) { return 2.5e-4f + 1.25e-6f * (float)W; }

// ix of the reference for target column xtf (an integer-valued float) under the shift sd: make_col_tap's chain
__device__ __forceinline__ float stream_ix(float xtf, float sd, float Wm1, float rcpWm1) {
#pragma clang fp contract(off)
  const float px = xtf + sd;
  const float q = div_by(px, Wm1, rcpWm1);
  const float h = q - 0.5f;
  const float hh = h + 0.5f;
  return hh * Wm1;
}


}  // namespace pd
