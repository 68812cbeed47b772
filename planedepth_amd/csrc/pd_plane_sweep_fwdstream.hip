// Segment-stream forward of the fused plane sweep: PD_WARP_DISP with one disparity per (image, plane) or per (image,
// plane, row), no per-pixel mask; softmax probabilities or, PD_RENDER_PROB, alpha compositing (reference trainer.py:540-603
// + 728-742) — the headline forward.
//
// The plane-group forward (pd_plane_sweep_rowshift.hip) keeps two groups of four planes in registers per wave (taps,
// colour taps, coordinates: 148 VGPRs = 3 waves per SIMD, 12 per CU) and splits a row's segments and planes over four
// waves that meet in LDS.  Measured: its load phase and its arithmetic are both latency-bound at that occupancy (all
// arithmetic compiled out: 0.136 ms; all tap loads compiled out: 0.092 ms; 30 VALU per pixel and plane are 0.02 ms of
// issue time), and neither persistent workgroups nor overlapped row staging move it (docs/archive/experiments/
// pd_plane_sweep_rowpersist.hip.txt, NOTEBOOK.md 9.1).  This kernel is the forward counterpart of the row-stream backward's loop (0.123 -> 0.104 ms isolated at 8x49x192x640, NOTEBOOK.md 9.2):
//   * one wave owns one 128-pixel segment of a target row and ALL planes: no split of the online softmax, no partial
//     sums through LDS, no barrier after the row's constants are staged, one pipeline fill per wave;
//   * a lane owns two adjacent target pixels; the taps of both on plane n are the three source values at xt + k ..
//     xt + k + 2, k = floor(s d_n): ONE 12-byte load per tensor and live source row (4-byte aligned; the buffer
//     descriptor's range check is padding_mode = "zeros"), half the memory instructions per pixel of the 8-byte form;
//   * one plane per iteration behind a register ring of PD_FS_D1 iterations of loads (6 VGPRs per slot): the state per
//     wave is two accumulator sets + the ring, 94 VGPRs instead of 148;
//   * a workgroup serves PD_FS_ROWS = 3 target rows (15 waves at W = 640, one workgroup per CU) — neighbours, grouped so that the
//     second source row of an inexact row is the main row of a team in the same workgroup wherever three rows allow
//     (fwdstream_rows; consecutive rows otherwise): the waves read the rows of every plane at about the same time;
//   * round 5 (profiles/r05_fwd_ladder.md): the workgroup is PERSISTENT — it walks its share of the launch's (image, row group)
//     items, the five waves of a row slot ("team") staging the next item's row into the slot's second LDS buffer as soon as the
//     TEAM is done and meeting only each other (an LDS counter) — no workgroup barrier between items, nothing idles while the
//     slowest wave of the slowest row finishes; wave priorities rotate every four planes (the SIMD serves its oldest wave
//     first; left alone, the waves of a team drift 40 % apart); a plane's staged shift is read one iteration ahead and lives
//     in SGPRs; the softmax runs against a fixed per-pixel reference (no rescale branch) with an exact fallback;
//   * the colour taps come out of LDS (packed float4 row, vertically pre-blended for rows with two live source rows, as
//     the row-stream backward stages it) when the plane is reduced;
//   * PD_RENDER_PROB: the planes of a pixel arrive in order in one wave, so the transmittance is a register.
//
// Exactness.  The pairing "left tap of target xt on plane n = source column xt + k" is the row-stream backward's premise
// (pd_rowshift_common.h: stream_ix, irregular_tol; NOTEBOOK.md 3.6.3): planes whose frac(s d) is closer than irregular_tol
// to an integer take a general per-pixel path (exact floor(ix) per pixel, dword loads), and so does the one segment per
// plane that straddles column 0 under a negative shift (a 12-byte load that STARTS left of the row reads as zeros as a
// whole).  Weights and samples are the reference's expressions in the row-shift forward's order.
#include <stdlib.h>

#include "pd_rowshift_common.h"

namespace pd {

#ifndef PD_FS_D1
#define PD_FS_D1 3   // prefetch depth in planes, one live source row (6 VGPRs per slot)
#endif
#ifndef PD_FS_D2
#define PD_FS_D2 1   // two live source rows (12 VGPRs per slot): this body sets the kernel's register count (depth 2: 108 VGPRs)
#endif
// Registers: a workgroup is 15 waves at W = 640 (three rows x five segments) and at most 16 in general, so one workgroup per CU is
// all that ever fits and the allocator may use the 128 VGPRs that four waves per SIMD leave (launch bound = the 1024-thread
// maximum).  Rounds 4's 96-register cap (five waves per SIMD) bought nothing — two such workgroups never fitted a CU.

#ifndef PD_FS_REVERSE
#define PD_FS_REVERSE 0   // row groups dispatched bottom-up (the backward then walks top-down: PD_BWD_REVERSE 0)
#endif
#ifndef PD_FS_ABL
#define PD_FS_ABL 0   // timing experiments only (wrong results; scripts/gpu_r5_ladder.sh): 1 colour cells at a 16-byte lane stride,
#endif                // 2 no LDS colour reads, 4 no softmax / mixture arithmetic, 8 no output / stash stores, 16 every row as one
                      // source row, 32 no coordinate chain, 64 no tap loads, 128 no tap interpolation, 256 no staging loads
#ifndef PD_FS_SHRING
#define PD_FS_SHRING 1  // 1: a plane's staged shift is read from LDS ONE iteration before its tap loads are issued and kept in
#endif                  // scalar registers until the plane is reduced (one LDS read per iteration, off the critical path, instead of
                        // two round trips at the head of every iteration)
#ifndef PD_FS_PRIO
#define PD_FS_PRIO 1    // 1: wave priority rotating every PD_FS_PRIO_PERIOD x 4 planes (every wave of a SIMD leads for a quarter of them)
#endif
#ifndef PD_FS_PRIO_PERIOD
#define PD_FS_PRIO_PERIOD 1
#endif
#ifndef PD_FS_FIXREF
#define PD_FS_FIXREF 1  // 1: softmax with a FIXED per-pixel reference (the first plane's scaled logit) on the regular planes: no
#endif                  // lazy-rescale branch per pixel and plane (a compare, an exec-mask branch and the copies of all seven running
                        // sums at its join: a sixth of the loop's VALU instructions).  The largest exponent a pixel used is tracked
                        // (one v_max per plane); a wave in which any pixel went beyond 2^kFixRefLimit redoes its planes with the
                        // rescaling accumulator (per-pixel general path) — exact for any input, never taken for logits within
                        // +-60 of each other
#ifndef PD_FS_TRACE
#define PD_FS_TRACE 0   // diagnostics build: s_memtime stamps per wave (entry, staged, loop end, exit) + HW_ID / XCC_ID into a device
#endif                  // array read back through pd_debug_fs_trace (scripts/diag_fwd_trace.py)
#ifndef PD_FS_LDS_PAD
#define PD_FS_LDS_PAD 0   // timing experiments: extra LDS bytes per workgroup (caps the workgroups per CU)
#endif
#ifndef PD_FS_ROWS
#define PD_FS_ROWS 3   // consecutive target rows per workgroup where its 16 waves allow (each row: one wave per segment).  Measured
#endif                 // at 8x49x192x640, isolated / in the step: 1 row 0.114 / 0.128 ms, 2 rows 0.119 / 0.131, 3 rows 0.104 / 0.122
                       // — 15 waves that read three adjacent rows (7.5 KB) of every plane at about the same time
#ifndef PD_DIAGNOSTICS   // timing-ablation / trace code (results wrong by design) compiles only into a library that says so: pd_build_flags()
#if PD_FS_ABL || PD_FS_TRACE || PD_FS_LDS_PAD
#error "timing-ablation / trace switches need -DPD_DIAGNOSTICS as well (pd_build_flags() then reports the build)"
#endif
#endif
constexpr float kFixRefLimit = 90.0f;  // PD_FS_FIXREF: largest base-2 exponent of a softmax term before the wave falls back
constexpr int kFsSeg = 2 * kWave;      // target pixels per wave
constexpr int kFsGuard = 4;            // zero cells on each side of the colour row
constexpr int kFsThreadsMax = 1024;    // 16 waves: rows up to 2048 pixels

typedef float v3f __attribute__((ext_vector_type(3)));

#if PD_FS_TRACE
constexpr int kFsTraceWgs = 4096, kFsTraceWords = 6;   // per wave: 4 stamps, hw ids, (image << 16 | first row)
__device__ unsigned long long g_fs_trace[kFsTraceWgs * 16 * kFsTraceWords];
__device__ __forceinline__ void fs_stamp(int slot) {
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (wg < kFsTraceWgs && (threadIdx.x & (kWave - 1)) == 0)
    g_fs_trace[((long)wg * 16 + (threadIdx.x >> 6)) * kFsTraceWords + slot] = __builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void fs_stamp_ids(int b, int y) {
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (wg < kFsTraceWgs && (threadIdx.x & (kWave - 1)) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    unsigned long long* p = g_fs_trace + ((long)wg * 16 + (threadIdx.x >> 6)) * kFsTraceWords;
    p[4] = ((unsigned long long)xcc << 32) | hw;
    p[5] = ((unsigned long long)b << 32) | (unsigned)y;
  }
}
#else
__device__ __forceinline__ void fs_stamp(int) {}
__device__ __forceinline__ void fs_stamp_ids(int, int) {}
#endif

__device__ __forceinline__ v3f fs_load3(Rsrc r, unsigned byte_off) {
  return __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, (int)byte_off, 0, 0));
}

template <int NROWS>
struct FsTaps {   // the taps of one plane for the lane's two pixels: L[c .. c+2] per live source row
  float l[NROWS][3], s[NROWS][3];
  float dist[2];  // PD_RENDER_PROB: the decoder's inter-plane distances at the two TARGET pixels (trainer.py:587)
};

struct FsRow {    // workgroup-uniform
  int b, y, yA, yB;
  float wA, wB;
};

template <bool MIX, int NROWS>
__device__ __forceinline__ void fs_issue(FsTaps<NROWS>& g, const SweepArgs& a, const FsRow& r, int n, unsigned off, int HW) {
  const float* pl = plane_ptr(a.logits + (long)r.b * a.N * HW, n, HW);
  const v3f la = fs_load3(row_rsrc(pl + (long)r.yA * a.W, a.W), off);
  g.l[0][0] = la.x; g.l[0][1] = la.y; g.l[0][2] = la.z;
  if (NROWS == 2) {
    const v3f lb = fs_load3(row_rsrc(pl + (long)r.yB * a.W, a.W), off);
    g.l[NROWS - 1][0] = lb.x; g.l[NROWS - 1][1] = lb.y; g.l[NROWS - 1][2] = lb.z;
  }
  if (MIX) {
    const float* ps = plane_ptr(a.sigma + (long)r.b * a.N * HW, n, HW);
    const v3f sa = fs_load3(row_rsrc(ps + (long)r.yA * a.W, a.W), off);
    g.s[0][0] = sa.x; g.s[0][1] = sa.y; g.s[0][2] = sa.z;
    if (NROWS == 2) {
      const v3f sb = fs_load3(row_rsrc(ps + (long)r.yB * a.W, a.W), off);
      g.s[NROWS - 1][0] = sb.x; g.s[NROWS - 1][1] = sb.y; g.s[NROWS - 1][2] = sb.z;
    }
  }
}

// The lane's two pixels on a plane that does not qualify for the 12-byte form: exact floor(ix) per pixel, dword loads
// (range-checked: an out-of-image tap reads as zero), colour taps from the guard-celled LDS row.
// One plane's samples of pixel i enter its running sums: the softmax, or (RENDER) front-to-back alpha compositing —
// the planes of a pixel arrive in order in this kernel (one wave walks them all), so the transmittance is one register.
template <bool MIX, bool RENDER>
__device__ __forceinline__ void fs_accumulate(FwdAcc& acc, RenderState& rs, float l, float s, float c0, float c1, float c2,
                                              float t0, float t1, float t2, float ea, bool automask, float dist, bool last) {
  if (RENDER) mixture_accumulate<MIX>(acc, render_prob(rs, render_alpha(l, dist, last)), s, c0, c1, c2, t0, t1, t2, ea, automask);
  else fwd_accumulate<MIX>(acc, l, s, c0, c1, c2, t0, t1, t2, ea, automask);
}

template <bool MIX, int NROWS, bool RENDER>
__device__ __forceinline__ void fs_general_plane(const SweepArgs& a, const FsRow& r, const float4* __restrict__ col, int n,
                                                 float sd, float xt0f, int HW, float Wm1, float rcpWm1, const float* t,
                                                 const float* ea, bool automask, FwdAcc* acc, RenderState* rs,
                                                 const float* dist) {
  const float* pl = plane_ptr(a.logits + (long)r.b * a.N * HW, n, HW);
  const float* ps = MIX ? plane_ptr(a.sigma + (long)r.b * a.N * HW, n, HW) : pl;
  const Rsrc lA = row_rsrc(pl + (long)r.yA * a.W, a.W), lB = row_rsrc(pl + (long)r.yB * a.W, a.W);
  const Rsrc sA = row_rsrc(ps + (long)r.yA * a.W, a.W), sB = row_rsrc(ps + (long)r.yB * a.W, a.W);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const ColTap ct = make_col_tap(xt0f + (float)i + sd, Wm1, rcpWm1);
    // Columns right of the row are dropped by the descriptor's range check.  Columns LEFT of it are not all: byte offset
    // -4 (column -1) plus the access size wraps to 0 in the 32-bit check and counts as in range — such a tap gets a zero
    // weight and a load at column 0 instead (scripts/probes/buf_probe.hip; the row-shift forward's x0 = -1 fix-up).
    const unsigned o0 = (unsigned)max(ct.x0, 0) << 2, o1 = (unsigned)max(ct.x0 + 1, 0) << 2;
    const float w0 = (ct.x0 >= 0) ? ct.w0 : 0.0f, w1 = (ct.x0 + 1 >= 0) ? ct.w1 : 0.0f;
    const float wa0 = (NROWS == 1) ? w0 : w0 * r.wA, wa1 = (NROWS == 1) ? w1 : w1 * r.wA;
    float l = buf_load(lA, o0) * wa0 + buf_load(lA, o1) * wa1;
    float s = 0.0f;
    if (MIX) s = buf_load(sA, o0) * wa0 + buf_load(sA, o1) * wa1;
    if (NROWS == 2) {
      const float wb0 = w0 * r.wB, wb1 = w1 * r.wB;
      l += buf_load(lB, o0) * wb0 + buf_load(lB, o1) * wb1;
      if (MIX) s += buf_load(sB, o0) * wb0 + buf_load(sB, o1) * wb1;
    }
    const int cell = min(max(ct.x0, -kFsGuard), a.W + 2) + kFsGuard;
    const float4 ca = col[cell], cb = col[cell + 1];
    const float c0 = ca.x * ct.w0 + cb.x * ct.w1, c1 = ca.y * ct.w0 + cb.y * ct.w1, c2 = ca.z * ct.w0 + cb.z * ct.w1;
    fs_accumulate<MIX, RENDER>(acc[i], rs[i], l, s, c0, c1, c2, t[i], t[2 + i], t[4 + i], ea[i], automask, dist[i], n == a.N - 1);
  }
}

// Stage one target row (b, y) for the waves that serve it (`tix` / `nthr` = this thread's index among them): the source
// colour row (vertically blended where the row has two live source rows) with zero guard cells, and the per-plane shifts.
// The number of live rows is a run-time value here: every wave of the workgroup runs this one function and meets at the
// kernel's ONE barrier, whatever its row's footprint (the bodies below, specialised by footprint, contain no barrier).
__device__ __forceinline__ void fs_stage_row(const SweepArgs& a, const RowSel& row, int b, int y, int tix, int nthr,
                                             float4* __restrict__ col, int2* __restrict__ shift) {
  const int W = a.W, N = a.N, HW = a.H * a.W;
  const int CW = W + 2 * kFsGuard;
  const bool two = row.nrows == 2;
  const float* srcb = a.src + (long)b * 3 * HW;
  auto blended = [&](int x) {   // source colour at column x of the (vertically blended) row; zero outside the row
    float4 cc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x >= 0 && x < W) {
      if (PD_FS_ABL & 256) return make_float4((float)x, 0.5f, 0.25f, 0.0f);
      const float* p = srcb + (long)row.yA * W + x;
      cc = make_float4(p[0], p[HW], p[2 * HW], 0.0f);
      if (two) {   // fl(B*wB + fl(A*wA)): the rounding the row-stream backward stages (its knife-edge note applies here too)
        const float* q = srcb + (long)row.yB * W + x;
        cc = make_float4(fmaf(q[0], row.wB, cc.x * row.wA), fmaf(q[HW], row.wB, cc.y * row.wA), fmaf(q[2 * HW], row.wB, cc.z * row.wA), 0.0f);
      }
    }
    return cc;
  };
  for (int cidx = tix; cidx < CW && y < a.H; cidx += nthr) {
    const int x = cidx - kFsGuard;
    col[cidx] = blended(x);
  }
  const float tol = irregular_tol(W);
  for (int i = tix; i < N && y < a.H; i += nthr) {
    const float sd = staged_shift(a, b, i, y);
    const float fl = floorf(sd), fr = sd - fl;
    const bool inview = fabsf(sd) < (float)(W + 1);
    const int irr = (inview && (fr < tol || fr > 1.0f - tol)) ? 1 : 0;
    shift[i] = make_int2(__float_as_int(sd), (int)fl * 2 + irr);
  }
}

// One wave's segment `seg` of the staged target row (b, y), all planes.  No barrier inside.
template <bool MIX, bool AUTO, int NROWS, bool RENDER>
__device__ __forceinline__ float fwdstream_body(const SweepArgs& a, const RowSel& row, int b, int y, int seg,
                                                float4* __restrict__ col, int2* __restrict__ shift,
                                                float* __restrict__ rgb_rec, float* __restrict__ ph_map,
                                                float* __restrict__ stash) {
  constexpr int D = (NROWS == 1) ? (RENDER ? PD_FS_D1 - 1 : PD_FS_D1) : PD_FS_D2;   // (compositing: the ring also carries dists — one slot less keeps it out of scratch)
  const int W = a.W, N = a.N, HW = a.H * a.W;
  FsRow r;
  r.y = y;
  r.b = b;
  r.yA = row.yA; r.yB = (NROWS == 2) ? row.yB : row.yA;
  r.wA = row.wA; r.wB = (NROWS == 2) ? row.wB : 0.0f;
  const bool automask = MIX ? AUTO : (bool)(a.flags & PD_AUTOMASK);
  const float Wm1 = (float)(W - 1), rcpWm1 = refined_rcp(Wm1);
  const float* srcb = a.src + (long)r.b * 3 * HW;

  // ---- this wave's segment -----------------------------------------------------------------------------------------
  const int lane = threadIdx.x & (kWave - 1);
  const int xt0 = seg * kFsSeg + lane * 2;
  const float xt0f = (float)xt0;
  const bool live = xt0 < W;                 // W is even: a lane's two pixels are inside or outside together
  const int pix = r.y * W + (live ? xt0 : 0);
  float t[6], ea[2] = {0.0f, 0.0f};          // target colour (r0 r1 g0 g1 b0 b1), 3 x identity-reprojection error
  {
    const float* tp = a.tgt + (long)r.b * 3 * HW + pix;
    const float2 t0 = *reinterpret_cast<const float2*>(tp), t1 = *reinterpret_cast<const float2*>(tp + HW),
                 t2 = *reinterpret_cast<const float2*>(tp + 2 * HW);
    t[0] = t0.x; t[1] = t0.y; t[2] = t1.x; t[3] = t1.y; t[4] = t2.x; t[5] = t2.y;
    if (automask) {
      const float* sp = srcb + pix;
      const float2 s0 = *reinterpret_cast<const float2*>(sp), s1 = *reinterpret_cast<const float2*>(sp + HW),
                   s2 = *reinterpret_cast<const float2*>(sp + 2 * HW);
      ea[0] = fabsf(s0.x - t[0]) + fabsf(s1.x - t[2]) + fabsf(s2.x - t[4]);
      ea[1] = fabsf(s0.y - t[1]) + fabsf(s1.y - t[3]) + fabsf(s2.y - t[5]);
    }
  }
  FwdAcc acc[2];
  RenderState rs[2];
  float dmax[2] = {-INFINITY, -INFINITY};   // PD_FS_FIXREF: largest exponent used so far
  FsTaps<NROWS> g[D + 1];
  int pn = 0;
#if PD_FS_SHRING
  int sh_sd[D + 1], sh_kk[D + 1];          // wave-uniform (SGPRs): the shifts of the planes whose taps are in flight
  int2 sh_next = shift[0];       // LDS read in flight: the shift of the next plane to be prefetched
#endif
  auto prefetch = [&](FsTaps<NROWS>& grp, int slot) {
    const int n = min(pn, N - 1);   // past the end: re-load the last plane (unused) — unconditional issue keeps the wait counts right
#if PD_FS_SHRING
    sh_sd[slot] = __builtin_amdgcn_readfirstlane(sh_next.x);
    sh_kk[slot] = __builtin_amdgcn_readfirstlane(sh_next.y);
    sh_next = shift[min(pn + 1, N - 1)];
    const int k = sh_kk[slot] >> 1;
#else
    const int k = __builtin_amdgcn_readfirstlane(shift[n].y) >> 1;
#endif
    if (PD_FS_ABL & 64) {   // timing only: no tap loads, lane- and plane-dependent stand-ins
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        grp.l[0][q] = grp.l[NROWS - 1][q] = xt0f * 1e-3f + (float)(k + q) * 1e-2f;
        grp.s[0][q] = grp.s[NROWS - 1][q] = 0.5f + xt0f * 1e-4f + (float)(k - q) * 1e-4f;
      }
    } else
    fs_issue<MIX, NROWS>(grp, a, r, n, (unsigned)(xt0 + k) << 2, HW);
    if (RENDER) {   // unshifted, coalesced: read where the pixels are, not where they sample (the last plane has none: alpha = 1)
      const float2 d2 = *reinterpret_cast<const float2*>(a.dists + ((long)r.b * (N - 1) + min(n, N - 2)) * HW + pix);
      grp.dist[0] = d2.x; grp.dist[1] = d2.y;
    }
    ++pn;
  };
  auto step = [&](const FsTaps<NROWS>& grp, int n, int slot, int next_slot) {
#if PD_FS_SHRING
    const float sd = __int_as_float(sh_sd[slot]);
    const int kk = sh_kk[slot];
#else
    const int2 sh = shift[n];
    const float sd = __int_as_float(__builtin_amdgcn_readfirstlane(sh.x));
    const int kk = __builtin_amdgcn_readfirstlane(sh.y);
#endif
    const int k = kk >> 1;
    const int c0 = seg * kFsSeg + k;   // source column of the segment's first left tap (wave-uniform); lane i loads c0 + 2i ..
    // 12-byte form: regular plane, and no lane's load starts at column -3, -2 or -1: a load that starts left of the row reads
    // as zeros as a whole although its last columns may be inside, and the dword at byte offset -4 passes the 32-bit range
    // check (offset + 4 wraps to 0).  Loads that start at column <= -4 are dropped cleanly, those at >= 0 are exact.
    const bool general = (kk & 1) || (c0 < 0 && c0 + 2 * (kWave - 1) >= -3);
    if (general) {
      fs_general_plane<MIX, NROWS, RENDER>(a, r, col, n, sd, xt0f, HW, Wm1, rcpWm1, t, ea, automask, acc, rs, grp.dist);
      return;
    }
    float4 cv0, cv1, cv2;
    {
      const int cell = min(max(xt0 + k, -kFsGuard), W + 1) + kFsGuard;
      if (PD_FS_ABL & 2) {   // timing only: no LDS colour reads
        cv0 = make_float4(xt0f * 1e-3f, 0.25f, 0.5f, 0.0f); cv1 = make_float4(0.75f, xt0f * 1e-3f, 0.5f, 0.0f); cv2 = make_float4(0.1f, 0.2f, xt0f * 1e-3f, 0.0f);
      } else {
        cv0 = col[cell]; cv1 = col[cell + 1]; cv2 = col[cell + 2];
      }
    }
    const float kf = (float)k;
    const float xs0 = xt0f + kf, xs1 = xs0 + 1.0f, xs2 = xs1 + 1.0f;   // integers below 2^24: exact in any order
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float xtf = xt0f + (float)i;
      const float xsf = (i == 0) ? xs0 : xs1;
      const float ix = (PD_FS_ABL & 32) ? xsf + (sd - kf) : stream_ix(xtf, sd, Wm1, rcpWm1);
      const float w1 = ix - xsf, w0 = ((i == 0) ? xs1 : xs2) - ix;   // torch's (ix - x0), (x1 - ix) with x0 = xt + k
      float l, s = 0.0f;
      if (PD_FS_ABL & 128) {   // timing only: the loaded values are consumed, not interpolated
        l = grp.l[0][i] + grp.l[0][i + 1] + grp.l[NROWS - 1][i] * r.wB;
        if (MIX) s = grp.s[0][i] + grp.s[0][i + 1] + grp.s[NROWS - 1][i + 1] * r.wB;
      } else if (NROWS == 1) {
        l = grp.l[0][i] * w0 + grp.l[0][i + 1] * w1;
        if (MIX) s = grp.s[0][i] * w0 + grp.s[0][i + 1] * w1;
      } else {
        const float a0 = w0 * r.wA, a1 = w1 * r.wA, b0 = w0 * r.wB, b1 = w1 * r.wB;
        l = grp.l[0][i] * a0 + grp.l[0][i + 1] * a1 + grp.l[NROWS - 1][i] * b0 + grp.l[NROWS - 1][i + 1] * b1;
        if (MIX) s = grp.s[0][i] * a0 + grp.s[0][i + 1] * a1 + grp.s[NROWS - 1][i] * b0 + grp.s[NROWS - 1][i + 1] * b1;
      }
      const float4 ca = (i == 0) ? cv0 : cv1, cb = (i == 0) ? cv1 : cv2;
      const float c0v = ca.x * w0 + cb.x * w1, c1v = ca.y * w0 + cb.y * w1, c2v = ca.z * w0 + cb.z * w1;
      if (PD_FS_ABL & 4) {   // timing only: no softmax / mixture arithmetic
        acc[i].Z += l; acc[i].S += s; acc[i].C0 += c0v; acc[i].C1 += c1v; acc[i].C2 += c2v; acc[i].Mx += w0; acc[i].m = 0.0f;
      } else if (PD_FS_FIXREF && !RENDER) {
        const float d = l * kLog2e - acc[i].m;   // (m = -inf until a plane set it: d = +inf trips the limit below)
        asm("v_max_f32 %0, %1, %2" : "=v"(dmax[i]) : "v"(dmax[i]), "v"(d));   // (fmaxf adds a canonicalising v_max of the running value)
        mixture_accumulate<MIX>(acc[i], exp2_fast(d), s, c0v, c1v, c2v, t[i], t[2 + i], t[4 + i], ea[i], automask);
      } else
      fs_accumulate<MIX, RENDER>(acc[i], rs[i], l, s, c0v, c1v, c2v, t[i], t[2 + i], t[4 + i], ea[i], automask, grp.dist[i], n == N - 1);
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j) prefetch(g[j], j);
  if (PD_FS_FIXREF && !RENDER) {   // the reference: plane 0's scaled logit at the lane's two pixels (when plane 0 takes the 12-byte form;
    const int2 sh0 = shift[0];   // otherwise the general path's rescaling accumulator sets it when it reduces that plane)
    const float sd = __int_as_float(__builtin_amdgcn_readfirstlane(sh0.x));
    const int kk = __builtin_amdgcn_readfirstlane(sh0.y), k = kk >> 1, c0 = seg * kFsSeg + k;
    if (!((kk & 1) || (c0 < 0 && c0 + 2 * (kWave - 1) >= -3))) {
      const float kf = (float)k;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float xtf = xt0f + (float)i, xsf = xtf + kf;
        const float ix = stream_ix(xtf, sd, Wm1, rcpWm1);
        const float w1 = ix - xsf, w0 = (xsf + 1.0f) - ix;
        float l;
        if (NROWS == 1) l = g[0].l[0][i] * w0 + g[0].l[0][i + 1] * w1;
        else {
          const float a0 = w0 * r.wA, a1 = w1 * r.wA, b0 = w0 * r.wB, b1 = w1 * r.wB;
          l = g[0].l[0][i] * a0 + g[0].l[0][i + 1] * a1 + g[0].l[NROWS - 1][i] * b0 + g[0].l[NROWS - 1][i + 1] * b1;
        }
        acc[i].m = l * kLog2e;
      }
    }
  }
  int n = 0;
  for (; n + (D + 1) <= N; n += D + 1) {
#if PD_FS_PRIO == 1   // every wave of a SIMD leads for a quarter of the planes
    switch (((threadIdx.x >> 8) + (n / ((D + 1) * PD_FS_PRIO_PERIOD))) & 3) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
#endif
#pragma unroll
    for (int j = 0; j <= D; ++j) {
      prefetch(g[(j + D) % (D + 1)], (j + D) % (D + 1));
      step(g[j], n + j, j, (j + 1) % (D + 1));
    }
  }
#pragma unroll
  for (int j = 0; j <= D; ++j) {
    if (n + j < N) {
      prefetch(g[(j + D) % (D + 1)], (j + D) % (D + 1));
      step(g[j], n + j, j, (j + 1) % (D + 1));
    }
  }
  if (PD_FS_FIXREF && !RENDER) {
    const bool beyond = dmax[0] > kFixRefLimit || dmax[1] > kFixRefLimit;
    if (__builtin_amdgcn_ballot_w64(beyond) != 0) {   // rare: this wave again, every plane through the rescaling accumulator
      acc[0] = FwdAcc(); acc[1] = FwdAcc();
      for (int q = 0; q < N; ++q)
        fs_general_plane<MIX, NROWS, RENDER>(a, r, col, q, __int_as_float(__builtin_amdgcn_readfirstlane(shift[q].x)), xt0f, HW, Wm1,
                                             rcpWm1, t, ea, automask, acc, rs, g[0].dist);
    }
  }
#if PD_FS_PRIO == 1
  __builtin_amdgcn_s_setprio(0);   // (the rotating priority of the plane loop ends with it)
#endif
  fs_stamp(2);
  if (!live) return 0.0f;
  // ---- finish the two pixels: outputs + the backward's stash, 8-byte stores -------------------------------------------
  const FwdResult r0 = fwd_finish<MIX>(acc[0], t[0], t[2], t[4], ea[0], automask, !RENDER);   // (compositing weights are used as they are)
  const FwdResult r1 = fwd_finish<MIX>(acc[1], t[1], t[3], t[5], ea[1], automask, !RENDER);
  if ((PD_FS_ABL & 8) && r0.ph != 123.456f) return r0.ph + r1.ph;   // timing only: no output / stash stores
  float* st = stash + (long)r.b * a.stash_k * HW + pix;
  *reinterpret_cast<float2*>(st) = make_float2(r0.lse2, r1.lse2);
  *reinterpret_cast<float2*>(st + HW) = make_float2(r0.Sn, r1.Sn);
  *reinterpret_cast<float2*>(st + 2 * HW) = make_float2(r0.mx, r1.mx);
  *reinterpret_cast<float2*>(st + 3 * HW) = make_float2(r0.sel, r1.sel);
  float* rg = rgb_rec + (long)r.b * 3 * HW + pix;
  *reinterpret_cast<float2*>(rg) = make_float2(r0.r0, r1.r0);
  *reinterpret_cast<float2*>(rg + HW) = make_float2(r0.r1, r1.r1);
  *reinterpret_cast<float2*>(rg + 2 * HW) = make_float2(r0.r2, r1.r2);
  *reinterpret_cast<float2*>(ph_map + (long)r.b * HW + pix) = make_float2(r0.ph, r1.ph);
  return r0.ph + r1.ph;
}

// Team barrier of the `nwaves` waves that serve one row slot (a workgroup-wide s_barrier would make every wave wait for the
// slowest wave of the slowest row): an LDS counter per slot that every wave bumps once per round after its share of the
// staging, and polls until the whole team has.  LDS operations of a wave retire in order, so the bump follows the wave's
// staging stores; the fences keep the compiler from moving LDS accesses across.
// INVARIANT: every wave of a team runs the round loop of fwdstream_kernel the same number of times and reaches this barrier
// in every round — the loop's only exits (`pos >= T`) are workgroup-uniform.  A per-wave early exit or `continue` before the
// barrier would leave the team's other waves polling for ever, with no diagnostic (PD_FS_ROUNDS=1 builds, one item per
// workgroup and a plain __syncthreads, are the fallback).
__device__ __forceinline__ void fs_team_barrier(int* cnt, int target) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & (kWave - 1)) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The order in which an image's row groups are dealt to the (persistent) workgroups, by value in the kernel arguments: n = 0
// means "as they come".
constexpr int kFsOrderMax = 1024;
struct FsOrder { int n; unsigned short it[kFsOrderMax]; };
// Which target rows make up a row group, by value in the kernel arguments as well: n = 0 means group g serves rows g * rows ..
// g * rows + rows - 1; otherwise slot s of group g serves row y[g * rows + s] (>= H: none).  The host's regrouping
// (fwdstream_rows) puts a row that blends two source rows into the group of the neighbour whose main row that second row is.
constexpr int kFsRowsMax = 640;
struct FsRows { int n; unsigned short y[kFsRowsMax]; };
static_assert(sizeof(SweepArgs) + 3 * sizeof(float*) + 4 * sizeof(int) + sizeof(FsOrder) + sizeof(FsRows) <= 4096,
              "the forward's kernel arguments must fit the 4 KB kernarg segment");

// A workgroup serves `rows` consecutive target rows x one of `cblocks` column blocks of `segs` segments each (rows wider than
// 640 pixels are cut into column blocks so that three rows still fit the 16 waves of a workgroup) — one "item".  With
// `rounds` > 1 it is persistent: it walks items blk, blk + nblk, blk + 2 nblk, ... of the launch's item list (row-major over
// the images: the heavy rows of every image first), and the teams of its row slots move from round to round on their own:
// the next item's rows are staged into the slot's second LDS buffer by the team itself as soon as IT is done (the other
// teams keep computing), and only the team meets (fs_team_barrier).  Measured on the per-wave timeline
// (profiles/r05_fwd_ladder.md): with one 15-wave workgroup per CU and a workgroup barrier per item, the staging, the
// dispatch turn-around and the 40 % spread between the first and the last wave of a workgroup idle the CU between items.
template <bool MIX, bool AUTO, bool RENDER>
__global__ __launch_bounds__(kFsThreadsMax) void fwdstream_kernel(SweepArgs a, float* __restrict__ rgb_rec,
                                                                            float* __restrict__ ph_map,
                                                                            float* __restrict__ stash, int rows, int cblocks,
                                                                            int rounds, int nbk, FsOrder order, FsRows rowtab) {
  extern __shared__ float4 lds4[];
  // LDS per row slot: (colour row float4[W + 8] | shift int2[N] (padded to 16 bytes)) x (rounds > 1 ? 2 buffers : 1); then the
  // wave totals of ph_map and the teams' counters
  const int nseg = (a.W + kFsSeg - 1) / kFsSeg, segs = (nseg + cblocks - 1) / cblocks;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = wave / segs;                                  // which of the workgroup's rows
  const int CWk = a.W + 2 * kFsGuard;
  const int row_f4 = CWk + (a.N + 1) / 2;   // float4 per row buffer
  const int nbuf = (!RENDER && rounds > 1) ? 2 : 1;
  float* parts = reinterpret_cast<float*>(lds4 + rows * nbuf * row_f4);
  int* team = reinterpret_cast<int*>(parts + kFsThreadsMax / kWave);
  const int groups = (a.H + rows - 1) / rows;                    // row groups per image
  // An ITEM is (row group, image), numbered group * B + image.  block -> (column block cb, position k among the nbk blocks of
  // a column block); round r of that block serves item order.it[r * nbk + k] (the host's balanced deal, fwdstream_order: heavy
  // row groups first, the rounds snaking so that a block's items add up alike) or, without a table, item r * nbk + k.
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  const int cb = blk % cblocks, k = blk / cblocks;
  const int T = groups * a.B;
  const int tix = threadIdx.x - slot * segs * kWave, nthr = segs * kWave;
  if (!RENDER && rounds > 1) {
    if (threadIdx.x < rows) team[threadIdx.x] = 0;
    __syncthreads();
  }
  float ph_sum = 0.0f;
  const int nrounds = RENDER ? 1 : rounds;   // (the compositing kernels keep one item per workgroup: their plane loop holds more state,
                                             // and the loop around it costs them 40-60 spilled registers)
  for (int r = 0; r < nrounds; ++r) {
    const int pos = r * nbk + k;
    if (pos >= T) break;                                         // (workgroup-uniform)
    const int item = order.n ? (int)order.it[pos] : pos;
    const int b = item % a.B, gsel = item / a.B;
    const int grp = PD_FS_REVERSE ? groups - 1 - gsel : gsel;
    const int seg = cb * segs + (wave - slot * segs);            // which segment of the row
    const int y = rowtab.n ? (int)rowtab.y[grp * rows + slot] : grp * rows + slot;
    const bool active = y < a.H && seg < nseg;
    const RowSel row = two_row_form(make_row_sel(y < a.H ? y : 0, a.H), a.row_eps);
    float4* col = lds4 + (slot * nbuf + (r & 1)) * row_f4;
    int2* shift = reinterpret_cast<int2*>(col + CWk);
    fs_stamp(0);
    fs_stamp_ids(b, y);
    fs_stage_row(a, row, b, y, tix, nthr, col, shift);
    // rounds == 1: the kernel's only barrier before the outputs, every wave reaches it whatever its row needs; persistent: the
    // team's own (the buffer written here was last read two rounds ago, and every wave of the team has passed the barrier of
    // the round in between since)
    if (!RENDER && rounds > 1) fs_team_barrier(team + slot, segs * (r + 1));
    else __syncthreads();
    fs_stamp(1);
    if (!active) {}
    else if (row.nrows == 2 && !(PD_FS_ABL & 16)) ph_sum += fwdstream_body<MIX, AUTO, 2, RENDER>(a, row, b, y, seg, col, shift, rgb_rec, ph_map, stash);
    else                                          ph_sum += fwdstream_body<MIX, AUTO, 1, RENDER>(a, row, b, y, seg, col, shift, rgb_rec, ph_map, stash);
    fs_stamp(3);
  }
  if (a.ph_mean) {  // fused `.mean()` of trainer.py:742: wave totals -> LDS -> ONE atomic per workgroup
    const float v = wave_sum_hi(ph_sum);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) parts[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tsum = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tsum += parts[w];
      unsafeAtomicAdd(a.ph_mean, tsum * a.inv_numel);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// Workgroup shape: rows of up to 5 segments (640 pixels) go whole, wider ones in column blocks of equal size; PD_FS_ROWS rows
// per workgroup where their waves fit its 16, else what fits.
#ifndef PD_FS_BALANCE
#define PD_FS_BALANCE 1   // 0: the persistent workgroups take the items in launch order (A/B)
#endif
#ifndef PD_FS_ROUNDS
#define PD_FS_ROUNDS 0   // items per (persistent) workgroup: 0 = as many as give every CU one workgroup, 1 = one item per workgroup
#endif
struct FsShape { int rows, cblocks, segs, rounds, nbk; size_t lds; };
static int device_cu_count() {
  static std::atomic<int> cus[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  if (dev < 0 || dev >= kMaxDevices) dev = kMaxDevices - 1;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  (void)hipGetLastError();
  cus[dev].store(n, std::memory_order_relaxed);
  return n;
}
static FsShape fwdstream_shape(const pd_sweep_desc* d) {
  FsShape s;
  const int nseg = ceil_div(d->W, kFsSeg);
  s.cblocks = ceil_div(nseg, 5);
  s.segs = ceil_div(nseg, s.cblocks);
  const int most = (kFsThreadsMax / kWave) / s.segs;
  s.rows = most < PD_FS_ROWS ? most : PD_FS_ROWS;
  if (s.rows > d->H) s.rows = d->H;
  // persistent rounds: one workgroup of (nearly) 16 waves per CU is all that fits, so give every CU ONE workgroup and let it
  // walk its share of the items; smaller workgroups (several resident per CU) keep the dispatcher's one item per workgroup
  const int items = ceil_div(d->H, s.rows) * s.cblocks * d->B;
  const int waves = s.rows * s.segs;
  s.rounds = PD_FS_ROUNDS ? PD_FS_ROUNDS : (2 * waves > kFsThreadsMax / kWave ? ceil_div(items, device_cu_count()) : 1);
  if (d->flags & PD_RENDER_PROB) s.rounds = 1;
  if (s.rounds < 1) s.rounds = 1;
  if (s.rounds > 8) s.rounds = 8;
  s.nbk = ceil_div(ceil_div(d->H, s.rows) * d->B, s.rounds);   // blocks per column block
  const size_t row_f4 = (size_t)d->W + 2 * kFsGuard + ((size_t)d->N + 1) / 2;
  s.lds = s.rows * (s.rounds > 1 ? 2 : 1) * row_f4 * sizeof(float4) + (size_t)(kFsThreadsMax / kWave) * sizeof(float) +
          (size_t)(kFsThreadsMax / kWave) * sizeof(int) + PD_FS_LDS_PAD;
  if (s.rounds > 1 && s.lds > device_lds_bytes()) {   // the second row buffers do not fit: one item per workgroup
    s.rounds = 1;
    s.nbk = ceil_div(d->H, s.rows) * d->B;
    s.lds = s.rows * row_f4 * sizeof(float4) + (size_t)(kFsThreadsMax / kWave) * (sizeof(float) + sizeof(int)) + PD_FS_LDS_PAD;
  }
  return s;
}

// Does target row y blend two source rows (the row bodies' "heavy" rows: twice the tap loads)?  make_row_sel + two_row_form
// on the host, operation by operation in fp32.  Only the DEAL depends on it (the kernel decides every row's footprint itself),
// so a disagreement with the device in some last bit would cost balance, not correctness.
static bool host_row_is_heavy(int y, int H, float row_eps, int* partner = nullptr) {
  if (partner) *partner = -1;
  const HostRowSel r = host_row_sel(y, H);   // (pd_rowgeom.h)
  if (r.nrows == 2) {
    const bool two = !(row_eps > 0.0f && fminf(r.wA, r.wB) < row_eps);
    if (two && partner) *partner = (r.yA == y) ? r.yB : r.yA;   // the OTHER source row: the main row of that target row
    return two;
  }
  return r.nrows == 1 && (r.wA != 1.0f || r.wy_main != 1.0f);
}

// Row groups that keep a heavy row and the neighbour it blends in together.  A team reads its row's second source row from global
// memory; where the team of that row sits in the same workgroup the lines are in the CU's L1 / the XCD's L2 (the teams walk the
// planes within a few planes of each other), where it sits in another workgroup they mostly are not: with consecutive rows
// 3g .. 3g + 2, 15 of the 48 heavy rows of H = 192 have their partner in the next group (NOTEBOOK 10.7: 30 MB of the forward's
// 41 MB of excess reads).  Groups need not be consecutive rows — each team has its own row buffer — so: chains of linked rows are
// cut into pieces of at most `rows`, the pieces go first-fit (longest first) into the G groups, single rows fill the holes in order.
// H = 192, rows = 3: two links left cut (the chains 24-28 and 30-33) instead of fifteen.
#ifndef PD_FS_REGROUP
#define PD_FS_REGROUP 1   // 0: consecutive rows (A/B)
#endif
static FsRows fwdstream_rows(int H, int R, float row_eps) {
  struct Cache { int H = 0, R = 0; float eps = -1.0f; FsRows t; };
  static thread_local Cache c;   // (the table depends on the height, the rows per group and the threshold only)
  const int G = ceil_div(H, R);
  if (c.H == H && c.R == R && c.eps == row_eps) return c.t;
  c.H = H; c.R = R; c.eps = row_eps;
  FsRows& t = c.t;
  t.n = 0;
  if (!PD_FS_REGROUP || R < 2 || G * R > kFsRowsMax) return t;
  static thread_local unsigned char link[kFsRowsMax], placed[kFsRowsMax], fill[kFsRowsMax];
  int nlinks = 0;
  for (int y = 0; y < H; ++y) link[y] = placed[y] = 0;
  for (int y = 0; y < H; ++y) {
    int p;
    if (host_row_is_heavy(y, H, row_eps, &p) && p >= 0 && p < H && (p == y + 1 || p == y - 1)) { link[p < y ? p : y] = 1; ++nlinks; }
  }
  if (!nlinks) return t;
  for (int g = 0; g < G; ++g) fill[g] = 0;
  for (int i = 0; i < G * R; ++i) t.y[i] = 0xFFFF;
  auto cap = [&](int g) { return g == G - 1 ? H - (G - 1) * R : R; };
  for (int len = R; len >= 2; --len) {          // pieces of a chain, longest first; first fit
    int y = 0;
    while (y < H) {
      int n = 1;
      while (y + n < H && n < R && link[y + n - 1]) ++n;   // (the chain is cut after R rows)
      if (n == len && !placed[y]) {
        for (int g = 0; g < G; ++g)
          if (fill[g] + n <= cap(g)) {
            for (int k = 0; k < n; ++k) { t.y[g * R + fill[g] + k] = (unsigned short)(y + k); placed[y + k] = 1; }
            fill[g] += n;
            break;
          }
        // (no group with room left: the rows stay single and fill holes below)
      }
      y += n;
    }
  }
  int g = 0;
  for (int y = 0; y < H; ++y) {                  // everything else, in order, into the holes
    if (placed[y]) continue;
    while (g < G && fill[g] >= cap(g)) ++g;
    if (g == G) { t.n = 0; return t; }           // (cannot happen: the capacities add up to H)
    t.y[g * R + fill[g]++] = (unsigned short)y;
  }
  t.n = G * R;
  return t;
}

// The persistent workgroups' deal: items (row group x image) sorted by the number of heavy rows in the group (heavy first;
// ties in launch order, images fastest), chunked into `rounds` chunks of nbk, every other chunk reversed — block k then serves
// the k-th heaviest item of the first chunk, the k-th LIGHTEST of the second, and so on (profiles/r05_fwd_ladder.md: with the
// dispatcher's order the CUs' totals differ by 13 % on the exact-rows forward).
static FsOrder fwdstream_order(const pd_sweep_desc* d, const FsShape& sh, float row_eps, const FsRows& rowtab) {
  FsOrder o;
  o.n = 0;
  const int groups = ceil_div(d->H, sh.rows), T = groups * d->B;
  if (!PD_FS_BALANCE || sh.rounds < 2 || T > kFsOrderMax) return o;
  int weight[kFsOrderMax], sorted[kFsOrderMax];
  for (int g = 0; g < groups; ++g) {
    int w = 0;
    for (int r = 0; r < sh.rows; ++r) {
      const int y = rowtab.n ? (int)rowtab.y[g * sh.rows + r] : g * sh.rows + r;
      if (y < d->H) w += host_row_is_heavy(y, d->H, row_eps) ? 1 : 0;
    }
    weight[g] = w;
  }
  int n = 0;   // counting sort by weight, descending; stable in (group, image).  (Light groups first with a top-down backward,
               // so that each kernel starts on what the other read last AND the backward starts with its heavy rows: measured
               // slower, forward 0.0954 / backward 0.1766 against 0.0914 / 0.1752 ms)
  for (int w = sh.rows; w >= 0; --w)
    for (int g = 0; g < groups; ++g)
      if (weight[g] == w)
        for (int b = 0; b < d->B; ++b) sorted[n++] = g * d->B + b;
  for (int r = 0; r < sh.rounds; ++r)
    for (int k = 0; k < sh.nbk; ++k) {
      const int pos = r * sh.nbk + k;
      if (pos >= T) break;
      const int lo = r * sh.nbk, hi = (lo + sh.nbk < T ? lo + sh.nbk : T) - 1;   // this round's chunk of the sorted list
      o.it[pos] = (unsigned short)sorted[(r & 1) ? (hi - k >= lo ? hi - k : lo + k) : pos];
    }
  o.n = T;
  return o;
}

bool fwdstream_applicable(const pd_sweep_desc* d, const SweepArgs& a) {
  if (!rowshift_applicable(d) || a.has_mask || !switches().fwd_stream) return false;
  if ((d->flags & PD_RENDER_PROB) && (((long)d->H * d->W) % 2 != 0 || (reinterpret_cast<uintptr_t>(a.dists) & 7))) return false;
  // pixel pairs: even width, 8-byte aligned rows of the per-pixel tensors (their bases come 8-byte aligned from any allocator
  // that hands out float2-aligned memory; checked because the boundary takes raw pointers)
  if (d->W % 2 != 0) return false;
  return fwdstream_shape(d).lds <= device_lds_bytes();
}

int fwdstream_fwd(const pd_sweep_desc* d, const SweepArgs& a, float* rgb_rec, float* ph_map, float* stash, hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(a.tgt) | reinterpret_cast<uintptr_t>(a.src) | reinterpret_cast<uintptr_t>(rgb_rec) |
       reinterpret_cast<uintptr_t>(ph_map) | reinterpret_cast<uintptr_t>(stash)) & 7)
    return rowshift_fwd(d, a, rgb_rec, ph_map, stash, stream);   // unaligned tensors: the one-pixel-per-lane forward
  const FsShape sh = fwdstream_shape(d);
  const dim3 grid(sh.nbk * sh.cblocks, 1), block(sh.segs * sh.rows * kWave);
  const FsRows rowtab = fwdstream_rows(d->H, sh.rows, a.row_eps);
  const FsOrder order = fwdstream_order(d, sh, a.row_eps, rowtab);
  const size_t shmem = sh.lds;
  const bool mix = (d->flags & PD_MIXTURE) != 0, am = (d->flags & PD_AUTOMASK) != 0, render = (d->flags & PD_RENDER_PROB) != 0;
#define PD_FS_LAUNCH(M, A, R)                                                                                              \
  do {                                                                                                                    \
    static LdsGrant granted;                                                                                              \
    if (int rc = grant_dynamic_lds((const void*)fwdstream_kernel<M, A, R>, shmem, &granted, "fwdstream_kernel")) return rc; \
    fwdstream_kernel<M, A, R><<<grid, block, shmem, stream>>>(a, rgb_rec, ph_map, stash, sh.rows, sh.cblocks, sh.rounds, sh.nbk, order, rowtab); \
  } while (0)
  if (render) {
    if (mix) { if (am) PD_FS_LAUNCH(true, true, true); else PD_FS_LAUNCH(true, false, true); }
    else PD_FS_LAUNCH(false, false, true);
  } else {
    if (mix) { if (am) PD_FS_LAUNCH(true, true, false); else PD_FS_LAUNCH(true, false, false); }
    else PD_FS_LAUNCH(false, false, false);
  }
#undef PD_FS_LAUNCH
  return check_launch("fwdstream_kernel");
}

}  // namespace pd

// Host-only diagnostics (not declared in include/planedepth_hip.h; tests/test_row_groups.py): the forward's row groups for a
// height and `rows` rows per group -> out[G * rows] (entries >= H: empty slots), *links / *cut = linked row pairs in all / those
// whose rows ended up in different groups.  Returns G * rows, 0 where the consecutive grouping is kept, -1 on bad arguments.
extern "C" int pd_debug_fwd_row_groups(int H, int rows, float row_eps, unsigned short* out, int* links, int* cut) {
  if (H < 1 || rows < 1 || !out) return -1;
  const pd::FsRows t = pd::fwdstream_rows(H, rows, row_eps);
  const int G = (H + rows - 1) / rows;
  for (int i = 0; i < G * rows && i < pd::kFsRowsMax; ++i) out[i] = t.n ? t.y[i] : (unsigned short)(i < H ? i : 0xFFFF);
  int nl = 0, nc = 0;
  for (int y = 0; y < H; ++y) {
    int p;
    if (!pd::host_row_is_heavy(y, H, row_eps, &p) || p < 0 || p >= H) continue;
    ++nl;
    int gy = -1, gp = -1;
    for (int i = 0; i < G * rows && i < pd::kFsRowsMax; ++i) { if (out[i] == y) gy = i / rows; if (out[i] == p) gp = i / rows; }
    if (gy != gp) ++nc;
  }
  if (links) *links = nl;
  if (cut) *cut = nc;
  return t.n;
}

#if PD_FS_TRACE
// diagnostics build only (not declared in include/planedepth_hip.h): copies the stamps of the last launch to the host
extern "C" int pd_debug_fs_trace(unsigned long long* host_dst, long words) {
  const long have = (long)pd::kFsTraceWgs * 16 * pd::kFsTraceWords;
  if (hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(pd::g_fs_trace), sizeof(unsigned long long) * (words < have ? words : have)) != hipSuccess) return 1;
  return 0;
}
#endif
