// Fused tail of DepthDecoder.forward (reference networks/depth_decoder.py:256-260, 274-291, softmax branch; SURVEY.md
// §8f rank 1): everything the decoder does with the outputs of dispconv / sigmaconv, in ONE pass over the planes:
//   logits = raw_logits * padding_mask;  pi = softmax_N(logits);  sigma = clamp(sigmoid(raw_sigma), .01, 1)
//   probability = (pi / sigma * mask) / sum_N(...)   [mixture]   |   probability = pi   [no mixture]
//   disp = sum_N probability * disp_layered;  depth = 0.1 * 0.58 * W / disp
// The reference runs ~12 full-tensor ATen passes for this.  Here one thread owns one pixel and streams its N planes
// once: the softmax normaliser cancels in `probability`, so disp is a ratio of two running sums and neither pi nor
// probability has to exist in memory.  Training consumes logits, sigma, disp and depth only (SURVEY.md §8b B1:
// outputs["probability"] is read for its shape); pi / probability are produced on demand by pd_decoder_tail_layers
// from the per-pixel stash {log-sum-exp, sum(pi*mask/sigma)}.
// Algorithmic bytes per pixel: forward reads 2N (+N mask) floats, writes N (sigma) (+N logits when there is a mask)
// + 4; backward reads 4N (+N) and writes 2N.  HBM-bound streaming, no reuse.
#include "pd_tail_common.h"

namespace pd {

struct TailArgs {
  int N, HW, W;
  int mix, dense;
  const float* raw_logits;
  const float* raw_sigma;
  const float* mask;   // may be NULL (all ones)
  const float* dl;     // [B,N] or [B,N,H,W]
};

template <bool MIX, bool HASMASK, int PX>
__global__ __launch_bounds__(kBlock) void tail_fwd_kernel(TailArgs a, float* __restrict__ logits, float* __restrict__ sigma,
                                                          float* __restrict__ disp, float* __restrict__ depth,
                                                          float* __restrict__ stash) {
  const int pix = (blockIdx.x * kBlock + threadIdx.x) * PX, b = blockIdx.y;
  if (pix >= a.HW) return;
  const long base = (long)b * a.N * a.HW + pix;
  float m[PX], Z[PX], Sw[PX], Sd[PX];  // running reference, sum e^(l-m), sum of weights, sum w*d
#pragma unroll
  for (int j = 0; j < PX; ++j) { m[j] = -INFINITY; Z[j] = Sw[j] = Sd[j] = 0.0f; }
#pragma unroll 2
  for (int n = 0; n < a.N; ++n) {
    const long i = base + (long)n * a.HW;
    const Px<PX> mk = HASMASK ? ldv<PX>(a.mask + i) : splat<PX>(1.0f);
    const Px<PX> rl = ldv<PX>(a.raw_logits + i);
    const Px<PX> rs = MIX ? ldv<PX>(a.raw_sigma + i) : splat<PX>(0.0f);
    const Px<PX> dv = a.dense ? ldv<PX>(a.dl + i) : splat<PX>(a.dl[b * a.N + n]);
    Px<PX> lo, so;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      const float l = rl.v[j] * mk.v[j];                           // depth_decoder.py:259
      lo.v[j] = l;
      float inv = 1.0f;
      if (MIX) {
        const float sg = clamp_sigma(sigmoid_f(rs.v[j]));          // :278-279
        so.v[j] = sg;
        inv = mk.v[j] / sg;                                        // :282-283 (mask applied to the weights)
      }
      if (l > m[j]) {  // move the reference to the new maximum
        const float sc = __expf(m[j] - l);
        Z[j] *= sc; Sw[j] *= sc; Sd[j] *= sc;
        m[j] = l;
      }
      const float e = __expf(l - m[j]);
      const float w = e * inv;
      Z[j] += e;
      Sw[j] += w;
      Sd[j] += w * dv.v[j];
    }
    if (HASMASK) stv<PX>(logits + i, lo);
    if (MIX) stv<PX>(sigma + i, so);
  }
  Px<PX> o_disp, o_depth, o_lse, o_sn;
#pragma unroll
  for (int j = 0; j < PX; ++j) {
    const float dsp = Sd[j] / Sw[j];                                // :284-285, 289 (the softmax normaliser cancels)
    o_disp.v[j] = dsp;
    o_depth.v[j] = 0.1f * 0.58f * (float)a.W / dsp;                 // :291
    o_lse.v[j] = m[j] + __logf(Z[j]);                               // log-sum-exp of the masked logits
    o_sn.v[j] = Sw[j] / Z[j];                                       // sum_N pi * mask / sigma
  }
  stv<PX>(disp + (long)b * a.HW + pix, o_disp);
  stv<PX>(depth + (long)b * a.HW + pix, o_depth);
  stv<PX>(stash + ((long)b * 2 + 0) * a.HW + pix, o_lse);
  stv<PX>(stash + ((long)b * 2 + 1) * a.HW + pix, o_sn);
}

// pi and probability (depth_decoder.py:275, 281-285) for callers that want the tensors.
template <bool MIX, bool HASMASK, int PX>
__global__ __launch_bounds__(kBlock) void tail_layers_kernel(TailArgs a, const float* __restrict__ stash,
                                                             float* __restrict__ pi, float* __restrict__ prob) {
  const int pix = (blockIdx.x * kBlock + threadIdx.x) * PX, b = blockIdx.y;
  if (pix >= a.HW) return;
  const long base = (long)b * a.N * a.HW + pix;
  const Px<PX> lse = ldv<PX>(stash + ((long)b * 2 + 0) * a.HW + pix);
  const Px<PX> sn = ldv<PX>(stash + ((long)b * 2 + 1) * a.HW + pix);
#pragma unroll 2
  for (int n = 0; n < a.N; ++n) {
    const long i = base + (long)n * a.HW;
    const Px<PX> mk = HASMASK ? ldv<PX>(a.mask + i) : splat<PX>(1.0f);
    const Px<PX> rl = ldv<PX>(a.raw_logits + i);
    const Px<PX> rs = MIX ? ldv<PX>(a.raw_sigma + i) : splat<PX>(0.0f);
    Px<PX> op, oq;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      const float p = __expf(rl.v[j] * mk.v[j] - lse.v[j]);
      op.v[j] = p;
      oq.v[j] = MIX ? p * mk.v[j] / clamp_sigma(sigmoid_f(rs.v[j])) / sn.v[j] : p;
    }
    if (pi) stv<PX>(pi + i, op);
    if (prob) stv<PX>(prob + i, oq);
  }
}

// Backward.  With w_n = pi_n m_n / sigma_n, S = sum w, P_n = w_n / S, disp = sum P_n d_n and upstream gD = d loss/d disp
// (+ the depth term): d disp / d w_n = (d_n - disp) / S, and since sum_k pi_k (d loss / d pi_k) = gD/S * sum_k w_k
// (d_k - disp) = 0 exactly, the softmax backward needs no second reduction:
//   g_logits_n += gD (d_n - disp) P_n;   g_sigma_n -= gD (d_n - disp) P_n / sigma_n;   g_d_n = gD P_n.
template <bool MIX, bool HASMASK, int PX>
__global__ __launch_bounds__(kBlock) void tail_bwd_kernel(TailArgs a, const float* __restrict__ stash,
                                                          const float* __restrict__ disp,
                                                          const float* __restrict__ g_logits,
                                                          const float* __restrict__ g_sigma,
                                                          const float* __restrict__ g_disp,
                                                          const float* __restrict__ g_depth,
                                                          float* __restrict__ g_raw_logits,
                                                          float* __restrict__ g_raw_sigma, float* __restrict__ g_dl,
                                                          float* __restrict__ partials) {
  extern __shared__ float red[];  // [N] block sums of the per-plane disparity gradient
  const int pix = (blockIdx.x * kBlock + threadIdx.x) * PX, b = blockIdx.y;
  const bool reduce = (g_dl != nullptr) && !a.dense;
  if (reduce) {
    for (int i = threadIdx.x; i < a.N; i += kBlock) red[i] = 0.0f;
    __syncthreads();
  }
  const bool active = pix < a.HW;
  const long base = (long)b * a.N * a.HW + (active ? pix : 0);
  Px<PX> lse = splat<PX>(0.0f), sn = splat<PX>(1.0f), dsp = splat<PX>(1.0f), gD = splat<PX>(0.0f);
  if (active) {
    lse = ldv<PX>(stash + ((long)b * 2 + 0) * a.HW + pix);
    sn = ldv<PX>(stash + ((long)b * 2 + 1) * a.HW + pix);
    dsp = ldv<PX>(disp + (long)b * a.HW + pix);
    if (g_disp) gD = ldv<PX>(g_disp + (long)b * a.HW + pix);
    if (g_depth) {
      const Px<PX> gz = ldv<PX>(g_depth + (long)b * a.HW + pix);
#pragma unroll
      for (int j = 0; j < PX; ++j) gD.v[j] -= gz.v[j] * (0.1f * 0.58f * (float)a.W) / (dsp.v[j] * dsp.v[j]);
    }
  }
  const int lane = threadIdx.x & (kWave - 1);
  for (int n = 0; n < a.N; ++n) {
    const long i = base + (long)n * a.HW;
    float gd = 0.0f;
    if (active) {
      const Px<PX> mk = HASMASK ? ldv<PX>(a.mask + i) : splat<PX>(1.0f);
      const Px<PX> rl = ldv<PX>(a.raw_logits + i);
      const Px<PX> rs = MIX ? ldv<PX>(a.raw_sigma + i) : splat<PX>(0.0f);
      const Px<PX> dv = a.dense ? ldv<PX>(a.dl + i) : splat<PX>(a.dl[b * a.N + n]);
      const Px<PX> gl = g_logits ? ldv<PX>(g_logits + i) : splat<PX>(0.0f);
      const Px<PX> gs = (MIX && g_sigma) ? ldv<PX>(g_sigma + i) : splat<PX>(0.0f);
      Px<PX> o_l, o_s, o_d;
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        const float p = __expf(rl.v[j] * mk.v[j] - lse.v[j]);
        float sgu = 1.0f, sg = 1.0f, P = p;
        if (MIX) {
          sgu = sigmoid_f(rs.v[j]);
          sg = clamp_sigma(sgu);
          P = p * mk.v[j] / sg / sn.v[j];
        }
        const float t = gD.v[j] * (dv.v[j] - dsp.v[j]) * P;
        o_l.v[j] = (gl.v[j] + t) * mk.v[j];
        const float gsig = gs.v[j] - t / sg;
        o_s.v[j] = (sgu == sg) ? gsig * sgu * (1.0f - sgu) : 0.0f;   // clamp gate (inclusive bounds), sigmoid'
        o_d.v[j] = gD.v[j] * P;
        gd += o_d.v[j];
      }
      if (g_raw_logits) stv<PX>(g_raw_logits + i, o_l);
      if (MIX && g_raw_sigma) stv<PX>(g_raw_sigma + i, o_s);
      if (g_dl && a.dense) stv<PX>(g_dl + i, o_d);
    }
    if (reduce) {
      const float v = wave_sum_hi(gd);
      if (lane == kWave - 1) lds_add(&red[n], v);
    }
  }
  if (reduce) {
    __syncthreads();
    float* dst = partials + ((long)b * gridDim.x + blockIdx.x) * a.N;
    for (int i = threadIdx.x; i < a.N; i += kBlock) dst[i] = red[i];
  }
}

// partials [B][R][N] -> out [B][N]; one wave per (n, b)
__global__ void tail_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out, int R, int N) {
  const int n = blockIdx.x, b = blockIdx.y;
  const float* p = partials + (long)b * R * N + n;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < R; i += kWave) acc += p[(long)i * N];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * N + n] = acc;
}

static int tail_validate(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                         const float* dl) {
  PD_REQUIRE(B > 0 && B <= 65535 && N > 0 && H > 0 && W > 0, "bad shape");
  PD_REQUIRE((long)H * W < (1L << 31), "image too large");
  PD_REQUIRE((flags & ~(PD_TAIL_MIXTURE | PD_TAIL_DISP_DENSE)) == 0, "unknown flags");
  PD_REQUIRE(raw_logits && dl, "NULL pointer");
  PD_REQUIRE(!(flags & PD_TAIL_MIXTURE) || raw_sigma, "mixture needs raw_sigma");
  return 0;
}

static TailArgs tail_args(int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                          const float* mask, const float* dl) {
  TailArgs a;
  a.N = N; a.HW = H * W; a.W = W;
  a.mix = (flags & PD_TAIL_MIXTURE) != 0;
  a.dense = (flags & PD_TAIL_DISP_DENSE) != 0;
  a.raw_logits = raw_logits; a.raw_sigma = raw_sigma; a.mask = mask; a.dl = dl;
  return a;
}

#define PD_TAIL_DISPATCH_PX(KERNEL, PX, mix, hasmask, grid, shmem, stream, ...)                             \
  do {                                                                                                       \
    if (mix) {                                                                                               \
      if (hasmask) KERNEL<true, true, PX><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                     \
      else         KERNEL<true, false, PX><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                    \
    } else {                                                                                                 \
      if (hasmask) KERNEL<false, true, PX><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                    \
      else         KERNEL<false, false, PX><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                   \
    }                                                                                                        \
  } while (0)
#define PD_TAIL_DISPATCH(KERNEL, px, mix, hasmask, grid, shmem, stream, ...)                                 \
  do {                                                                                                       \
    if ((px) == 4) PD_TAIL_DISPATCH_PX(KERNEL, 4, mix, hasmask, grid, shmem, stream, __VA_ARGS__);           \
    else           PD_TAIL_DISPATCH_PX(KERNEL, 1, mix, hasmask, grid, shmem, stream, __VA_ARGS__);           \
  } while (0)

}  // namespace pd

using namespace pd;

extern "C" size_t pd_decoder_tail_bwd_workspace_floats(int B, int N, int H, int W) {
  return (size_t)B * ceil_div(H * W, kBlock) * N;
}

extern "C" int pd_decoder_tail_fwd(int B, int N, int H, int W, int flags, const float* raw_logits,
                                   const float* raw_sigma, const float* padding_mask, const float* disp_layered,
                                   float* logits, float* sigma, float* disp, float* depth, float* stash,
                                   pd_stream_t stream) {
  if (int rc = tail_validate(B, N, H, W, flags, raw_logits, raw_sigma, disp_layered)) return rc;
  PD_REQUIRE(disp && depth && stash, "NULL output");
  PD_REQUIRE(!(flags & PD_TAIL_MIXTURE) || sigma, "mixture needs the sigma output");
  PD_REQUIRE(!padding_mask || logits, "a padding mask needs the logits output");
  const TailArgs a = tail_args(N, H, W, flags, raw_logits, raw_sigma, padding_mask, disp_layered);
  const int px = tail_px(H, W, {raw_logits, raw_sigma, padding_mask, a.dense ? disp_layered : nullptr, logits, sigma,
                                disp, depth, stash});
  dim3 grid(ceil_div(ceil_div(H * W, px), kBlock), B);
  PD_TAIL_DISPATCH(tail_fwd_kernel, px, a.mix, padding_mask != nullptr, grid, 0, (hipStream_t)stream, a, logits, sigma,
                   disp, depth, stash);
  return check_launch("tail_fwd_kernel");
}

extern "C" int pd_decoder_tail_layers(int B, int N, int H, int W, int flags, const float* raw_logits,
                                      const float* raw_sigma, const float* padding_mask, const float* stash, float* pi,
                                      float* probability, pd_stream_t stream) {
  if (int rc = tail_validate(B, N, H, W, flags, raw_logits, raw_sigma, raw_logits)) return rc;
  PD_REQUIRE(stash && (pi || probability), "NULL pointer");
  const TailArgs a = tail_args(N, H, W, flags, raw_logits, raw_sigma, padding_mask, nullptr);
  const int px = tail_px(H, W, {raw_logits, raw_sigma, padding_mask, stash, pi, probability});
  dim3 grid(ceil_div(ceil_div(H * W, px), kBlock), B);
  PD_TAIL_DISPATCH(tail_layers_kernel, px, a.mix, padding_mask != nullptr, grid, 0, (hipStream_t)stream, a, stash, pi,
                   probability);
  return check_launch("tail_layers_kernel");
}

extern "C" int pd_decoder_tail_bwd(int B, int N, int H, int W, int flags, const float* raw_logits,
                                   const float* raw_sigma, const float* padding_mask, const float* disp_layered,
                                   const float* stash, const float* disp, const float* g_logits, const float* g_sigma,
                                   const float* g_disp, const float* g_depth, float* g_raw_logits, float* g_raw_sigma,
                                   float* g_disp_layered, float* workspace, pd_stream_t stream) {
  if (int rc = tail_validate(B, N, H, W, flags, raw_logits, raw_sigma, disp_layered)) return rc;
  PD_REQUIRE(stash && disp, "NULL pointer");
  PD_REQUIRE(g_raw_logits || g_raw_sigma || g_disp_layered, "no gradient requested");
  const TailArgs a = tail_args(N, H, W, flags, raw_logits, raw_sigma, padding_mask, disp_layered);
  const bool reduce = g_disp_layered && !a.dense;
  PD_REQUIRE(!reduce || workspace, "per-plane disparity gradient needs the workspace");
  PD_REQUIRE((size_t)N * sizeof(float) <= 64 * 1024, "too many planes");
  const int px = tail_px(H, W, {raw_logits, raw_sigma, padding_mask, a.dense ? disp_layered : nullptr, stash, disp,
                                g_logits, g_sigma, g_disp, g_depth, g_raw_logits, g_raw_sigma,
                                a.dense ? g_disp_layered : nullptr});
  dim3 grid(ceil_div(ceil_div(H * W, px), kBlock), B);
  const size_t shmem = reduce ? (size_t)N * sizeof(float) : 0;
  PD_TAIL_DISPATCH(tail_bwd_kernel, px, a.mix, padding_mask != nullptr, grid, shmem, (hipStream_t)stream, a, stash, disp,
                   g_logits, g_sigma, g_disp, g_depth, g_raw_logits, g_raw_sigma, g_disp_layered, workspace);
  if (int rc = check_launch("tail_bwd_kernel")) return rc;
  if (reduce) {
    tail_reduce_kernel<<<dim3(N, B), kWave, 0, (hipStream_t)stream>>>(workspace, g_disp_layered, (int)grid.x, N);
    return check_launch("tail_reduce_kernel");
  }
  return 0;
}
