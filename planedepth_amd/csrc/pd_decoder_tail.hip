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
#include "pd_common.h"

namespace pd {

constexpr float kTailSigmaMin = 0.01f, kTailSigmaMax = 1.0f;

struct TailArgs {
  int N, HW, W;
  int mix, dense;
  const float* raw_logits;
  const float* raw_sigma;
  const float* mask;   // may be NULL (all ones)
  const float* dl;     // [B,N] or [B,N,H,W]
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float clamp_sigma(float s) { return fminf(fmaxf(s, kTailSigmaMin), kTailSigmaMax); }

template <bool MIX, bool HASMASK>
__global__ __launch_bounds__(kBlock) void tail_fwd_kernel(TailArgs a, float* __restrict__ logits, float* __restrict__ sigma,
                                                          float* __restrict__ disp, float* __restrict__ depth,
                                                          float* __restrict__ stash) {
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= a.HW) return;
  const long base = (long)b * a.N * a.HW + pix;
  float m = -INFINITY, Z = 0.0f, Sw = 0.0f, Sd = 0.0f;  // running reference, sum e^(l-m), sum of weights, sum w*d
#pragma unroll 4
  for (int n = 0; n < a.N; ++n) {
    const long i = base + (long)n * a.HW;
    const float mk = HASMASK ? a.mask[i] : 1.0f;
    const float l = a.raw_logits[i] * mk;                        // depth_decoder.py:259
    if (HASMASK) logits[i] = l;
    float inv = 1.0f;
    if (MIX) {
      const float sg = clamp_sigma(sigmoid_f(a.raw_sigma[i]));   // :278-279
      sigma[i] = sg;
      inv = mk / sg;                                             // :282-283 (mask applied to the weights)
    }
    const float d = a.dense ? a.dl[i] : a.dl[b * a.N + n];
    if (l > m) {  // move the reference to the new maximum
      const float sc = __expf(m - l);
      Z *= sc; Sw *= sc; Sd *= sc;
      m = l;
    }
    const float e = __expf(l - m);
    const float w = e * inv;
    Z += e;
    Sw += w;
    Sd += w * d;
  }
  const float dsp = Sd / Sw;                                      // :284-285, 289 (the softmax normaliser cancels)
  disp[(long)b * a.HW + pix] = dsp;
  depth[(long)b * a.HW + pix] = 0.1f * 0.58f * (float)a.W / dsp;  // :291
  stash[((long)b * 2 + 0) * a.HW + pix] = m + __logf(Z);          // log-sum-exp of the masked logits
  stash[((long)b * 2 + 1) * a.HW + pix] = Sw / Z;                 // sum_N pi * mask / sigma
}

// pi and probability (depth_decoder.py:275, 281-285) for callers that want the tensors.
template <bool MIX, bool HASMASK>
__global__ __launch_bounds__(kBlock) void tail_layers_kernel(TailArgs a, const float* __restrict__ stash,
                                                             float* __restrict__ pi, float* __restrict__ prob) {
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  if (pix >= a.HW) return;
  const long base = (long)b * a.N * a.HW + pix;
  const float lse = stash[((long)b * 2 + 0) * a.HW + pix];
  const float invS = 1.0f / stash[((long)b * 2 + 1) * a.HW + pix];
#pragma unroll 4
  for (int n = 0; n < a.N; ++n) {
    const long i = base + (long)n * a.HW;
    const float mk = HASMASK ? a.mask[i] : 1.0f;
    const float p = __expf(a.raw_logits[i] * mk - lse);
    if (pi) pi[i] = p;
    if (prob) prob[i] = MIX ? p * mk / clamp_sigma(sigmoid_f(a.raw_sigma[i])) * invS : p;
  }
}

// Backward.  With w_n = pi_n m_n / sigma_n, S = sum w, P_n = w_n / S, disp = sum P_n d_n and upstream gD = d loss/d disp
// (+ the depth term): d disp / d w_n = (d_n - disp) / S, and since sum_k pi_k (d loss / d pi_k) = gD/S * sum_k w_k
// (d_k - disp) = 0 exactly, the softmax backward needs no second reduction:
//   g_logits_n += gD (d_n - disp) P_n;   g_sigma_n -= gD (d_n - disp) P_n / sigma_n;   g_d_n = gD P_n.
template <bool MIX, bool HASMASK>
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
  const int pix = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
  const bool reduce = (g_dl != nullptr) && !a.dense;
  if (reduce) {
    for (int i = threadIdx.x; i < a.N; i += kBlock) red[i] = 0.0f;
    __syncthreads();
  }
  const bool active = pix < a.HW;
  const long base = (long)b * a.N * a.HW + (active ? pix : 0);
  float lse = 0.0f, invS = 0.0f, dsp = 1.0f, gD = 0.0f;
  if (active) {
    lse = stash[((long)b * 2 + 0) * a.HW + pix];
    invS = 1.0f / stash[((long)b * 2 + 1) * a.HW + pix];
    dsp = disp[(long)b * a.HW + pix];
    if (g_disp) gD = g_disp[(long)b * a.HW + pix];
    if (g_depth) gD -= g_depth[(long)b * a.HW + pix] * (0.1f * 0.58f * (float)a.W) / (dsp * dsp);
  }
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll 2
  for (int n = 0; n < a.N; ++n) {
    const long i = base + (long)n * a.HW;
    float gd = 0.0f;
    if (active) {
      const float mk = HASMASK ? a.mask[i] : 1.0f;
      const float p = __expf(a.raw_logits[i] * mk - lse);
      float sgu = 1.0f, sg = 1.0f, P = p;
      if (MIX) {
        sgu = sigmoid_f(a.raw_sigma[i]);
        sg = clamp_sigma(sgu);
        P = p * mk / sg * invS;
      }
      const float d = a.dense ? a.dl[i] : a.dl[b * a.N + n];
      const float t = gD * (d - dsp) * P;
      if (g_raw_logits) g_raw_logits[i] = ((g_logits ? g_logits[i] : 0.0f) + t) * mk;
      if (MIX && g_raw_sigma) {
        const float gs = (g_sigma ? g_sigma[i] : 0.0f) - t / sg;
        g_raw_sigma[i] = (sgu == sg) ? gs * sgu * (1.0f - sgu) : 0.0f;  // clamp gate (inclusive bounds), sigmoid'
      }
      gd = gD * P;
      if (g_dl && a.dense) g_dl[i] = gd;
    }
    if (reduce) {
      const float v = wave_sum_hi(gd);
      if (lane == kWave - 1) atomicAdd(&red[n], v);
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

#define PD_TAIL_DISPATCH(KERNEL, mix, hasmask, grid, shmem, stream, ...)                                     \
  do {                                                                                                       \
    if (mix) {                                                                                               \
      if (hasmask) KERNEL<true, true><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                         \
      else         KERNEL<true, false><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                        \
    } else {                                                                                                 \
      if (hasmask) KERNEL<false, true><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                        \
      else         KERNEL<false, false><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                       \
    }                                                                                                        \
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
  dim3 grid(ceil_div(H * W, kBlock), B);
  PD_TAIL_DISPATCH(tail_fwd_kernel, a.mix, padding_mask != nullptr, grid, 0, (hipStream_t)stream, a, logits, sigma, disp,
                   depth, stash);
  return check_launch("tail_fwd_kernel");
}

extern "C" int pd_decoder_tail_layers(int B, int N, int H, int W, int flags, const float* raw_logits,
                                      const float* raw_sigma, const float* padding_mask, const float* stash, float* pi,
                                      float* probability, pd_stream_t stream) {
  if (int rc = tail_validate(B, N, H, W, flags, raw_logits, raw_sigma, raw_logits)) return rc;
  PD_REQUIRE(stash && (pi || probability), "NULL pointer");
  const TailArgs a = tail_args(N, H, W, flags, raw_logits, raw_sigma, padding_mask, nullptr);
  dim3 grid(ceil_div(H * W, kBlock), B);
  PD_TAIL_DISPATCH(tail_layers_kernel, a.mix, padding_mask != nullptr, grid, 0, (hipStream_t)stream, a, stash, pi,
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
  dim3 grid(ceil_div(H * W, kBlock), B);
  const size_t shmem = reduce ? (size_t)N * sizeof(float) : 0;
  PD_TAIL_DISPATCH(tail_bwd_kernel, a.mix, padding_mask != nullptr, grid, shmem, (hipStream_t)stream, a, stash, disp,
                   g_logits, g_sigma, g_disp, g_depth, g_raw_logits, g_raw_sigma, g_disp_layered, workspace);
  if (int rc = check_launch("tail_bwd_kernel")) return rc;
  if (reduce) {
    tail_reduce_kernel<<<dim3(N, B), kWave, 0, (hipStream_t)stream>>>(workspace, g_disp_layered, (int)grid.x, N);
    return check_launch("tail_reduce_kernel");
  }
  return 0;
}
