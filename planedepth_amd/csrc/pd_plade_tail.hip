// Fused tail of PladeNet.forward with --render_probability (reference networks/plade_net.py:309-341; the only live producer
// of outputs["dists"], which the sweep's alpha-compositing branch reads, trainer.py:584-591): everything the network does
// with the outputs of conv0 (N-1 logit channels) and conv_sigma, in ONE front-to-back pass over the planes:
//   depth_layered = 0.1 * 0.58 * W / disp_layered;  dists_n = (depth_layered_{n+1} - depth_layered_n) * |K^-1 [x, y, 1]|
//   alpha_n = 1 - exp(-relu(logit_n) * dists_n)  (n < N-1),  alpha_{N-1} = 1
//   pi_n = alpha_n * prod_{m<n} (1 - alpha_m + 1e-10);  logits = cat(raw_logits, ones)
//   sigma = clamp(sigmoid(raw_sigma), .01, 1);  probability = (pi / sigma) / sum_N(pi / sigma)   [mixture]  |  pi
//   disp = sum_N probability * disp_layered;  depth = 0.1 * 0.58 * W / disp
// The reference runs ~20 full-tensor ATen passes (two cats, a cumprod, ...).  One thread owns one pixel (four with 16-byte
// accesses where the shapes allow) and streams its planes once; pi / probability are produced on demand
// (pd_plade_tail_layers).  Backward: with p_n = alpha_n T_n, g_alpha_n = g_p_n T_n - (sum_{k>n} g_p_k p_k) / (1 - alpha_n +
// 1e-10), and sum_k g_p_k p_k is known in closed form — 0 with the mixture weights (d disp / d u_k = (d_k - disp) / S sums
// to zero against u), g_disp * disp without — so the backward is one front-to-back pass too; the distance gradient
// reaches disp_layered through the two depth layers it subtracts.
#include "pd_tail_common.h"

namespace pd {

struct PladeArgs {
  int N, HW, W;
  int mix, dense;
  const float* raw_logits;   // [B,N-1,H,W]
  const float* raw_sigma;    // [B,N,H,W]
  const float* dl;           // [B,N] or [B,N,H,W]
  const float* ray;          // [H*W]
};

template <int PX>
__device__ __forceinline__ Px<PX> plade_disp(const PladeArgs& a, int b, int n, long pix) {
  return a.dense ? ldv<PX>(a.dl + ((long)b * a.N + n) * a.HW + pix) : splat<PX>(a.dl[b * a.N + n]);
}

template <bool MIX, int PX>
__global__ __launch_bounds__(kBlock) void plade_fwd_kernel(PladeArgs a, float* __restrict__ logits, float* __restrict__ dists,
                                                           float* __restrict__ sigma, float* __restrict__ disp,
                                                           float* __restrict__ depth, float* __restrict__ stash) {
  const int pix = (blockIdx.x * kBlock + threadIdx.x) * PX, b = blockIdx.y;
  if (pix >= a.HW) return;
  const int N = a.N;
  const float c = 0.1f * 0.58f * (float)a.W;
  const Px<PX> r = ldv<PX>(a.ray + pix);
  float T[PX], Sw[PX], Sd[PX], zc[PX];
  Px<PX> dv = plade_disp<PX>(a, b, 0, pix);
#pragma unroll
  for (int j = 0; j < PX; ++j) { T[j] = 1.0f; Sw[j] = Sd[j] = 0.0f; zc[j] = c / dv.v[j]; }
  for (int n = 0; n < N; ++n) {
    const bool last = (n == N - 1);
    const Px<PX> dn = last ? dv : plade_disp<PX>(a, b, n + 1, pix);       // disparity of the NEXT plane
    const Px<PX> rl = last ? splat<PX>(0.0f) : ldv<PX>(a.raw_logits + ((long)b * (N - 1) + n) * a.HW + pix);
    const Px<PX> rs = MIX ? ldv<PX>(a.raw_sigma + ((long)b * N + n) * a.HW + pix) : splat<PX>(0.0f);
    Px<PX> o_l, o_t, o_s;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      float alpha = 1.0f, zn = zc[j];
      o_l.v[j] = 1.0f;                                                     // :322 the appended ones channel
      if (!last) {
        zn = c / dn.v[j];                                                  // :311
        const float dist = (zn - zc[j]) * r.v[j];                          // :312-315
        o_t.v[j] = dist;
        o_l.v[j] = rl.v[j];
        alpha = 1.0f - __expf(-fmaxf(rl.v[j], 0.0f) * dist);               // :317
      }
      const float p = alpha * T[j];                                        // :320
      T[j] *= (1.0f - alpha) + 1e-10f;
      if (MIX) {
        const float sg = clamp_sigma(sigmoid_f(rs.v[j]));                  // :327-328
        o_s.v[j] = sg;
        const float u = p / sg;                                            // :331
        Sw[j] += u;
        Sd[j] += u * dv.v[j];
      } else {
        Sd[j] += p * dv.v[j];
      }
      zc[j] = zn;
    }
    stv<PX>(logits + ((long)b * N + n) * a.HW + pix, o_l);
    if (!last) stv<PX>(dists + ((long)b * (N - 1) + n) * a.HW + pix, o_t);
    if (MIX) stv<PX>(sigma + ((long)b * N + n) * a.HW + pix, o_s);
    dv = dn;
  }
  Px<PX> o_disp, o_depth, o_sw;
#pragma unroll
  for (int j = 0; j < PX; ++j) {
    const float dsp = MIX ? Sd[j] / Sw[j] : Sd[j];                         // :332-333, 338
    o_disp.v[j] = dsp;
    o_depth.v[j] = c / dsp;                                                // :340
    o_sw.v[j] = MIX ? Sw[j] : 1.0f;
  }
  stv<PX>(disp + (long)b * a.HW + pix, o_disp);
  stv<PX>(depth + (long)b * a.HW + pix, o_depth);
  stv<PX>(stash + (long)b * a.HW + pix, o_sw);
}

// pi and probability (plade_net.py:320, 330-333) for callers that want the tensors
template <bool MIX, int PX>
__global__ __launch_bounds__(kBlock) void plade_layers_kernel(PladeArgs a, const float* __restrict__ stash, float* __restrict__ pi,
                                                              float* __restrict__ prob) {
  const int pix = (blockIdx.x * kBlock + threadIdx.x) * PX, b = blockIdx.y;
  if (pix >= a.HW) return;
  const int N = a.N;
  const float c = 0.1f * 0.58f * (float)a.W;
  const Px<PX> r = ldv<PX>(a.ray + pix), sw = ldv<PX>(stash + (long)b * a.HW + pix);
  float T[PX], zc[PX];
  Px<PX> dv = plade_disp<PX>(a, b, 0, pix);
#pragma unroll
  for (int j = 0; j < PX; ++j) { T[j] = 1.0f; zc[j] = c / dv.v[j]; }
  for (int n = 0; n < N; ++n) {
    const bool last = (n == N - 1);
    const Px<PX> dn = last ? dv : plade_disp<PX>(a, b, n + 1, pix);
    const Px<PX> rl = last ? splat<PX>(0.0f) : ldv<PX>(a.raw_logits + ((long)b * (N - 1) + n) * a.HW + pix);
    const Px<PX> rs = MIX ? ldv<PX>(a.raw_sigma + ((long)b * N + n) * a.HW + pix) : splat<PX>(0.0f);
    Px<PX> op, oq;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      float alpha = 1.0f, zn = zc[j];
      if (!last) {
        zn = c / dn.v[j];
        alpha = 1.0f - __expf(-fmaxf(rl.v[j], 0.0f) * ((zn - zc[j]) * r.v[j]));
      }
      const float p = alpha * T[j];
      T[j] *= (1.0f - alpha) + 1e-10f;
      op.v[j] = p;
      oq.v[j] = MIX ? p / clamp_sigma(sigmoid_f(rs.v[j])) / sw.v[j] : p;
      zc[j] = zn;
    }
    if (pi) stv<PX>(pi + ((long)b * N + n) * a.HW + pix, op);
    if (prob) stv<PX>(prob + ((long)b * N + n) * a.HW + pix, oq);
    dv = dn;
  }
}

template <bool MIX, int PX>
__global__ __launch_bounds__(kBlock) void plade_bwd_kernel(PladeArgs a, const float* __restrict__ stash, const float* __restrict__ disp,
                                                           const float* __restrict__ g_logits, const float* __restrict__ g_dists,
                                                           const float* __restrict__ g_sigma, const float* __restrict__ g_disp,
                                                           const float* __restrict__ g_depth, float* __restrict__ g_raw_logits,
                                                           float* __restrict__ g_raw_sigma, float* __restrict__ g_dl,
                                                           float* __restrict__ partials) {
  extern __shared__ float red[];  // [N] block sums of the per-plane disparity gradient
  const int pix = (blockIdx.x * kBlock + threadIdx.x) * PX, b = blockIdx.y;
  const int N = a.N;
  const bool reduce = (g_dl != nullptr) && !a.dense;
  if (reduce) {
    for (int i = threadIdx.x; i < N; i += kBlock) red[i] = 0.0f;
    __syncthreads();
  }
  const bool active = pix < a.HW;
  const long px0 = active ? pix : 0;
  const float c = 0.1f * 0.58f * (float)a.W;
  Px<PX> r = splat<PX>(0.0f), sw = splat<PX>(1.0f), dsp = splat<PX>(1.0f), gD = splat<PX>(0.0f);
  if (active) {
    r = ldv<PX>(a.ray + pix);
    sw = ldv<PX>(stash + (long)b * a.HW + pix);
    dsp = ldv<PX>(disp + (long)b * a.HW + pix);
    if (g_disp) gD = ldv<PX>(g_disp + (long)b * a.HW + pix);
    if (g_depth) {
      const Px<PX> gz = ldv<PX>(g_depth + (long)b * a.HW + pix);
#pragma unroll
      for (int j = 0; j < PX; ++j) gD.v[j] -= gz.v[j] * c / (dsp.v[j] * dsp.v[j]);
    }
  }
  float T[PX], zc[PX], prefix[PX], gprev[PX];
  Px<PX> dv = plade_disp<PX>(a, b, 0, px0);
#pragma unroll
  for (int j = 0; j < PX; ++j) { T[j] = 1.0f; zc[j] = c / dv.v[j]; prefix[j] = 0.0f; gprev[j] = 0.0f; }
  const int lane = threadIdx.x & (kWave - 1);
  for (int n = 0; n < N; ++n) {
    const bool last = (n == N - 1);
    float gd_sum = 0.0f;
    if (active) {
      const Px<PX> dn = last ? dv : plade_disp<PX>(a, b, n + 1, px0);
      const Px<PX> rl = last ? splat<PX>(0.0f) : ldv<PX>(a.raw_logits + ((long)b * (N - 1) + n) * a.HW + pix);
      const Px<PX> rs = MIX ? ldv<PX>(a.raw_sigma + ((long)b * N + n) * a.HW + pix) : splat<PX>(0.0f);
      const Px<PX> gl = (g_logits && !last) ? ldv<PX>(g_logits + ((long)b * N + n) * a.HW + pix) : splat<PX>(0.0f);
      const Px<PX> gt = (g_dists && !last) ? ldv<PX>(g_dists + ((long)b * (N - 1) + n) * a.HW + pix) : splat<PX>(0.0f);
      const Px<PX> gs = (MIX && g_sigma) ? ldv<PX>(g_sigma + ((long)b * N + n) * a.HW + pix) : splat<PX>(0.0f);
      Px<PX> o_l, o_s, o_d;
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        float alpha = 1.0f, zn = zc[j], dist = 0.0f;
        if (!last) {
          zn = c / dn.v[j];
          dist = (zn - zc[j]) * r.v[j];
          alpha = 1.0f - __expf(-fmaxf(rl.v[j], 0.0f) * dist);
        }
        const float p = alpha * T[j];
        float g_p, gd_direct, rtot;
        if (MIX) {
          const float sgu = sigmoid_f(rs.v[j]), sg = clamp_sigma(sgu);
          const float u = p / sg;
          const float g_u = gD.v[j] * (dv.v[j] - dsp.v[j]) / sw.v[j];      // d disp / d u_n = (d_n - disp) / S
          g_p = g_u / sg;
          const float gsig = gs.v[j] - g_u * u / sg;                       // d u / d sigma = -u / sigma
          o_s.v[j] = (sgu == sg) ? gsig * sgu * (1.0f - sgu) : 0.0f;       // clamp gate (inclusive bounds), sigmoid'
          gd_direct = gD.v[j] * u / sw.v[j];                               // d disp / d d_n = probability_n
          rtot = 0.0f;                                                     // sum_k g_p_k p_k = g_disp (disp - disp)
        } else {
          g_p = gD.v[j] * dv.v[j];
          gd_direct = gD.v[j] * p;
          rtot = gD.v[j] * dsp.v[j];
        }
        prefix[j] += g_p * p;
        const float keep = (1.0f - alpha) + 1e-10f;
        float g_dist = 0.0f;
        if (!last) {
          const float g_alpha = g_p * T[j] - (rtot - prefix[j]) / keep;
          const float da = 1.0f - alpha;                                   // exp(-relu(l) dist)
          o_l.v[j] = gl.v[j] + ((rl.v[j] > 0.0f) ? g_alpha * dist * da : 0.0f);
          g_dist = gt.v[j] + g_alpha * fmaxf(rl.v[j], 0.0f) * da;
        }
        // depth layer n is the far end of distance n-1 and the near end of distance n
        const float g_z = r.v[j] * (gprev[j] - g_dist);
        o_d.v[j] = gd_direct - g_z * c / (dv.v[j] * dv.v[j]);              // depth_layered = c / disp_layered
        gd_sum += o_d.v[j];
        T[j] *= keep;
        gprev[j] = g_dist;
        zc[j] = zn;
      }
      if (g_raw_logits && !last) stv<PX>(g_raw_logits + ((long)b * (N - 1) + n) * a.HW + pix, o_l);
      if (MIX && g_raw_sigma) stv<PX>(g_raw_sigma + ((long)b * N + n) * a.HW + pix, o_s);
      if (g_dl && a.dense) stv<PX>(g_dl + ((long)b * N + n) * a.HW + pix, o_d);
      dv = dn;
    }
    if (reduce) {
      const float v = wave_sum_hi(gd_sum);
      if (lane == kWave - 1) lds_add(&red[n], v);
    }
  }
  if (reduce) {
    __syncthreads();
    float* dst = partials + ((long)b * gridDim.x + blockIdx.x) * N;
    for (int i = threadIdx.x; i < N; i += kBlock) dst[i] = red[i];
  }
}

// partials [B][R][N] -> out [B][N]; one wave per (n, b); fixed order
__global__ void plade_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out, int R, int N) {
  const int n = blockIdx.x, b = blockIdx.y;
  const float* p = partials + (long)b * R * N + n;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < R; i += kWave) acc += p[(long)i * N];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[(long)b * N + n] = acc;
}

static int plade_validate(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                          const float* dl, const float* ray) {
  PD_REQUIRE(B > 0 && B <= 65535 && N >= 2 && H > 0 && W > 0, "bad shape (alpha compositing needs N >= 2 planes)");
  PD_REQUIRE((long)H * W < (1L << 31), "image too large");
  PD_REQUIRE((flags & ~(PD_TAIL_MIXTURE | PD_TAIL_DISP_DENSE)) == 0, "unknown flags");
  PD_REQUIRE(raw_logits && dl && ray, "NULL pointer");
  PD_REQUIRE(!(flags & PD_TAIL_MIXTURE) || raw_sigma, "mixture needs raw_sigma");
  return 0;
}

static PladeArgs plade_args(int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma, const float* dl,
                            const float* ray) {
  PladeArgs a;
  a.N = N; a.HW = H * W; a.W = W;
  a.mix = (flags & PD_TAIL_MIXTURE) != 0;
  a.dense = (flags & PD_TAIL_DISP_DENSE) != 0;
  a.raw_logits = raw_logits; a.raw_sigma = raw_sigma; a.dl = dl; a.ray = ray;
  return a;
}

#define PD_PLADE_DISPATCH(KERNEL, px, mix, grid, shmem, stream, ...)                                        \
  do {                                                                                                       \
    if ((px) == 4) { if (mix) KERNEL<true, 4><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                 \
                     else     KERNEL<false, 4><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__); }              \
    else           { if (mix) KERNEL<true, 1><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__);                 \
                     else     KERNEL<false, 1><<<grid, kBlock, shmem, stream>>>(__VA_ARGS__); }              \
  } while (0)

}  // namespace pd

using namespace pd;

extern "C" int pd_plade_tail_fwd(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                                 const float* disp_layered, const float* ray_norm, float* logits, float* dists, float* sigma,
                                 float* disp, float* depth, float* stash, pd_stream_t stream) {
  if (int rc = plade_validate(B, N, H, W, flags, raw_logits, raw_sigma, disp_layered, ray_norm)) return rc;
  PD_REQUIRE(logits && dists && disp && depth && stash, "NULL output");
  PD_REQUIRE(!(flags & PD_TAIL_MIXTURE) || sigma, "mixture needs the sigma output");
  const PladeArgs a = plade_args(N, H, W, flags, raw_logits, raw_sigma, disp_layered, ray_norm);
  const int px = tail_px(H, W, {raw_logits, raw_sigma, a.dense ? disp_layered : nullptr, ray_norm, logits, dists, sigma, disp,
                                depth, stash});
  dim3 grid(ceil_div(ceil_div(H * W, px), kBlock), B);
  PD_PLADE_DISPATCH(plade_fwd_kernel, px, a.mix, grid, 0, (hipStream_t)stream, a, logits, dists, sigma, disp, depth, stash);
  return check_launch("plade_fwd_kernel");
}

extern "C" int pd_plade_tail_layers(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                                    const float* disp_layered, const float* ray_norm, const float* stash, float* pi,
                                    float* probability, pd_stream_t stream) {
  if (int rc = plade_validate(B, N, H, W, flags, raw_logits, raw_sigma, disp_layered, ray_norm)) return rc;
  PD_REQUIRE(stash && (pi || probability), "NULL pointer");
  const PladeArgs a = plade_args(N, H, W, flags, raw_logits, raw_sigma, disp_layered, ray_norm);
  const int px = tail_px(H, W, {raw_logits, raw_sigma, a.dense ? disp_layered : nullptr, ray_norm, stash, pi, probability});
  dim3 grid(ceil_div(ceil_div(H * W, px), kBlock), B);
  PD_PLADE_DISPATCH(plade_layers_kernel, px, a.mix, grid, 0, (hipStream_t)stream, a, stash, pi, probability);
  return check_launch("plade_layers_kernel");
}

extern "C" int pd_plade_tail_bwd(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                                 const float* disp_layered, const float* ray_norm, const float* stash, const float* disp,
                                 const float* g_logits, const float* g_dists, const float* g_sigma, const float* g_disp,
                                 const float* g_depth, float* g_raw_logits, float* g_raw_sigma, float* g_disp_layered,
                                 float* workspace, pd_stream_t stream) {
  if (int rc = plade_validate(B, N, H, W, flags, raw_logits, raw_sigma, disp_layered, ray_norm)) return rc;
  PD_REQUIRE(stash && disp, "NULL pointer");
  PD_REQUIRE(g_raw_logits || g_raw_sigma || g_disp_layered, "no gradient requested");
  const PladeArgs a = plade_args(N, H, W, flags, raw_logits, raw_sigma, disp_layered, ray_norm);
  const bool reduce = g_disp_layered && !a.dense;
  PD_REQUIRE(!reduce || workspace, "per-plane disparity gradient needs the workspace (pd_decoder_tail_bwd_workspace_floats)");
  PD_REQUIRE((size_t)N * sizeof(float) <= 64 * 1024, "too many planes");
  const int px = tail_px(H, W, {raw_logits, raw_sigma, a.dense ? disp_layered : nullptr, ray_norm, stash, disp, g_logits, g_dists,
                                g_sigma, g_disp, g_depth, g_raw_logits, g_raw_sigma, a.dense ? g_disp_layered : nullptr});
  dim3 grid(ceil_div(ceil_div(H * W, px), kBlock), B);
  const size_t shmem = reduce ? (size_t)N * sizeof(float) : 0;
  PD_PLADE_DISPATCH(plade_bwd_kernel, px, a.mix, grid, shmem, (hipStream_t)stream, a, stash, disp, g_logits, g_dists, g_sigma,
                    g_disp, g_depth, g_raw_logits, g_raw_sigma, g_disp_layered, workspace);
  if (int rc = check_launch("plade_bwd_kernel")) return rc;
  if (reduce) {
    plade_reduce_kernel<<<dim3(N, B), kWave, 0, (hipStream_t)stream>>>(workspace, g_disp_layered, (int)grid.x, N);
    return check_launch("plade_reduce_kernel");
  }
  return 0;
}
