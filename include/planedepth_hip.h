/*
 * planedepth_hip.h — C ABI of the MI355X (gfx950) implementation of PlaneDepth's
 * photometric-reconstruction hot path.
 *
 * The reference (svip-lab/PlaneDepth) has NO native/FFI boundary for this path: it is
 * Python calling stock ATen ops.  This header is therefore the
 * boundary a maintainer would bind with ctypes; every entry point
 * cites the reference code it replaces.
 *
 * Conventions
 *   - all tensors are fp32, contiguous NCHW device pointers owned by the caller
 *     (outputs are caller-allocated; nothing is allocated, freed or synchronised here);
 *   - `stream` is a hipStream_t (0 = the null stream); launches are asynchronous;
 *   - every function returns PD_OK (0) or a PD_ERR_* code; pd_last_error() returns a
 *     thread-local human-readable message for the last non-zero return on this thread;
 *   - safe to call from any thread (e.g. the autograd thread) and for any device: the entry points work on the CURRENT
 *     device (hipSetDevice / torch.cuda.device by the caller) and keep nothing between calls except per-device caches of
 *     device facts — the LDS a workgroup can be given, the dynamic-LDS limit already granted to a kernel — held in
 *     atomics indexed by the device ordinal (the rare raise is serialised by a mutex).  The few process-environment
 *     tuning switches (PD_NO_ROWPAIR, PD_ROW_WAVES, PD_UNI_CHUNK, PD_PP_ROWS, PD_PP_SEG, PD_PP_CHAIN, PD_FWD_STREAM, PD_ROW_EPS) are read ONCE, when
 *     the library is first used, never on the launch path; kernel selection per call goes through pd_sweep_desc.impl.
 */
#ifndef PLANEDEPTH_HIP_H
#define PLANEDEPTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pd_stream_t; /* hipStream_t */

enum pd_status {
  PD_OK = 0,
  PD_ERR_ARG = 1,         /* NULL pointer / bad shape / bad enum */
  PD_ERR_UNSUPPORTED = 2, /* valid request this build does not implement */
  PD_ERR_LAUNCH = 3       /* hipGetLastError() != hipSuccess after a launch */
};

/* How the per-plane sampling grid is generated (reference: opt.warp_type, trainer.py:533-560). */
enum pd_warp_mode {
  PD_WARP_DISP = 0,      /* trainer.py:540-554: x -> x + sign*disp_layered, y unchanged          */
  PD_WARP_HOMOGRAPHY = 1 /* layers.py:221-233: p = H_t2s [x,y,1]^T, projective divide, mask      */
};

enum pd_sweep_flags {
  PD_MIXTURE = 1,     /* opt.use_mixture_loss: sigma channel, Laplacian-mixture NLL (trainer.py:594-602, 728-730) */
  PD_AUTOMASK = 2,    /* opt.automask: min with the identity-reprojection loss (trainer.py:731-734, 739-741)      */
  PD_RENDER_PROB = 4, /* opt.render_probability: alpha compositing instead of softmax (trainer.py:584-591)        */
  PD_DISP_DENSE = 8,  /* disp mode: `plane` is a dense [B,N,H,W] map (xz/yz planes) instead of [B,N] scalars      */
  PD_DISP_ROWS = 16,  /* disp mode: `plane` is [B,N,H], one disparity per plane and ROW (xy + xz planes: the decoder's
                         map is constant along x, depth_decoder.py:153-181).  Only where pd_sweep_uses_rowshift() */
  PD_MASK_ROWS = 32   /* disp mode: `padding_mask` is [B,N,H], one value per plane and row (the xz horizon mask is
                         constant along x, depth_decoder.py:166).  Only where pd_sweep_uses_rowshift(): a masked plane
                         row is treated as shifted out of view, so the mask costs no per-pixel traffic at all */
  ,
  PD_HOMO_UNIFORM = 64 /* homography mode: every plane of an image shares ONE homography.  What Trainer.predict_poses hands
                         over for the novel frames without COLMAP: zero translation (trainer.py:386-400), hence
                         K (R + t n^T/d) K^-1 = K R K^-1 for every plane.  `plane` is [B,4,3,3]: slice 0 = H_t2s, slices
                         1..3 = the homographies of virtual planes with n/d = e_0, e_1, e_2 (same values at t = 0);
                         `plane_aux` stays [B*N,3] (the facing test depends on the plane's normal); `padding_mask` carries
                         the [B,N,3] weights n_n/d_n or NULL.  g_plane [B,4,3,3] = (sum_n G_n, sum_n G_n n_n[j]/d_n):
                         back-propagated through f(R + t e_j^T) it gives the exact translation gradient of the per-plane
                         formulation.  Served by pd_plane_sweep_uniform.hip: geometry once per pixel, atomic-free
                         two-pass backward */
  ,
  PD_BWD_ACCUMULATE = 128 /* pd_plane_sweep_bwd only: g_logits / g_sigma are ADDED TO instead of overwritten, so the target
                         views of one step (trainer.py:532: the same logits / sigma feed every side) sum their
                         gradients in place instead of through [B,N,H,W]-sized add kernels.  Honoured where
                         pd_sweep_bwd_accumulates() says so (the plane-uniform kernels: read-modify-write stores; the
                         general kernels: their atomics simply skip the zero-fill); refused elsewhere */
  ,
  PD_BWD_DEFER_GATHER = 256 /* pd_plane_sweep_bwd, PD_HOMO_UNIFORM only: run the first pass (per-pixel plane gradients into the
                         scratch inside `workspace`, g_plane, g_dists) and leave g_logits / g_sigma untouched; the caller
                         then hands the workspaces of TWO such calls over the same logits / sigma to
                         pd_uniform_gather_pair, which gathers both views in one kernel (one store per gradient element
                         instead of a read-modify-write per view) */
  ,
  PD_PH_MEAN_ZEROED = 512 /* pd_plane_sweep_fwd: the caller hands `ph_mean` over holding 0.0f already (e.g. a slot of a buffer it
                         zeroed once for many calls): the kernels add into it, and the entry point then issues no memset
                         launch of its own (4-5 us per call next to a 0.1 ms kernel).  Ignored by the other entry points */
  ,
  PD_BWD_PLANE_ZEROED = 1024 /* pd_plane_sweep_bwd / _bwd_tail, PD_WARP_DISP with one disparity per plane: the caller hands `g_plane`
                         [B,N] over holding zeros (e.g. a slice of a buffer it zeroed once for many calls).  Where
                         pd_sweep_bwd_plane_adds(d) == 1 (the row-stream backward) every row workgroup then ADDS its share with
                         atomics and the entry point launches no reduction kernel of its own (4-5 us + a launch gap per call next
                         to a 0.18 ms kernel; the sum's order, hence its last bits, varies from run to run); elsewhere the kernels
                         overwrite `g_plane` as always — the promise is harmless there */
};

enum pd_padding_mode { PD_PAD_ZEROS = 0, PD_PAD_BORDER = 1 };

/* Kernel selection.  The row kernels apply to PD_WARP_DISP with per-plane or per-row disparities (forward:
 * pd_plane_sweep_fwdstream.hip, a wave per 128-pixel segment streaming over the planes — pd_plane_sweep_rowshift.hip,
 * plane groups, with a per-pixel mask or PD_RENDER_PROB; backward: pd_plane_sweep_rowstream.hip, source-ordered); homography_warp
 * with one matrix per image (PD_HOMO_UNIFORM) or per plane runs a two-pass gather backward without atomics
 * (pd_plane_sweep_uniform.hip, pd_plane_sweep_gather.hip); the general kernels handle everything (and are the
 * cross-check for the specialised ones in the tests). */
enum pd_sweep_impl {
  PD_IMPL_AUTO = 0,      /* the specialised kernels where they apply, exact footprints (default) */
  PD_IMPL_GENERAL = 1,   /* general kernels only: atomic scatter in the backward (the cross-check in the tests) */
  PD_IMPL_FAST_ROWS = 2  /* as AUTO, but a second source row whose bilinear weight is below 2^-16 (fp32 noise of the
                            reference's y round trip, <= 6e-6) is dropped: ~11% faster, results within 1e-4 of the
                            tensors' range on random inputs instead of 1e-6 (opt-in) */
  ,
  PD_IMPL_TILE = 3       /* EXPERIMENTS BUILD ONLY (-DPD_EXPERIMENTS, pd_experiments() == 1; otherwise PD_ERR_UNSUPPORTED):
                            as GENERAL, and homography_warp's backward runs the owned-tile kernel (pd_plane_sweep_tile.hip:
                            LDS accumulators per source tile, plain stores, no zero-fill) instead of the atomic scatter.
                            Exact and atomic-free in HBM, but 2-2.5x SLOWER on gfx950 (ds_add_f32 costs ~110 cycles per
                            wave instruction: NOTEBOOK.md 3.4.6) - kept as an in-suite cross-check and as the record of
                            that measurement */
  ,
  PD_IMPL_ROWS1 = 4      /* as AUTO, but forward and backward are the target-ordered, one-pixel-per-lane row-shift kernels
                            (pd_plane_sweep_rowshift.hip, the headline kernels of rounds 1-2) instead of the segment-stream
                            forward (pd_plane_sweep_fwdstream.hip) and the source-ordered row-stream backward
                            (pd_plane_sweep_rowstream.hip): cross-check and A/B runs */
  ,
  PD_IMPL_UNIFORM_DIRECT = 5 /* as AUTO, but pass 2 of the two-pass homography backwards (plane-uniform and per-plane) gathers
                            directly from the scratch instead of staging it through LDS (the form large boxes fall back
                            to anyway): cross-check */
  ,
  PD_IMPL_EXACT_ROWS = 6 /* as AUTO, with every second source row served whatever its weight: the reference's last-ulp row
                            weights (the y round trip of trainer.py:552 + grid_sample returns y + e, |e| <= 6e-6, on a
                            quarter of the rows at H = 192).  What AUTO itself does about those rows is stated at
                            PD_IMPL_AUTO / pd_sweep_auto_fast_rows() */
};

/* The bilinear weight below which PD_IMPL_AUTO (and the impls documented "as AUTO") drops a second source row: 0 = never
 * (as PD_IMPL_EXACT_ROWS); PD_IMPL_FAST_ROWS uses 2^-16. */
float pd_sweep_auto_row_eps(void);

typedef struct pd_sweep_desc {
  int32_t B, N, H, W;
  int32_t mode;  /* pd_warp_mode */
  int32_t flags; /* OR of pd_sweep_flags */
  float sign;    /* PD_WARP_DISP: +1 for target "r" (x + d), -1 for target "l" (x - d), 0 = grid untouched */
  int32_t impl;  /* pd_sweep_impl: 0 = pick the fastest applicable kernels, 1 = force the general (atomic) kernels */
} pd_sweep_desc;

int pd_version(void);
const char* pd_last_error(void);
/* sha256[:16] over the kernel sources and headers this library was compiled from (the .hip and .h files under csrc/ and
 * include/: __graft_entry__.source_hash), baked in at build time (-DPD_SRC_HASH); "unknown" for a build without it. */
const char* pd_source_hash(void);
/* 1 if the library was built with -DPD_EXPERIMENTS: the kernels measured SLOWER than the defaults (four-pixels-per-lane
 * row kernels, owned-tile backward, one-kernel plane-uniform backward; NOTEBOOK.md 3.5) are then compiled in and selectable
 * (PD_IMPL_TILE; PD_QUAD_FWD / PD_QUAD_BWD / PD_UNI_FUSED in the environment).  The product library returns 0. */
int pd_experiments(void);
/* What else this binary was compiled with, as bits: 1 = -DPD_EXPERIMENTS (as pd_experiments()); 2 = -DPD_DIAGNOSTICS — a
 * timing-ablation or trace build of the headline kernels (PD_FS_ABL, PD_STREAM_ABL, PD_ABLATE, PD_FS_TRACE, ...: parts of the
 * arithmetic or of the memory traffic compiled OUT, results WRONG by design; those switches refuse to compile without
 * -DPD_DIAGNOSTICS).  The product library returns 0; tests/test_capi.py and bench.py's `library` block check it. */
int pd_build_flags(void);

/* 1 if pd_plane_sweep_bwd adds into a pre-zeroed g_plane under PD_BWD_PLANE_ZEROED for this descriptor (see the flag; a
 * per-pixel padding mask sends the call to a kernel that overwrites instead), else 0. */
int pd_sweep_bwd_plane_adds(const pd_sweep_desc* d);
/* 1 if pd_plane_sweep_bwd honours PD_BWD_ACCUMULATE for this descriptor (see the flag), else 0. */
int pd_sweep_bwd_accumulates(const pd_sweep_desc* d);
/* 1 if this descriptor is served by the row-shift kernels (PD_WARP_DISP, scalar or per-row disparities), else 0. */
int pd_sweep_uses_rowshift(const pd_sweep_desc* d);

/* Floats per image the forward pass stashes for the backward pass (softmax statistics + mask bits). */
size_t pd_sweep_stash_floats(const pd_sweep_desc* d);
/* Floats of scratch pd_plane_sweep_bwd needs for its per-block partial reductions of the plane-parameter gradient. */
size_t pd_sweep_bwd_workspace_floats(const pd_sweep_desc* d);

/*
 * Fused replacement for Trainer.pred_novel_images (trainer.py:523-603, one target view) plus the photometric part
 * of Trainer.compute_losses (trainer.py:717-742) — SURVEY.md rows A1, A2/A3, A5-A9.
 *
 *   src, tgt      [B,3,H,W]  source colour (inputs[(color,"l")]) and target colour
 *   logits        [B,N,H,W]  outputs["logits"];  sigma [B,N,H,W] outputs["sigma"] (NULL unless PD_MIXTURE)
 *   plane         PD_WARP_DISP:       outputs["disp_layered"] as [B,N] ([B,N,H,W] with PD_DISP_DENSE, [B,N,H] with
 *                                     PD_DISP_ROWS)
 *                 PD_WARP_HOMOGRAPHY: H_t2s [B*N,3,3] (layers.py:219)
 *   plane_aux     PD_WARP_HOMOGRAPHY: R·n [B*N,3] (layers.py:223); else NULL
 *   inv_K3        PD_WARP_HOMOGRAPHY: inv_K[:, :3, :3] as [B,3,3]; else NULL
 *   padding_mask  PD_WARP_DISP: outputs["padding_mask"] [B,N,H,W] float 0/1, or NULL (= all ones);
 *                 PD_WARP_HOMOGRAPHY: must be NULL (the mask is computed, layers.py:223-226)
 *   dists         [B,N-1,H,W] outputs["dists"] with PD_RENDER_PROB, else NULL
 * outputs
 *   rgb_rec       [B,3,H,W]  outputs[("rgb_rec", side)]
 *   ph_map        [B,1,H,W]  per-pixel photometric loss BEFORE the mean / mask_novel product of trainer.py:735-742
 *   ph_mean       [1] or NULL: mean(ph_map), i.e. the `.mean()` of trainer.py:742 fused into the sweep (block sums and
 *                 one fp32 atomic per wave: the last bits depend on the order; use ph_map where that matters)
 *   stash         [B, pd_sweep_stash_floats/(H*W), H, W] opaque, consumed by pd_plane_sweep_bwd
 */
int pd_plane_sweep_fwd(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                       const float* sigma, const float* plane, const float* plane_aux, const float* inv_K3,
                       const float* padding_mask, const float* dists, float* rgb_rec, float* ph_map, float* ph_mean,
                       float* stash, pd_stream_t stream);

/*
 * Backward of the above: what autograd computes through trainer.py:567-603 + 728-742 in the reference
 * (grid_sampler_2d_backward, softmax/clamp/div/sum/log/abs backward ...).
 *   g_rgb_rec [B,3,H,W] upstream gradient of rgb_rec (perceptual loss etc.), may be NULL (= zeros)
 *   g_ph_map  [B,1,H,W] upstream gradient of ph_map, may be NULL (= zeros)
 *   g_ph_mean [1] device scalar: upstream gradient of ph_mean, may be NULL (= zero); both gradients add up
 * outputs (each may be NULL to skip it)
 *   g_logits, g_sigma  [B,N,H,W]  — fully overwritten
 *   g_plane            same shape as `plane` (disp: [B,N] or dense [B,N,H,W]; homography: [B*N,3,3]) — overwritten
 *   g_dists            [B,N-1,H,W] gradient of `dists` (PD_RENDER_PROB only) — overwritten
 *   workspace          pd_sweep_bwd_workspace_floats(d) floats of scratch (partial sums of g_plane; boundary spill of
 *                      the row-shift kernels) — uninitialised is fine, one per concurrent call
 */
int pd_plane_sweep_bwd(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                       const float* sigma, const float* plane, const float* plane_aux, const float* inv_K3,
                       const float* padding_mask, const float* dists, const float* rgb_rec, const float* stash,
                       const float* g_rgb_rec, const float* g_ph_map, const float* g_ph_mean, float* g_logits,
                       float* g_sigma, float* g_plane, float* g_dists, float* workspace, pd_stream_t stream);

/*
 * pd_plane_sweep_bwd with the backward of the fused decoder tail (pd_decoder_tail_fwd, networks/depth_decoder.py:258-291)
 * riding along — SURVEY.md 8f rank 1's "removes the round-trips": where the decoder's logits / sigma come from
 * pd_decoder_tail_fwd without a padding mask (xy planes only), the gradients this call writes are those of the decoder's
 * CONV outputs:
 *   g_raw_logits = g_logits + t,  g_raw_sigma = (g_sigma - t / sigma) * sigmoid'(raw_sigma) * [clamp passed],
 *   t = gD (d_n - disp) P_n,  P = softmax(logits) / sigma / sum(pi / sigma),  gD = g_disp - g_depth * 0.1 * 0.58 * W / disp^2,
 * and g_plane [B,N] carries the sum of the warp's and the tail's (sum_pixels gD P_n) disparity gradients.  The
 * [B,N,H,W]-sized g_logits / g_sigma are then never re-read by a tail kernel (pd_decoder_tail_bwd reads 4 N + writes 2 N
 * floats per pixel for what costs this kernel ~20 instructions per element on values it holds anyway).
 *   logits, sigma   what pd_decoder_tail_fwd returned (logits = its raw_logits input when there is no mask)
 *   raw_sigma       [B,N,H,W] the sigma conv's output (read only where sigma sits on the clamp's lower bound)
 *   tail_stash, disp  pd_decoder_tail_fwd's stash [B,2,H,W] and disp [B,1,H,W]
 *   g_disp, g_depth   [B,1,H,W] upstream gradients of the tail's disp / depth outputs, either may be NULL
 * Served where pd_sweep_bwd_tail_fuses(d) == 1 (PD_WARP_DISP, PD_MIXTURE, one disparity per plane, sign = +-1, the row-stream
 * backward's plain LDS layout); PD_ERR_UNSUPPORTED otherwise (the caller then runs pd_plane_sweep_bwd + pd_decoder_tail_bwd).
 */
int pd_sweep_bwd_tail_fuses(const pd_sweep_desc* d);
int pd_plane_sweep_bwd_tail(const pd_sweep_desc* d, const float* src, const float* tgt, const float* logits,
                            const float* sigma, const float* plane, const float* rgb_rec, const float* stash,
                            const float* g_rgb_rec, const float* g_ph_map, const float* g_ph_mean, const float* raw_sigma,
                            const float* tail_stash, const float* disp, const float* g_disp, const float* g_depth,
                            float* g_raw_logits, float* g_raw_sigma, float* g_plane, float* workspace, pd_stream_t stream);

/*
 * The per-plane tensors the reference stores in `outputs` and the fused path never needs (trainer.py:582-602):
 * rgb_rec_layered [B,N,3,H,W], logit_rec, probability_rec, sigma_rec, pi_rec [B,N,H,W].  Any output may be NULL.
 * Forward only (values for logging / inspection; gradients flow through pd_plane_sweep_fwd/bwd).
 */
int pd_plane_sweep_layers(const pd_sweep_desc* d, const float* src, const float* logits, const float* sigma,
                          const float* plane, const float* plane_aux, const float* inv_K3, const float* padding_mask,
                          const float* dists, float* rgb_rec_layered, float* logit_rec, float* probability_rec,
                          float* sigma_rec, float* pi_rec, pd_stream_t stream);

/*
 * SSIM (layers.py:276-306: 3x3 box, ReflectionPad2d(1), clamp((1-n/d)/2,0,1)) and
 * Trainer.compute_reprojection_loss (trainer.py:687-699) — SURVEY.md row A10.
 *   pd_ssim_fwd:        x,y [B,C,H,W] -> out [B,C,H,W]
 *   pd_reproj_loss_fwd: pred,target [B,3,H,W] -> loss [B,1,H,W] = use_ssim ? 0.85*mean_c ssim + 0.15*mean_c|t-p| : mean_c|t-p|
 *   pd_reproj_loss_bwd: g_loss [B,1,H,W] -> g_pred [B,3,H,W] (overwritten); g_target may be NULL
 *   pd_ssim_bwd:        g_out [B,C,H,W] -> g_x, g_y [B,C,H,W] (either may be NULL)
 */
int pd_ssim_fwd(int B, int C, int H, int W, const float* x, const float* y, float* out, pd_stream_t stream);
int pd_ssim_bwd(int B, int C, int H, int W, const float* x, const float* y, const float* g_out, float* g_x, float* g_y,
                pd_stream_t stream);
int pd_reproj_loss_fwd(int B, int H, int W, int use_ssim, const float* pred, const float* target, float* loss,
                       pd_stream_t stream);
int pd_reproj_loss_bwd(int B, int H, int W, int use_ssim, const float* pred, const float* target, const float* g_loss,
                       float* g_pred, float* g_target, pd_stream_t stream);

/*
 * multimodal_loss (layers.py:465-466) on materialised tensors:
 *   out[b,0,p] = -log( sum_n pi * dist(error; sigma) + 1e-7 ),  dist = laplacian (layers.py:454) if `laplacian` else
 *   gaussian (layers.py:451).  error, sigma, pi [B,N,H,W] -> out [B,1,H,W].  bwd: any of g_error/g_sigma/g_pi may be NULL.
 */
int pd_mixture_nll_fwd(int B, int N, int H, int W, int laplacian, const float* error, const float* sigma,
                       const float* pi, float* out, pd_stream_t stream);
int pd_mixture_nll_bwd(int B, int N, int H, int W, int laplacian, const float* error, const float* sigma,
                       const float* pi, const float* g_out, float* g_error, float* g_sigma, float* g_pi,
                       pd_stream_t stream);

/*
 * Fused tail of DepthDecoder.forward (networks/depth_decoder.py:256-260, 274-291, softmax branch):
 *   logits = raw_logits * padding_mask; pi = softmax_N(logits); sigma = clamp(sigmoid(raw_sigma), .01, 1);
 *   probability = (pi / sigma * mask) / sum_N (mixture) or pi; disp = sum_N probability * disp_layered;
 *   depth = 0.1 * 0.58 * W / disp.
 * raw_logits, raw_sigma (mixture only), padding_mask (NULL = all ones) [B,N,H,W]; disp_layered [B,N], or [B,N,H,W] with
 * PD_TAIL_DISP_DENSE.  fwd writes logits (only when there is a mask; without one logits == raw_logits), sigma
 * (mixture), disp and depth [B,1,H,W] and a stash [B,2,H,W] {log-sum-exp, sum pi*mask/sigma}.  pi / probability are not
 * materialised by fwd (training reads them for their shape only): pd_decoder_tail_layers writes either or both on demand.
 * bwd: upstream g_logits, g_sigma [B,N,H,W], g_disp, g_depth [B,1,H,W] (each may be NULL = zero) -> g_raw_logits,
 * g_raw_sigma [B,N,H,W], g_disp_layered (same layout as disp_layered; the [B,N] form needs `workspace` of
 * pd_decoder_tail_bwd_workspace_floats floats); outputs that are NULL are skipped.
 */
enum pd_tail_flags { PD_TAIL_MIXTURE = 1, PD_TAIL_DISP_DENSE = 2 };
size_t pd_decoder_tail_bwd_workspace_floats(int B, int N, int H, int W);
int pd_decoder_tail_fwd(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                        const float* padding_mask, const float* disp_layered, float* logits, float* sigma, float* disp,
                        float* depth, float* stash, pd_stream_t stream);
int pd_decoder_tail_layers(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                           const float* padding_mask, const float* stash, float* pi, float* probability,
                           pd_stream_t stream);
int pd_decoder_tail_bwd(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                        const float* padding_mask, const float* disp_layered, const float* stash, const float* disp,
                        const float* g_logits, const float* g_sigma, const float* g_disp, const float* g_depth,
                        float* g_raw_logits, float* g_raw_sigma, float* g_disp_layered, float* workspace,
                        pd_stream_t stream);

/*
 * Fused tail of PladeNet.forward with --render_probability (networks/plade_net.py:309-341; the live producer of
 * outputs["dists"], which the sweep's PD_RENDER_PROB branch consumes, trainer.py:584-591):
 *   depth_layered = 0.1 * 0.58 * W / disp_layered; dists_n = (depth_layered_{n+1} - depth_layered_n) * ray_norm;
 *   alpha_n = 1 - exp(-relu(raw_logits_n) * dists_n) (n < N-1), alpha_{N-1} = 1; pi_n = alpha_n prod_{m<n}(1 - alpha_m + 1e-10);
 *   logits = cat(raw_logits, ones); sigma = clamp(sigmoid(raw_sigma), .01, 1);
 *   probability = (pi / sigma) / sum_N (mixture) or pi; disp = sum_N probability * disp_layered; depth = 0.1 * 0.58 * W / disp.
 * raw_logits [B,N-1,H,W] (conv0's output), raw_sigma [B,N,H,W] (conv_sigma's, mixture only); disp_layered [B,N], or [B,N,H,W]
 * with PD_TAIL_DISP_DENSE; ray_norm [H,W] = |K^-1 [x, y, 1]| of layers.py:468-492 (create_camera_plane).  fwd writes logits
 * [B,N,H,W], dists [B,N-1,H,W], sigma (mixture), disp, depth [B,1,H,W] and a stash [B,1,H,W] {sum pi/sigma};
 * pd_plade_tail_layers writes pi / probability on demand.  bwd: upstream g_logits [B,N,H,W] (its last channel is the constant
 * ones plane), g_dists [B,N-1,H,W], g_sigma [B,N,H,W], g_disp, g_depth (each may be NULL = zero) -> g_raw_logits [B,N-1,H,W],
 * g_raw_sigma, g_disp_layered (layout of disp_layered; the [B,N] form needs `workspace` of
 * pd_decoder_tail_bwd_workspace_floats floats); outputs that are NULL are skipped.
 */
int pd_plade_tail_fwd(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                      const float* disp_layered, const float* ray_norm, float* logits, float* dists, float* sigma, float* disp,
                      float* depth, float* stash, pd_stream_t stream);
int pd_plade_tail_layers(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                         const float* disp_layered, const float* ray_norm, const float* stash, float* pi, float* probability,
                         pd_stream_t stream);
int pd_plade_tail_bwd(int B, int N, int H, int W, int flags, const float* raw_logits, const float* raw_sigma,
                      const float* disp_layered, const float* ray_norm, const float* stash, const float* disp,
                      const float* g_logits, const float* g_dists, const float* g_sigma, const float* g_disp,
                      const float* g_depth, float* g_raw_logits, float* g_raw_sigma, float* g_disp_layered, float* workspace,
                      pd_stream_t stream);

/*
 * get_smooth_loss_disp (layers.py:243-256; trainer.py:768):
 *   out[0] = mean_x |d(x)-d(x+1)| exp(-gamma mean_c|I(x)-I(x+1)|) + the same along y.
 * disp [B,1,H,W], img [B,C,H,W]; both may be crops of wider tensors: unit column stride, the other strides (in floats)
 * are passed explicitly.  bwd: g_out[0] (device scalar) -> g_disp, contiguous [B,1,H,W].
 */
int pd_smooth_loss_fwd(int B, int C, int H, int W, const float* disp, int64_t disp_stride_b, int64_t disp_stride_h,
                       const float* img, int64_t img_stride_b, int64_t img_stride_c, int64_t img_stride_h, float gamma,
                       float* out, pd_stream_t stream);
int pd_smooth_loss_bwd(int B, int C, int H, int W, const float* disp, int64_t disp_stride_b, int64_t disp_stride_h,
                       const float* img, int64_t img_stride_b, int64_t img_stride_c, int64_t img_stride_h, float gamma,
                       const float* g_out, float* g_disp, pd_stream_t stream);
/* As pd_smooth_loss_bwd, for a crop that drops the first x_pad columns of a wider tensor (trainer.py:768 crops 0.2 W):
 * g_disp is the gradient of the UNCROPPED disparity, contiguous [B,1,H,W + x_pad], its first x_pad columns written as
 * the zeros they are — the caller's autograd graph then needs no slice node (a zero-fill and a strided copy per step). */
int pd_smooth_loss_bwd_padded(int B, int C, int H, int W, int x_pad, const float* disp, int64_t disp_stride_b,
                              int64_t disp_stride_h, const float* img, int64_t img_stride_b, int64_t img_stride_c,
                              int64_t img_stride_h, float gamma, const float* g_out, float* g_disp, pd_stream_t stream);

/*
 * Warps of Trainer.generate_post_process_disp (trainer.py:421-466), forward only:
 *   pd_warp_softmax  out[b,n] = softmax over n of planes[b,n] sampled at (x + sign*disp[b,n], y)   [B,N,H,W] -> [B,N,H,W]
 *   pd_warp_sum      out[b,0] = min(cap, sum over n of planes[b,n] sampled at (x + sign*disp[b,n], y))      -> [B,1,H,W]
 * bilinear, zeros padding, align_corners=True, coordinates through the reference's normalise / un-normalise round trip.
 * disp [B,N], with PD_PP_DISP_ROWS [B,N,H] (one disparity per plane and row: the decoder's xz planes) or, with PD_PP_DISP_DENSE,
 * [B,N,H,W]; PD_PP_FLIP_SRC reads `planes` mirrored along x (the .flip(-1) of
 * trainer.py:451) without a flipped copy.  Per-plane disparities with an even W and N <= 64 (pd_warp_sum: any N) take the segment
 * form (two pixels per lane, 12-byte taps, the softmax's samples of all planes in registers: sampled once); PD_PP_SEG=0 keeps
 * the one-pixel-per-lane row kernels, PD_PP_ROWS=0 the per-pixel gather form (both exact: cross-checks).
 */
enum pd_pp_flags { PD_PP_DISP_DENSE = 1, PD_PP_FLIP_SRC = 2, PD_PP_DISP_ROWS = 4 };
int pd_warp_softmax(int B, int N, int H, int W, float sign, int flags, const float* planes, const float* disp,
                    float* out, pd_stream_t stream);
int pd_warp_sum(int B, int N, int H, int W, float sign, int flags, const float* planes, const float* disp, float cap,
                float* out, pd_stream_t stream);
/* disp_pp of trainer.py:458-461 in one pass: disp [2B,1,H,W] (the fixed model's output for cat([image, mirrored image])),
 * o_fr / o_l [B,1,H,W] (the two occlusion masks pd_warp_sum produced) -> disp_pp [B,1,H,W];
 *   mean = disp[b] / 2 + flip(disp[B + b]) / 2;  pp = mean o_fr + disp[b] (1 - o_fr);  disp_pp = pp o_l + flip(disp[B + b]) (1 - o_l). */
int pd_pp_combine(int B, int H, int W, const float* disp, const float* o_fr, const float* o_l, float* disp_pp,
                  pd_stream_t stream);

/* trainer.py:443-465 behind ONE call (three launches on `stream` where a row's softmax fits the CU's LDS — per-plane disparities,
 * even W <= 1024, N <= 64: the "row chains" keep softmax(warp(logits)) in LDS and take the second warp's plane sum from there, the
 * [B,N,H,W] intermediate never reaches memory; PD_PP_CHAIN=0 or any other shape: the six launches of the single warps): logits / probability [2B,N,H,W] and disp [2B,1,H,W] are the fixed
 * model's outputs for cat([image, mirrored image]) (B = half of that batch; of `probability` only the first B images are read),
 * disp_layered [2B,N], PD_PP_DISP_ROWS [2B,N,H] or PD_PP_DISP_DENSE [2B,N,H,W]; workspace: pd_post_process_workspace_floats floats;
 * -> disp_pp, mask_novel [B,1,H,W]. */
size_t pd_post_process_workspace_floats(int B, int N, int H, int W);
int pd_post_process(int B, int N, int H, int W, int flags, const float* logits, const float* probability, const float* disp,
                    const float* disp_layered, float* workspace, float* disp_pp, float* mask_novel, pd_stream_t stream);

/*
 * Trainer.add_flip_right_inputs (trainer.py:252-276): out [2B,C,H,W] = cat([own, flip(other, -1)]);
 * negate_c0 flips the sign of channel 0 in the mirrored half (the x-coordinate channel of `grid`, trainer.py:258-260).
 */
int pd_cat_flip(int B, int C, int H, int W, const float* own, const float* other, int negate_c0, float* out,
                pd_stream_t stream);

/*
 * inputs["grid"] of the data pipeline on the device: RandomResizeCrop / Resize of
 * datasets/pair_transforms.py:27-37, 63-68.  params [B][4] int32 on the device = (full_w, full_h, w0, h0): the resized
 * frame's size and the crop window's origin; grid [B,2,H,W] receives torch.linspace(-1, 1, full_w)[w0 + x] and
 * torch.linspace(-1, 1, full_h)[h0 + y] (scalar formula; within one ulp of ATen's vectorised CPU kernel).
 */
int pd_crop_grid(int B, int H, int W, const int32_t* params, float* grid, pd_stream_t stream);

/*
 * The decoders' disparity levels (networks/depth_decoder.py:147-152, networks/plade_net.py:280-285), one launch each way:
 *   disp[m] = disp_max * (disp_min / disp_max) ** (levels[m] / (no_levels - 1));   distance[m] = dist_num / disp[m]
 * over M = B * N values (levels = arange(no_levels) + the learnt residual; dist_num = 0.1 * 0.58 * W, `distance` may be NULL).
 * pd_plane_levels_bwd: g_levels from the upstream gradients of disp and / or distance (either may be NULL) and the forward's disp.
 */
int pd_plane_levels_fwd(int M, int no_levels, float disp_min, float disp_max, float dist_num, const float* levels, float* disp,
                        float* distance, pd_stream_t stream);
int pd_plane_levels_bwd(int M, int no_levels, float disp_min, float disp_max, float dist_num, const float* disp,
                        const float* g_disp, const float* g_distance, float* g_levels, pd_stream_t stream);

/*
 * Photometric loss under the occlusion mask `mask_novel` (trainer.py:724-742; the mask is produced after
 * pred_novel_images, trainer.py:342-349, so it cannot enter the sweep's forward):
 *   pred = rgb_rec * mask + target * (1 - mask)                      -> pred [B,3,H,W] (may be NULL)
 *   mixture = 0:  mean over (b, pixel) of mean_c |pred - target|, or of min(that, mean_c |source - target|) when
 *                 `source` (the automask's identity view) is given   (trainer.py:738-742)
 *   mixture = 1:  mean over (b, pixel) of ph_map * mask, ph_map [B,1,H,W] = the sweep's per-pixel NLL (:736, :742)
 * mask [B,1,H,W] (NULL = all ones).  partials: workspace of B * ceil(H*W / 256) floats; mean: 1 float.
 * Backward: g_mean [1] (d loss / d mean, device scalar, may be NULL) and g_pred [B,3,H,W] (upstream gradient of the
 * blended prediction, e.g. from the perceptual net; may be NULL) -> g_rgb_rec [B,3,H,W], g_ph_map [B,1,H,W] (mixture).
 */
int pd_masked_photometric_fwd(int B, int H, int W, int mixture, const float* rgb_rec, const float* target,
                              const float* source, const float* mask, const float* ph_map, float* pred, float* partials,
                              float* mean, pd_stream_t stream);
int pd_masked_photometric_bwd(int B, int H, int W, int mixture, const float* rgb_rec, const float* target,
                              const float* source, const float* mask, const float* g_mean, const float* g_pred,
                              float* g_rgb_rec, float* g_ph_map, pd_stream_t stream);

/*
 * The O(B*N) 3x3 algebra of HomographyWarp.forward (layers.py:206-219, 223-225) and its adjoint, one launch each:
 *   M = R + t n^T / d;  H_t2s = inverse(K M K^-1);  Rn = R n          (R, t from T [B,4,4]; K, inv_K [B,4,4])
 * evaluated in fp64 and rounded once to fp32.  distance [B,N], norm [B,N,3].
 *   PD_HMAT_PLANES       H_t2s [B,N,3,3], Rn [B,N,3]
 *   PD_HMAT_UNIFORM      zero-translation poses (trainer.py:386-400): H_t2s [B,4,3,3] in the PD_HOMO_UNIFORM layout
 *                        (slice 0: plane 0 with the translation detached; slices 1..3: virtual planes n/d = e_j with the
 *                        rotation detached), Rn [B,N,3]
 *   PD_HMAT_STEREO_ROWS  identity rotation, x translation, normals without x component (mono_dataset.py:203-211,
 *                        depth_decoder.py:153-207): shift [B,N,rows] = h01*y + h02 in pixels and mask [B,N,rows] =
 *                        facing test && z > 1e-7 (layers.py:223-225) per (plane, row): the inputs of the disp_warp sweep
 *                        with PD_DISP_ROWS | PD_MASK_ROWS.  H_t2s and Rn may be NULL.
 * Backward: g_H in the forward's H_t2s layout (may be NULL in PD_HMAT_STEREO_ROWS) and/or g_shift [B,N,rows];
 * g_distance [B,N], g_norm [B,N,3], g_T [B,4,4] (rows 0..2; row 3 zero) — each may be NULL.  K and inv_K get none (the
 * reference's are dataset constants).  In PD_HMAT_STEREO_ROWS only g_distance is meaningful (h00 is not part of the
 * shift, so the pose / normal derivatives are incomplete).
 */
enum pd_hmat_mode { PD_HMAT_PLANES = 0, PD_HMAT_UNIFORM = 1, PD_HMAT_STEREO_ROWS = 2 };
int pd_homography_matrices_fwd(int B, int N, int mode, int rows, const float* distance, const float* norm, const float* T,
                               const float* K, const float* inv_K, float* H_t2s, float* Rn, float* shift, float* mask,
                               pd_stream_t stream);
int pd_homography_matrices_bwd(int B, int N, int mode, int rows, const float* distance, const float* norm, const float* T,
                               const float* K, const float* inv_K, const float* g_H, const float* g_shift,
                               float* g_distance, float* g_norm, float* g_T, pd_stream_t stream);

/*
 * Geometry modules.
 *   pd_backproject     BackprojectDepth.forward, layers.py:150-156: depth [B,1,H,W], inv_K [B,4,4] -> cam [B,4,H*W]
 *   pd_backproject_bwd g_cam [B,4,H*W] -> g_depth [B,1,H,W]
 *   pd_project3d       Project3D.forward, layers.py:169-182: cam [B,4,H*W], P=(K@T)[:, :3, :] [B,3,4] -> grid [B,H,W,2]
 *   pd_project3d_bwd   g_grid -> g_cam [B,4,H*W] (may be NULL) and g_P [B,3,4] (may be NULL; needs workspace of
 *                      12 * B * ceil(H*W/256) floats)
 *   pd_homography_grid HomographyWarp.forward per-pixel part, layers.py:221-233: H_t2s [M,3,3], Rn [M,3],
 *                      inv_K3 [M,3,3] -> grid [M,H,W,2], mask [M,H,W] (uint8 0/1)
 *   pd_homography_grid_bwd  g_grid [M,H,W,2] -> g_H [M,3,3] (workspace: 9 * M * ceil(H*W/256) floats)
 */
int pd_backproject(int B, int H, int W, const float* depth, const float* inv_K, float* cam, pd_stream_t stream);
int pd_backproject_bwd(int B, int H, int W, const float* inv_K, const float* g_cam, float* g_depth, pd_stream_t stream);
int pd_project3d(int B, int H, int W, float eps, const float* cam, const float* P, float* grid, pd_stream_t stream);
int pd_project3d_bwd(int B, int H, int W, float eps, const float* cam, const float* P, const float* g_grid,
                     float* g_cam, float* g_P, float* workspace, pd_stream_t stream);
int pd_homography_grid(int M, int H, int W, const float* H_t2s, const float* Rn, const float* inv_K3, float* grid,
                       uint8_t* mask, pd_stream_t stream);
int pd_homography_grid_bwd(int M, int H, int W, const float* H_t2s, const float* g_grid, float* g_H, float* workspace,
                           pd_stream_t stream);

/*
 * Second pass of TWO plane-uniform backward calls (PD_HOMO_UNIFORM | PD_BWD_DEFER_GATHER, the same descriptor and the
 * same logits / sigma: the novel frames -1 / +1 of the reference's mono training, trainer.py:532) in one kernel:
 *   g_logits / g_sigma [B,N,H,W] = (PD_BWD_ACCUMULATE in d->flags: their old contents +) view a's gradient + view b's.
 * `plane_*` [B,4,3,3] and `inv_K3_*` [B,3,3] are the views' arguments of pd_plane_sweep_bwd, `workspace_*` the workspaces
 * those calls filled (pd_sweep_bwd_workspace_floats each).  g_sigma may be NULL without PD_MIXTURE.
 */
int pd_uniform_gather_pair(const pd_sweep_desc* d, const float* plane_a, const float* inv_K3_a, float* workspace_a,
                           const float* plane_b, const float* inv_K3_b, float* workspace_b, float* g_logits, float* g_sigma,
                           pd_stream_t stream);

/*
 * TWO plane-uniform target views of the same source image in one call each way (the novel frames -1 / +1 of the reference's
 * mono training: `for target_side in self.target_sides` over the same outputs, trainer.py:532).  The views' workgroups for
 * the same image tile are dispatched next to each other on one XCD, so the second view finds the logits / sigma lines of the
 * first in that XCD's L2; results are those of two pd_plane_sweep_fwd / pd_plane_sweep_bwd calls, bit for bit.
 *   pd_sweep_view: what differs between the views.  Forward: tgt, plane [B,4,3,3], plane_aux [B*N,3], inv_K3 [B,3,3], dists
 *                  (PD_RENDER_PROB) in; rgb_rec, ph_map, ph_mean (may be NULL), stash out.  Backward: the same inputs +
 *                  padding_mask ([B,N,3] translation weights or NULL), rgb_rec, stash, g_rgb_rec / g_ph_map / g_ph_mean
 *                  (each may be NULL) in; g_plane [B,4,3,3] (may be NULL), g_dists (may be NULL) and `workspace`
 *                  (pd_sweep_bwd_workspace_floats(d) floats per view) out.
 *   pd_uniform_fwd_pair   d: PD_WARP_HOMOGRAPHY | PD_HOMO_UNIFORM; PD_PH_MEAN_ZEROED as in pd_plane_sweep_fwd.
 *   pd_uniform_bwd_pair   g_logits / g_sigma [B,N,H,W] = (PD_BWD_ACCUMULATE: their old contents +) both views' gradients;
 *                         g_logits NULL: only g_plane / g_dists are produced.
 */
typedef struct pd_sweep_view {
  const float* tgt;
  const float* plane;
  const float* plane_aux;
  const float* inv_K3;
  const float* padding_mask;
  const float* dists;
  float* rgb_rec;
  float* ph_map;
  float* ph_mean;
  float* stash;
  const float* g_rgb_rec;
  const float* g_ph_map;
  const float* g_ph_mean;
  float* g_plane;
  float* g_dists;
  float* workspace;
} pd_sweep_view;
int pd_uniform_fwd_pair(const pd_sweep_desc* d, const float* src, const float* logits, const float* sigma,
                        const pd_sweep_view* view_a, const pd_sweep_view* view_b, pd_stream_t stream);
int pd_uniform_bwd_pair(const pd_sweep_desc* d, const float* src, const float* logits, const float* sigma,
                        const pd_sweep_view* view_a, const pd_sweep_view* view_b, float* g_logits, float* g_sigma,
                        pd_stream_t stream);

/*
 * F.grid_sample(input, grid, mode="bilinear", padding_mode=zeros|border, align_corners=True) as the reference calls it
 * (trainer.py:444-463, 573-577, 624-628) — SURVEY.md row A5.
 *   input [M,C,Hi,Wi], grid [M,Ho,Wo,2] -> out [M,C,Ho,Wo]
 *   bwd: g_out -> g_input [M,C,Hi,Wi] (must be ZERO-FILLED by the caller; accumulated with atomics; may be NULL),
 *                 g_grid [M,Ho,Wo,2] (overwritten; may be NULL)
 */
int pd_grid_sample_fwd(int M, int C, int Hi, int Wi, int Ho, int Wo, int padding_mode, const float* input,
                       const float* grid, float* out, pd_stream_t stream);
int pd_grid_sample_bwd(int M, int C, int Hi, int Wi, int Ho, int Wo, int padding_mode, const float* input,
                       const float* grid, const float* g_out, float* g_input, float* g_grid, pd_stream_t stream);

/*
 * Diagnostics (used by tests/ and scripts/, not by the product path).
 *   pd_selftest_division        counts, over `count` samples lo + i*step, where the row kernels' fast division by W-1
 *                               (refined reciprocal) differs from the IEEE quotient; *d_mismatches (device int) += count.
 *   pd_debug_gather_flags, pd_debug_poison_lds, pd_debug_count_lds_nans   see below (PD_DEBUG_POISON_LDS=1 makes the Python layer poison before every launch).
 */
int pd_selftest_division(float Wm1, int count, float lo, float step, int* d_mismatches, pd_stream_t stream);
/* The gather backward's device flags of the LAST pd_plane_sweep_bwd launch that used `workspace` with this descriptor
 * (per-plane homographies, PD_IMPL_AUTO): host_out[0] = 1 if some plane was irregular (line at infinity near the view,
 * strong minification, non-finite matrix) and took the atomic fix-up, host_out[1] = 1 if a gather window was cut at its
 * size limit (a plane the prepare kernel should have sent to the fix-up: always 0 in the tests).  Synchronises `stream`. */
int pd_debug_gather_flags(const pd_sweep_desc* d, const float* workspace, int* host_out, pd_stream_t stream);
/* Fills the LDS of the device's CUs with NaNs (a kernel that reads shared memory it never wrote then yields NaNs). */
int pd_debug_poison_lds(pd_stream_t stream);
/* *d_count += the NaNs 2048 workgroups find in 32 KB of shared memory they never wrote (checks that the poison sticks). */
int pd_debug_count_lds_nans(int* d_count, pd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PLANEDEPTH_HIP_H */
