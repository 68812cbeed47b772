"""CPU oracle for PlaneDepth's photometric-reconstruction hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module, and only as the
checker (or as the timed CPU baseline).  The product path (``planedepth_amd``)
never imports it and fails loudly when its HIP library is missing.

What it is: an independent restatement, in plain PyTorch tensor algebra on the
CPU (dtype-generic: run it in float32 for "the reference's arithmetic" or in
float64 for a high-precision yard-stick), of the algorithm in the reference's

  * ``trainer.py:523-603``  Trainer.pred_novel_images   (plane sweep + composite)
  * ``trainer.py:687-699``  Trainer.compute_reprojection_loss (SSIM + L1)
  * ``trainer.py:701-773``  Trainer.compute_losses      (photometric part)
  * ``layers.py:128-234``   BackprojectDepth / Project3D / HomographyWarp
  * ``layers.py:243-256``   get_smooth_loss_disp
  * ``layers.py:276-306``   SSIM
  * ``layers.py:451-466``   gaussian / laplacian / multimodal_loss
  * ``networks/depth_decoder.py:256-291``  the decoder's tail
  * ``trainer.py:404-466``  generate_post_process_disp  
  * torch ``F.grid_sample(bilinear, zeros|border, align_corners=True)`` — the
    third-party op the reference calls at ``trainer.py:573-577, 624-628``;
    restated here from its published formula (``bilinear_sample``) and checked
    against torch's own kernel in ``tests/test_oracle.py``.

Pinning: the reference ships no tests or golden vectors for this
path.  The oracle is pinned instead against outputs of the reference itself,
imported in the build container by ``tests/golden/make_golden.py`` (stand-in
modules for the packages the image lacks); the resulting vectors live in
``tests/golden/`` and are checked by ``tests/test_oracle.py`` on every run, and
``tests/test_oracle_vs_reference.py`` re-runs the live comparison whenever
``/root/reference`` is present.

Gradients come from autograd through this restatement, i.e. they are the
gradients the reference's own autograd graph would produce (same op structure,
including the quirks listed in SURVEY.md H4: zeros padding feeding the softmax,
inclusive clamp gradient, first-wins ties in the automask ``min``).
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    "disp_to_depth", "backproject_depth", "project_3d", "homography_grid", "disp_grid",
    "bilinear_sample", "plane_sweep", "laplacian", "gaussian", "distribution", "multimodal_loss",
    "photometric_loss", "ssim", "reprojection_loss", "smooth_loss_disp", "warp_and_loss",
]


# ----------------------------------------------------------------------------- geometry
def disp_to_depth(disp, width):
    """depth = 0.1 * 0.58 * W / disp — inlined at trainer.py:535, 612 and depth_decoder.py:154, 291."""
    return 0.1 * 0.58 * width / disp


def _pixel_rays(H, W, dtype, device=None):
    """Homogeneous pixel coordinates [1, 3, H*W], rows (x, y, 1) — layers.py:137-148."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype, device=device),
                            torch.arange(W, dtype=dtype, device=device), indexing="ij")
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=dtype, device=device)], 0)[None]


def backproject_depth(depth, inv_K):
    """layers.py:150-156.  depth [B,1,H,W], inv_K [B,4,4] -> homogeneous camera points [B,4,H*W]."""
    B, _, H, W = depth.shape
    rays = torch.matmul(inv_K[:, :3, :3], _pixel_rays(H, W, depth.dtype, depth.device))
    pts = depth.reshape(B, 1, H * W) * rays
    return torch.cat([pts, torch.ones(B, 1, H * W, dtype=depth.dtype, device=depth.device)], 1)


def _normalise_grid(px, py, H, W):
    """layers.py:178-181 / 230-233 / trainer.py:550-552: pixel -> [-1, 1] (x by W-1, y by H-1)."""
    gx = (px / (W - 1) - 0.5) * 2
    gy = (py / (H - 1) - 0.5) * 2
    return torch.stack([gx, gy], -1)


def project_3d(points, K, T, H, W, eps=1e-7):
    """layers.py:169-182.  points [B,4,HW], K/T [B,4,4] -> sampling grid [B,H,W,2]."""
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    z = cam[:, 2, :] + eps
    px = (cam[:, 0, :] / z).reshape(-1, H, W)
    py = (cam[:, 1, :] / z).reshape(-1, H, W)
    return _normalise_grid(px, py, H, W)


def homography_matrices(d, n, T, K, inv_K):
    """The [BN,3,3] algebra of layers.py:206-219: returns (H_t2s, R·n).

    d [B,N], n [B,N,3]; T, K, inv_K already expanded to [BN,4,4] as at trainer.py:557-559.
    """
    B, N = d.shape
    dd = d.reshape(B * N, 1, 1)
    nn_ = n.reshape(B * N, 1, 3)
    R = T[:, :3, :3]
    t = T[:, :3, 3:4]
    Rtnd = R + torch.matmul(t, nn_) / dd
    H_s2t = torch.matmul(K[:, :3, :3], torch.matmul(Rtnd, inv_K[:, :3, :3]))
    H_t2s = torch.inverse(H_s2t)
    Rn = torch.matmul(R, nn_[:, 0, :, None])  # [BN,3,1]
    return H_t2s, Rn


def homography_grid(d, n, T, K, inv_K, H, W, H_t2s=None, Rn=None):
    """layers.py:206-234 -> (grid [BN,H,W,2], padding_mask bool [B,N,1,H,W]).

    ``H_t2s`` (and ``Rn`` [BN,3], the rotated normals of the facing test) may be supplied to pin the [BN,3,3] algebra (tests
    that isolate the per-pixel part).
    """
    B, N = d.shape
    H_own, Rn_own = homography_matrices(d, n, T, K, inv_K)
    H_t2s = H_own if H_t2s is None else H_t2s
    Rn = Rn_own if Rn is None else Rn.reshape(B * N, 3, 1).to(H_own.dtype)
    pix = _pixel_rays(H, W, d.dtype, d.device).expand(B * N, -1, -1)
    p = torch.matmul(H_t2s, pix)
    facing = (torch.matmul(inv_K[:, :3, :3], pix) * Rn).sum(1) > 0.0
    z = p[:, 2, :]
    mask = (facing & (z > 1e-7)).reshape(B, N, 1, H, W)
    z = torch.where(z < 1e-7, torch.full_like(z, 1e-7).detach(), z)  # in-place assignment: no grad where clamped
    px = (p[:, 0, :] / z).reshape(B * N, H, W)
    py = (p[:, 1, :] / z).reshape(B * N, H, W)
    return _normalise_grid(px, py, H, W), mask


def disp_grid(disp_layered, target_side):
    """trainer.py:540-554.  disp_layered [B,N,H,W] -> grid [BN,H,W,2].

    Target "r" samples the source (left) view at x + d, target "l" at x - d.
    """
    B, N, H, W = disp_layered.shape
    dt = disp_layered.dtype
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    if target_side == "l":
        px = xs - disp_layered
    elif target_side == "r":
        px = xs + disp_layered
    else:  # the reference leaves the x coordinate untouched for any other key
        px = xs.expand(B, N, H, W)
    py = ys.expand(B, N, H, W)
    return _normalise_grid(px, py, H, W).reshape(B * N, H, W, 2)


# ----------------------------------------------------------------------------- sampling
def bilinear_sample(feat, grid, padding_mode="zeros"):
    """F.grid_sample(mode='bilinear', align_corners=True) restated.

    feat [M,C,H,W], grid [M,Ho,Wo,2] in [-1,1] -> [M,C,Ho,Wo].
    zeros: a tap contributes only if it lies inside the image.  border: the
    *coordinate* is clamped to [0, size-1] first (gradient through the clamp is
    zero outside), then all taps are clipped to the valid index range.
    """
    M, C, H, W = feat.shape
    ix = (grid[..., 0] + 1) / 2 * (W - 1)
    iy = (grid[..., 1] + 1) / 2 * (H - 1)
    if padding_mode == "border":
        ix = ix.clamp(0, W - 1)
        iy = iy.clamp(0, H - 1)
    elif padding_mode != "zeros":
        raise ValueError(padding_mode)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    flat = feat.reshape(M, C, H * W)

    def tap(xf, yf, w):
        inside = (xf >= 0) & (xf <= W - 1) & (yf >= 0) & (yf <= H - 1)
        xi = xf.clamp(0, W - 1).long()
        yi = yf.clamp(0, H - 1).long()
        idx = (yi * W + xi).reshape(M, 1, -1).expand(-1, C, -1)
        v = torch.gather(flat, 2, idx).reshape(M, C, *xf.shape[1:])
        return v * (w * inside.to(w.dtype))[:, None]

    return tap(x0, y0, w_nw) + tap(x1, y0, w_ne) + tap(x0, y1, w_sw) + tap(x1, y1, w_se)


# ----------------------------------------------------------------------------- plane sweep
def plane_sweep(src, logits, sigma, grid, padding_mask, *, use_mixture_loss=True,
                render_probability=False, dists=None, sampler=bilinear_sample):
    """trainer.py:567-603 for one target view.

    src [B,3,H,W] source colour; logits/sigma [B,N,H,W]; grid [BN,H,W,2];
    padding_mask broadcastable to [B,N,1,H,W] (float 0/1 or bool).
    Returns the dict of tensors the reference stores into ``outputs`` (keys without the side).
    """
    B, N, H, W = logits.shape
    chans = [src[:, None].expand(-1, N, -1, -1, -1).reshape(B * N, 3, H, W), logits.reshape(B * N, 1, H, W)]
    if use_mixture_loss:
        chans.append(sigma.reshape(B * N, 1, H, W))
    feats = torch.cat(chans, 1)
    rec = sampler(feats, grid, "zeros").reshape(B, N, -1, H, W)
    rec = rec * padding_mask.to(rec.dtype)
    out = {}
    out["rgb_rec_layered"] = rec[:, :, :3]
    out["logit_rec"] = logit_rec = rec[:, :, 3]
    if render_probability:  # trainer.py:584-591 (NeRF-style alpha compositing, front to back)
        alpha = 1.0 - torch.exp(-F.relu(logit_rec[:, :-1]) * dists)
        ones = torch.ones_like(alpha[:, :1])
        alpha = torch.cat([alpha, ones], 1)
        trans = torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], 1), 1)[:, :-1]
        prob = alpha * trans
    else:
        prob = torch.softmax(logit_rec, 1)
    if use_mixture_loss:  # trainer.py:594-602
        sigma_rec = rec[:, :, 4].clamp(0.01, 1.0)
        out["sigma_rec"] = sigma_rec
        out["pi_rec"] = prob
        w = prob / sigma_rec
        prob = w / w.sum(1, True)
    out["probability_rec"] = prob
    out["rgb_rec"] = (out["rgb_rec_layered"] * prob[:, :, None]).sum(1)
    return out


# ----------------------------------------------------------------------------- losses
def gaussian(error, sigma):
    """layers.py:451-452."""
    return torch.exp(-0.5 * error ** 2 / sigma ** 2) / sigma / (2 * math.pi) ** 0.5


def laplacian(error, b):
    """layers.py:454-455."""
    return 0.5 * torch.exp(-(torch.abs(error) / b)) / b


def distribution(error, sigma, dist="gaussian"):
    """layers.py:457-459."""
    return gaussian(error, sigma) if dist == "gaussian" else laplacian(error, sigma)


def multimodal_loss(error, sigma, pi, dist="gaussian"):
    """layers.py:465-466: -log(sum_n pi_n * p(error_n; sigma_n) + 1e-7), keepdim over the plane axis."""
    return -torch.log(torch.sum(pi * distribution(error, sigma, dist), dim=1, keepdim=True) + 1e-7)


def photometric_loss(sweep, target, src, *, use_mixture_loss=True, automask=False, mask_novel=None):
    """Photometric part of trainer.py:717-742 for one target view.

    Returns (ph_map [B,1,H,W] *before* the final mean, pred [B,3,H,W] as handed to the perceptual net).
    """
    pred = sweep["rgb_rec"]
    if mask_novel is not None:
        pred = pred * mask_novel + target * (1.0 - mask_novel)
    if use_mixture_loss:
        err = torch.abs(sweep["rgb_rec_layered"] - target[:, None]).mean(2)
        ph = multimodal_loss(err, sweep["sigma_rec"], sweep["pi_rec"], dist="lap")
        if automask:
            err_auto = torch.abs(src[:, None] - target[:, None]).mean(2)
            ph_auto = multimodal_loss(err_auto, sweep["sigma_rec"].detach(), sweep["pi_rec"].detach(), dist="lap")
            ph, _ = torch.cat([ph, ph_auto], 1).min(1, True)
        if mask_novel is not None:
            ph = ph * mask_novel
    else:
        ph = torch.abs(pred - target).mean(1, True)
        if automask:
            ph_auto = torch.abs(src - target).mean(1, True)
            ph, _ = torch.cat([ph, ph_auto], 1).min(1, True)
    return ph, pred


def _box3_reflect(x):
    """ReflectionPad2d(1) followed by AvgPool2d(3, 1) — layers.py:281-287, 293-297."""
    return F.avg_pool2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), 3, 1)


def ssim(x, y):
    """layers.py:292-306: per-pixel, per-channel (1 - SSIM)/2 clamped to [0, 1], 3x3 box window."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu_x = _box3_reflect(x)
    mu_y = _box3_reflect(y)
    sigma_x = _box3_reflect(x ** 2) - mu_x ** 2
    sigma_y = _box3_reflect(y ** 2) - mu_y ** 2
    sigma_xy = _box3_reflect(x * y) - mu_x * mu_y
    num = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    den = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


def reprojection_loss(pred, target, use_ssim=True):
    """trainer.py:687-699: 0.85 * mean_c SSIM + 0.15 * mean_c L1 (or L1 only)."""
    l1 = torch.abs(target - pred).mean(1, True)
    if not use_ssim:
        return l1
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def smooth_loss_disp(disp, img, gamma=1):
    """layers.py:243-256: edge-aware first-order smoothness of a disparity map."""
    dx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    dy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    ix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    iy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (dx * torch.exp(-gamma * ix)).mean() + (dy * torch.exp(-gamma * iy)).mean()


# ----------------------------------------------------------------------------- whole path, one target view
def warp_and_loss(src, target, logits, sigma, *, warp_type="disp_warp", target_side="r",
                  disp_layered=None, padding_mask=None,
                  distance=None, norm=None, T=None, K=None, inv_K=None,
                  use_mixture_loss=True, automask=False, mask_novel=None,
                  render_probability=False, dists=None, sampler=bilinear_sample, H_t2s=None, Rn=None):
    """pred_novel_images + photometric part of compute_losses for ONE target view.

    Returns dict(rgb_rec, ph_map, ph_loss (=ph_map.mean()), pred, sweep=<all layered tensors>).
    """
    B, N, H, W = logits.shape
    if warp_type == "disp_warp":
        grid = disp_grid(disp_layered, target_side)
        mask = padding_mask[:, :, None]
    elif warp_type == "homography_warp":
        ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731  trainer.py:557-559
        grid, mask = homography_grid(distance, norm, ex(T), ex(K), ex(inv_K), H, W, H_t2s=H_t2s, Rn=Rn)
    elif warp_type == "depth_warp":  # trainer.py:533-538 (padding_mask never assigned there — SURVEY F4; use the decoder's)
        depths = disp_to_depth(disp_layered, W)
        ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
        cam = backproject_depth(depths.reshape(B * N, 1, H, W), ex(inv_K))
        grid = project_3d(cam, ex(K), ex(T), H, W)
        mask = padding_mask[:, :, None]
    else:
        raise ValueError(warp_type)
    sweep = plane_sweep(src, logits, sigma, grid, mask, use_mixture_loss=use_mixture_loss,
                        render_probability=render_probability, dists=dists, sampler=sampler)
    ph_map, pred = photometric_loss(sweep, target, src, use_mixture_loss=use_mixture_loss,
                                    automask=automask, mask_novel=mask_novel)
    return dict(rgb_rec=sweep["rgb_rec"], ph_map=ph_map, ph_loss=ph_map.mean(), pred=pred, sweep=sweep, grid=grid)


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY.md §8(f) rows
# ---------------------------------------------------------------------------------------------------------------------
def decoder_tail(raw_logits, raw_sigma, padding_mask, disp_layered, width, use_mixture_loss=True):
    """networks/depth_decoder.py:256-260, 274-291 (softmax branch): what DepthDecoder.forward does with the outputs of
    ``dispconv`` / ``sigmaconv``.  Returns the dict entries it writes."""
    out = {}
    logits = raw_logits * padding_mask                                  # :259
    out["logits"] = logits
    out["probability"] = torch.softmax(logits, 1)                       # :275
    if use_mixture_loss:
        sigma = torch.clamp(torch.sigmoid(raw_sigma), 0.01, 1.0)        # :278-279
        out["sigma"] = sigma
        out["pi"] = pi = out["probability"]                             # :281
        weights = pi / sigma * padding_mask                             # :282-283
        out["probability"] = weights / weights.sum(1, True)             # :284-285
    out["disp"] = (out["probability"] * disp_layered).sum(1, True)      # :289
    out["depth"] = 0.1 * 0.58 * width / out["disp"]                     # :291
    return out


def camera_ray_norm(height, width, dtype=torch.float32):
    """layers.py:468-492 (create_camera_plane) followed by the norm PladeNet takes of it (plade_net.py:315): length of the
    camera ray K^-1 [x, y, 1] of every pixel for the KITTI-shaped intrinsics; [1, H, W].  The reference inverts K with
    torch.inverse in fp32; so does this (dtype = float32), and in float64 for the fp64 evaluation."""
    K = torch.tensor([[0.58 * width, 0, 0.5 * width], [0, 1.92 * height, 0.5 * height], [0, 0, 1]], dtype=torch.float32).to(dtype)
    K_inv = torch.inverse(K)
    ys, xs = torch.meshgrid(torch.arange(height, dtype=dtype), torch.arange(width, dtype=dtype), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width, dtype=dtype)], 0)
    cam = torch.matmul(K_inv, pix).reshape(1, 3, height, width)
    return torch.linalg.norm(cam, dim=1)


def plade_tail(raw_logits, raw_sigma, disp_layered, width, ray_norm, use_mixture_loss=True):
    """networks/plade_net.py:309-341 with --render_probability: what PladeNet.forward does with the outputs of ``conv0``
    (N-1 logit channels) and ``conv_sigma``.  ``ray_norm`` = camera_ray_norm(H, W).  Returns the dict entries it writes."""
    out = {}
    depth_layered = 0.1 * 0.58 * width / disp_layered                                   # :311
    dists = depth_layered[:, 1:] - depth_layered[:, :-1]                                # :312
    dists = dists * ray_norm[:, None]                                                   # :314-315
    out["dists"] = dists
    alpha = 1.0 - torch.exp(-torch.relu(raw_logits) * dists)                            # :317
    ones = torch.ones_like(alpha[:, :1])
    alpha = torch.cat([alpha, ones], 1)                                                 # :318-319
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], 1), 1)[:, :-1]
    probability = alpha * trans                                                         # :320
    out["probability"] = probability
    out["logits"] = torch.cat([raw_logits, ones], 1)                                    # :322
    if use_mixture_loss:
        sigma = torch.clamp(torch.sigmoid(raw_sigma), 0.01, 1.0)                        # :327-328
        out["sigma"] = sigma
        out["pi"] = pi = probability                                                    # :330
        weights = pi / sigma                                                            # :331
        out["probability"] = weights / weights.sum(1, True)                             # :332-333
    out["disp"] = (out["probability"] * disp_layered).sum(1, True)                      # :338
    out["depth"] = 0.1 * 0.58 * width / out["disp"]                                     # :340
    return out


def post_process_disp(logits, probability, disp, disp_layered):
    """trainer.py:421-466: the occlusion-aware blend of the prediction for the image and for its mirror image.  The
    arguments are the fixed model's outputs for the batch cat([image, flipped image]) (2B leading entries)."""
    B2, N, H, W = probability.shape
    B = B2 // 2
    xs = torch.arange(W, dtype=logits.dtype).view(1, 1, 1, W).expand(B, N, H, W)
    ys = torch.arange(H, dtype=logits.dtype).view(1, 1, H, 1).expand(B, N, H, W)

    def grid(shift):                                                    # :427-441
        gx = ((xs + shift) / (W - 1) - 0.5) * 2
        gy = (ys / (H - 1) - 0.5) * 2
        return torch.stack([gx, gy], -1).reshape(B * N, H, W, 2)

    grid_r = grid(disp_layered[:B].expand(B, N, H, W))
    grid_l = grid(-disp_layered[B:].expand(B, N, H, W))

    def warp(t, g):
        return bilinear_sample(t.reshape(B * N, 1, H, W), g, "zeros").reshape(B, N, H, W)

    plr = torch.softmax(warp(logits[:B], grid_r), 1)                    # :443-446
    o_l = warp(plr, grid_l).sum(1, True).clamp(max=1)                   # :447-449
    pfrl = torch.softmax(warp(logits[B:].flip(-1), grid_l), 1)          # :451-453
    o_fr = warp(pfrl, grid_r).sum(1, True).clamp(max=1)                 # :454-456
    mean_disp = disp[:B] * 0.5 + disp[B:].flip(-1) * 0.5                # :458
    disp_pp = mean_disp * o_fr + disp[:B] * (1 - o_fr)                  # :460
    disp_pp = disp_pp * o_l + disp[-B:].flip(-1) * (1 - o_l)            # :461
    mask_novel = warp(probability[:B], grid_r).sum(1, True).clamp(max=1)  # :463-465
    return disp_pp, mask_novel


def add_flip_right_inputs(inputs, novel_frame_ids=()):
    """trainer.py:252-276: batch doubling with the mirrored other view (restated; pure data movement)."""
    new = {}
    for key in ("color", "color_aug", "depth_gt"):
        if (key, "l") in inputs and (key, "r") in inputs:
            new[(key, "l")] = torch.cat([inputs[(key, "l")], inputs[(key, "r")].flip(-1)], 0)
            new[(key, "r")] = torch.cat([inputs[(key, "r")], inputs[(key, "l")].flip(-1)], 0)
    g = inputs["grid"].clone()
    g[:, 0] *= -1.0
    new["grid"] = torch.cat([inputs["grid"], g.flip(-1)], 0)
    for key in ("K", "inv_K", ("Rt", "l"), ("Rt", "r")):
        new[key] = inputs[key].repeat(2, 1, 1)
    for f in novel_frame_ids:
        for key in ("color", "color_aug"):
            new[(key, f)] = torch.cat([inputs[(key, f)], inputs[(key, f)].flip(-1)], 0)
    return new
