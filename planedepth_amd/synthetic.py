"""Seeded synthetic stand-ins for the KITTI minibatch and the depth decoder's outputs.

There is no dataset (and no network) on the build or GPU boxes, so tests, the golden-vector generator and
``bench.py`` all draw the hot path's inputs from here: tensors with the shapes, value ranges and dict layout
the reference's data pipeline and decoder produce (SURVEY.md rows A1 / D-inputs).
CPU tensors are returned; callers move them to the device.
"""
import numpy as np
import torch


def intrinsics(B, H, W):
    """K as datasets/mono_dataset.py builds it (normalised 0.58 / 1.92 focal, centred), inv_K = pinv(K)."""
    K = np.array([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    inv_K = np.linalg.pinv(K)
    K = torch.from_numpy(K)[None].repeat(B, 1, 1)
    inv_K = torch.from_numpy(inv_K)[None].repeat(B, 1, 1)
    return K, inv_K


def dataset_intrinsics(B, H, W):
    """K / inv_K with the dataset's own float32 arithmetic (kitti_dataset.py:29-32 scaled in place, mono_dataset.py:
    194-198) — an ulp away from ``intrinsics`` (which keeps the values BASELINE.md's known answers were taken with)."""
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= W
    K[1, :] *= H
    inv_K = np.linalg.pinv(K)
    return torch.from_numpy(K)[None].repeat(B, 1, 1), torch.from_numpy(inv_K)[None].repeat(B, 1, 1)


def small_pose(gen, B, rot=0.01, trans=0.05, stereo=False):
    """A rigid motion [B,4,4]: stereo = identity with tx=-0.1, otherwise a small random rotation + translation."""
    T = torch.eye(4)[None].repeat(B, 1, 1)
    if stereo:
        T[:, 0, 3] = -0.1
        return T
    w = torch.randn(B, 3, generator=gen) * rot
    th = w.norm(dim=1, keepdim=True).clamp_min(1e-12)
    k = w / th
    Kx = torch.zeros(B, 3, 3)
    Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    s, c = torch.sin(th)[:, :, None], torch.cos(th)[:, :, None]
    T[:, :3, :3] = torch.eye(3)[None] + s * Kx + (1 - c) * torch.matmul(Kx, Kx)
    T[:, :3, 3] = torch.randn(B, 3, generator=gen) * trans
    return T


def build_case(B, N, H, W, seed, *, disp_min, disp_max, n_xz=0, dense_disp=False, special_disp=None,
               stereo_T=True, with_mask_novel=False, render_probability=False, sigma_interior=False):
    """Synthetic decoder outputs + dataset inputs in the reference's dict format."""
    g = torch.Generator().manual_seed(seed)
    color_l = torch.rand(B, 3, H, W, generator=g)
    color_r = torch.rand(B, 3, H, W, generator=g)
    logits = torch.randn(B, N, H, W, generator=g)
    sigma = torch.rand(B, N, H, W, generator=g).clamp(0.01, 1.0)
    if sigma_interior:  # keep sigma strictly inside the clamp range: no knife-edge clamp gradients
        sigma = 0.011 + 0.978 * sigma
    res = torch.rand(B, N, 1, 1, generator=g) - 0.5
    level = torch.arange(N, dtype=torch.float32)[None, :, None, None] + res
    disp_pp = disp_max * (disp_min / disp_max) ** (level / max(N - 1, 1))  # [B,N,1,1], the learnable per-plane disparity
    if special_disp is not None:
        disp_pp = torch.tensor(special_disp, dtype=torch.float32)[None, :, None, None].repeat(B, 1, 1, 1)
    padding_mask = torch.ones(B, N, H, W)
    row_gain = torch.ones(1, N, H, 1)
    if n_xz:  # last n_xz planes behave like ground planes: disparity grows with the row, masked above the horizon
        ycoord = torch.linspace(-1, 1, H)[None, None, :, None]
        row_gain = row_gain.clone()
        row_gain[:, N - n_xz:] = (0.4 + 0.6 * ycoord.clamp_min(0.0)).expand(1, n_xz, H, 1)
        padding_mask[:, N - n_xz:] = (ycoord >= 1e-7).float().expand(B, n_xz, H, W)
    K, inv_K = intrinsics(B, H, W)
    case = dict(color_l=color_l, color_r=color_r, logits=logits, sigma=sigma, disp_pp=disp_pp, row_gain=row_gain,
                padding_mask=padding_mask, K=K, inv_K=inv_K,
                Rt=small_pose(g, B, stereo=stereo_T), g_rgb_rec=torch.randn(B, 3, H, W, generator=g),
                dense_disp=bool(dense_disp or n_xz))
    if with_mask_novel:
        case["mask_novel"] = (torch.rand(B, 1, H, W, generator=g) > 0.3).float() * torch.rand(B, 1, H, W, generator=g)
    if render_probability:
        case["dists"] = torch.rand(B, N - 1, H, W, generator=g) * 2.0
    return case


def survey_fullsize_case(B=1, N=49, H=192, W=640, seed=1234, sigma_interior=False, n_xz=0):
    """Inputs exactly as SURVEY.md §8(c) C-golden / BASELINE.md §3 describe them (seed 1234, that draw order)."""
    g = torch.Generator().manual_seed(seed)
    color_l = torch.rand(B, 3, H, W, generator=g)
    color_r = torch.rand(B, 3, H, W, generator=g)
    logits = torch.randn(B, N, H, W, generator=g)
    sigma = torch.rand(B, N, H, W, generator=g).clamp(0.01, 1.0)
    if sigma_interior:
        sigma = 0.011 + 0.978 * sigma
    res = torch.rand(B, N, 1, 1, generator=g) - 0.5
    disp_pp = 300.0 * (2.0 / 300.0) ** ((torch.arange(N, dtype=torch.float32)[None, :, None, None] + res) / (N - 1))
    K, inv_K = intrinsics(B, H, W)
    Rt = torch.eye(4)[None].repeat(B, 1, 1)
    Rt[:, 0, 3] = -0.1
    g2 = torch.Generator().manual_seed(4321)
    row_gain, padding_mask = torch.ones(1, N, H, 1), torch.ones(B, N, H, W)
    if n_xz:  # the last n_xz planes act as ground (xz) planes: disparity grows with the row, masked above the horizon
        ycoord = torch.linspace(-1, 1, H)[None, None, :, None]
        row_gain[:, N - n_xz:] = (0.4 + 0.6 * ycoord.clamp_min(0.0)).expand(1, n_xz, H, 1)
        padding_mask[:, N - n_xz:] = (ycoord >= 1e-7).float().expand(B, n_xz, H, W)
    return dict(color_l=color_l, color_r=color_r, logits=logits, sigma=sigma, disp_pp=disp_pp,
                row_gain=row_gain, padding_mask=padding_mask, K=K, inv_K=inv_K, Rt=Rt,
                g_rgb_rec=torch.randn(B, 3, H, W, generator=g2) * 1e-5, dense_disp=bool(n_xz))


def decoder_plane_geometry(grid, residual, *, no_levels=49, xz_levels=14, disp_min=2.0, disp_max=300.0, xz_min=0.1852,
                           xz_max=0.3704, rows=False):
    """The plane set the reference decoder hands to the hot path (networks/depth_decoder.py:146-207, its defaults :28), from
    ``inputs["grid"]`` [B,2,H,W] and the learnt level residuals [B, no_levels + xz_levels] (``sigmoid(residualconv) - 0.5``,
    :150; zeros without ``--plane_residual``): ``no_levels`` fronto-parallel (xy) planes at exponentially spaced disparities
    (:152) and ``xz_levels`` ground (xz) planes at heights ``h`` (:162), whose disparity grows with the image row (:171-181),
    which exist below the horizon only (``padding_mask = y_grid >= 1e-7``, :166) and whose normal / distance for
    homography_warp are ``[0, 1, t] / |.|``, ``h / |.|`` with ``t`` the principal point's offset from the crop centre
    (:197-207).  Returns dict(disp_layered [B,N,H,W] (``rows``: [B,N,H,1] — every plane of this set is constant along x),
    padding_mask (same shape, float), distance [B,N], norm [B,N,3]).  Benchmark / test input: nothing of the product."""
    B, _, H, W = grid.shape
    dt, dev = grid.dtype, grid.device
    lv = torch.arange(no_levels, dtype=dt, device=dev)[None] + residual[:, :no_levels]
    disp_xy = disp_max * (disp_min / disp_max) ** (lv / (no_levels - 1))                       # [B, no_levels]
    distance = 0.1 * 0.58 * W / disp_xy
    norm = torch.tensor([0.0, 0.0, 1.0], dtype=dt, device=dev)[None, None].expand(B, no_levels, -1)
    Wd = 1 if rows else W
    disp_layered = disp_xy[:, :, None, None].expand(-1, -1, H, Wd)
    padding_mask = torch.ones(B, no_levels, H, Wd, dtype=dt, device=dev)
    if xz_levels > 0:
        gl = torch.arange(xz_levels, dtype=dt, device=dev)[None] + residual[:, no_levels:no_levels + xz_levels]
        h = xz_min + (xz_max - xz_min) * gl / (xz_levels - 1)                                  # [B, xz_levels]
        y = grid[:, 1:, :, :Wd].clone()                                                        # [B,1,H,Wd]
        mask_xz = (y >= 1e-7).expand(-1, xz_levels, -1, -1)
        y[y < 1e-7] = 1e-7
        ground = h[:, :, None, None].expand(-1, -1, H, Wd) * 1.92 / (y / 2.0)
        ground = (grid[:, :1, :, -1:] - grid[:, :1, :, :1]) / 2.0 * ground
        ground = 0.1 * 0.58 * W / ground
        disp_layered = torch.cat([disp_layered, ground], 1)
        padding_mask = torch.cat([padding_mask, mask_xz.to(dt)], 1)
        gyc = (grid[:, 1, -1, 0] + grid[:, 1, 0, 0]) / 2
        py = (gyc + 1) * H / 2
        fs = (grid[:, 0, 0, -1] - grid[:, 0, 0, 0]) / 2.0
        t = (py - H / 2) / (H * 1.92 * fs)
        inv = 1 / ((1 + t ** 2) ** 0.5)
        xz_norm = torch.stack([torch.zeros_like(t), torch.ones_like(t), t], 1) * inv[:, None]
        norm = torch.cat([norm, xz_norm[:, None].expand(-1, xz_levels, -1)], 1)
        distance = torch.cat([distance, h * inv[:, None]], 1)
    return dict(disp_layered=disp_layered, padding_mask=padding_mask, distance=distance, norm=norm)


# ---------------------------------------------------------------------------------------------------------------------
# the reference data pipeline's conventions without the dataset
# ---------------------------------------------------------------------------------------------------------------------
def crop_grid(H, W, full_h, full_w, h0, w0):
    """``inputs["grid"]`` as ``RandomResizeCrop`` makes it (datasets/pair_transforms.py:35-37): x / y coordinates in
    [-1, 1] over the RESIZED full image (``full_w`` x ``full_h``), cropped to the ``H x W`` window at (h0, w0).
    ``Resize`` (no crop, pair_transforms.py:63-64) is the special case full = (H, W), h0 = w0 = 0.  -> [2,H,W]."""
    gx, gy = torch.meshgrid(torch.linspace(-1, 1, full_w), torch.linspace(-1, 1, full_h), indexing="xy")
    return torch.stack([gx, gy], 0)[:, h0:h0 + H, w0:w0 + W].clone()


def crop_params(B, H, W, seed=0, crop=True, full=(375, 1242)):
    """[B,4] int32 (full_w, full_h, w0, h0) as RandomResizeCrop draws them (pair_transforms.py:27-32) for a KITTI-sized
    frame; ``crop=False``: the plain Resize grid."""
    rng = np.random.RandomState(seed)
    rows = []
    for _ in range(B):
        if crop:
            full_h0, full_w0 = full
            fmin = max((H + 1) / full_h0, (W + 1) / full_w0)                   # pair_transforms.py:29
            factor = rng.uniform(fmin, max(fmin, 1.0))
            fh, fw = int(full_h0 * factor), int(full_w0 * factor)
            h0, w0 = rng.randint(0, fh - H + 1), rng.randint(0, fw - W + 1)
            rows.append((fw, fh, w0, h0))
        else:
            rows.append((W, H, 0, 0))
    return torch.tensor(rows, dtype=torch.int32)


def kitti_like_inputs_on_device(B, H, W, seed=0, *, crop=True, novel_frame_ids=(), device="cuda"):
    """``kitti_like_inputs`` produced ON the device: images from a device generator, ``grid`` by
    the ``pd_crop_grid`` kernel (the reference's linspace/meshgrid/crop to one ulp), K / inv_K / Rt constants
    uploaded once.  What a data-loader-free end-to-end step (bench.py --ddp_step) feeds the networks."""
    from . import ops
    g = torch.Generator(device=device).manual_seed(seed)
    inputs = {}
    for s in ("l", "r") + tuple(novel_frame_ids):
        img = torch.rand(B, 3, H, W, generator=g, device=device)
        inputs[("color", s)] = img
        inputs[("color_aug", s)] = img.clone()
    inputs["grid"] = ops.crop_grid(crop_params(B, H, W, seed, crop).to(device), H, W)
    K, inv_K = dataset_intrinsics(B, H, W)
    inputs["K"], inputs["inv_K"] = K.to(device), inv_K.to(device)
    Tl, Tr = torch.eye(4)[None].repeat(B, 1, 1), torch.eye(4)[None].repeat(B, 1, 1)
    Tl[:, 0, 3], Tr[:, 0, 3] = 0.1, -0.1
    inputs[("Rt", "l")], inputs[("Rt", "r")] = Tl.to(device), Tr.to(device)
    return inputs


def kitti_like_inputs(B, H, W, seed=0, *, crop=True, novel_frame_ids=(), device="cpu"):
    """A minibatch with the keys, shapes and conventions of the reference's KITTI pipeline (datasets/mono_dataset.py:
    193-211 + pair_transforms.py), from a seeded generator instead of image files:

    * ``("color", s)``, ``("color_aug", s)`` for s in l, r (+ novel frame ids): [B,3,H,W] in [0,1];
    * ``"grid"`` [B,2,H,W]: per-sample random resize factor and crop window as RandomResizeCrop draws them
      (``crop=False``: the plain Resize grid);
    * ``"K"`` = K_KITTI scaled by (W, H), ``"inv_K"`` = ``np.linalg.pinv(K)`` (mono_dataset.py:194-198);
    * ``("Rt","l")`` / ``("Rt","r")``: identity with tx = +0.1 / -0.1 (mono_dataset.py:203-211).
    """
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    inputs = {}
    for s in ("l", "r") + tuple(novel_frame_ids):
        img = torch.rand(B, 3, H, W, generator=g)
        inputs[("color", s)] = img
        inputs[("color_aug", s)] = img.clone()
    grids = []
    for _ in range(B):
        if crop:
            full_h0, full_w0 = 375, 1242                                       # KITTI full resolution
            fmin = max((H + 1) / full_h0, (W + 1) / full_w0)                   # pair_transforms.py:29
            factor = rng.uniform(fmin, max(fmin, 1.0))
            fh, fw = int(full_h0 * factor), int(full_w0 * factor)
            h0, w0 = rng.randint(0, fh - H + 1), rng.randint(0, fw - W + 1)
            grids.append(crop_grid(H, W, fh, fw, h0, w0))
        else:
            grids.append(crop_grid(H, W, H, W, 0, 0))
    inputs["grid"] = torch.stack(grids, 0)
    inputs["K"], inputs["inv_K"] = dataset_intrinsics(B, H, W)
    Tl, Tr = torch.eye(4)[None].repeat(B, 1, 1), torch.eye(4)[None].repeat(B, 1, 1)
    Tl[:, 0, 3], Tr[:, 0, 3] = 0.1, -0.1
    inputs[("Rt", "l")], inputs[("Rt", "r")] = Tl, Tr
    return {k: v.to(device) for k, v in inputs.items()}
