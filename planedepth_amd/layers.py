"""Drop-in replacements for the hot-path names of the reference's ``layers.py``.

Same class names, constructor arguments, ``forward`` signatures and attribute names as the reference
(``BackprojectDepth`` layers.py:128-156, ``Project3D`` :159-182, ``HomographyWarp`` :184-234, ``SSIM`` :276-306,
``gaussian/laplacian/distribution/multimodal_loss`` :451-466, ``get_smooth_loss_disp`` :243-256), so
``from planedepth_amd.layers import *`` can stand in for ``from layers import *`` on the hot path
(``networks/depth_decoder.py:16``, ``trainer.py:26``).  The heavy lifting is done by HIP kernels (``ops``).

Differences from the reference, on purpose:
  * constant buffers are plain (non-persistent) buffers created lazily on the input's device instead of
    ``nn.Parameter(...).cuda()`` at construction time, so building a module needs no GPU and ``state_dict()`` is
    not polluted — the attribute names (``id_coords``, ``ones``, ``pix_coords``) are kept for code that reads them;
  * modules accept CUDA tensors only and raise on CPU tensors (the product has no CPU path).
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

__all__ = ["BackprojectDepth", "Project3D", "HomographyWarp", "SSIM", "gaussian", "laplacian", "distribution",
           "bimodal_loss", "multimodal_loss", "get_smooth_loss_disp", "disp_to_depth", "depth_to_disp"]


def disp_to_depth(disp, width):
    """depth = 0.1 * 0.58 * W / disp — the conversion the reference inlines (trainer.py:535, 612;
    networks/depth_decoder.py:154, 291).  (The ``disp_to_depth`` of BASELINE.json's north_star)"""
    return 0.1 * 0.58 * width / disp


def depth_to_disp(depth, width):
    """Inverse of :func:`disp_to_depth` (same expression: the map is an involution up to the constant)."""
    return 0.1 * 0.58 * width / depth


class _PixelGridMixin:
    """Lazily built constants with the reference's attribute names (layers.py:137-148 / 193-204)."""

    def _init_grid(self, height, width):
        self.height = height
        self.width = width
        self._const_device = None

    def _consts(self, device):
        if self._const_device != device:
            ys, xs = torch.meshgrid(torch.arange(self.height, dtype=torch.float32, device=device),
                                    torch.arange(self.width, dtype=torch.float32, device=device), indexing="ij")
            self.id_coords = torch.stack([xs, ys], 0)
            self.ones = torch.ones(1, 1, self.height * self.width, device=device)
            self.pix_coords = torch.cat([xs.reshape(1, 1, -1), ys.reshape(1, 1, -1), self.ones], 1)
            self._const_device = device
        return self.pix_coords


class BackprojectDepth(nn.Module, _PixelGridMixin):
    """Layer to transform a depth image into a point cloud (reference layers.py:128-156)."""

    def __init__(self, height, width):
        super().__init__()
        self._init_grid(height, width)

    def forward(self, depth, inv_K):
        return ops.backproject_depth(depth, inv_K)


class Project3D(nn.Module):
    """Layer which projects 3D points into a camera with intrinsics K and at position T (layers.py:159-182)."""

    def __init__(self, height, width, eps=1e-7):
        super().__init__()
        self.height = height
        self.width = width
        self.eps = eps

    def forward(self, points, K, T):
        return ops.project_3d(points, K, T, self.height, self.width, self.eps)


class HomographyWarp(nn.Module, _PixelGridMixin):
    """Plane-induced homography sampling grid + padding mask (reference layers.py:184-234).

    forward(d [B,N], n [B,N,3], T, K, inv_K [B*N,4,4]) -> (pix_coords [B*N,H,W,2], padding_mask bool [B,N,1,H,W])
    """

    def __init__(self, height, width):
        super().__init__()
        self._init_grid(height, width)

    def forward(self, d, n, T, K, inv_K):
        return ops.homography_grid(d, n, T, K, inv_K, self.height, self.width)


class SSIM(nn.Module):
    """Layer to compute the SSIM loss between a pair of images (reference layers.py:276-306)."""

    def __init__(self):
        super().__init__()
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, x, y):
        return ops.ssim(x, y)


# --- mixture distributions (layers.py:451-466).  Thin tensor expressions: the fused sweep evaluates the same
# --- formula in-kernel; these exist so code that calls them directly keeps working on GPU tensors.
def gaussian(error, sigma):
    return torch.exp(-0.5 * error ** 2 / sigma ** 2) / sigma / (2 * np.pi) ** 0.5


def laplacian(error, b):
    return 0.5 * torch.exp(-(torch.abs(error) / b)) / b


def distribution(error, sigma, dist="gaussian"):
    return gaussian(error, sigma) if dist == "gaussian" else laplacian(error, sigma)


def bimodal_loss(error0, error1, sigma0, sigma1, w0, w1, dist="gaussian"):
    return -torch.log(w0 * distribution(error0, sigma0, dist) + w1 * distribution(error1, sigma1, dist))


def multimodal_loss(error, sigma, pi, dist="gaussian"):
    """-log(sum_n pi * distribution(error, sigma) + 1e-7) over the plane axis (reference layers.py:465-466) — HIP kernel."""
    return ops.multimodal_loss(error, sigma, pi, dist)


def get_smooth_loss_disp(disp, img, gamma=1):
    """Edge-aware smoothness of a disparity image (reference layers.py:243-256) — one HIP kernel each way; the 0.2W
    crops of trainer.py:768 are read in place through their strides."""
    return ops.smooth_loss_disp(disp, img, gamma)
