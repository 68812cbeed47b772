"""Standalone geometry operators: BackprojectDepth / Project3D / HomographyWarp grids (layers.py:128-234) and grid_sample.
"""
import ctypes
import os

from . import _capi as C
from . import _state as S
from ._buffers import torch, _timed, _desc, _contig, _zero_scalar, _zero_block, _plane_grad_buffer
from .sweep import homography_matrices, homography_matrices_fused

# ---------------------------------------------------------------------------------------------------------------------
# Geometry
# ---------------------------------------------------------------------------------------------------------------------
class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        lib = C.load()
        B, _, H, W = depth.shape
        C.require_gpu_tensor("depth", depth, (B, 1, H, W))
        C.require_gpu_tensor("inv_K", inv_K, (B, 4, 4))
        depth, inv_K = depth.contiguous(), inv_K.contiguous()
        cam = torch.empty(B, 4, H * W, device=depth.device, dtype=torch.float32)
        with C.on_device(depth.device):
            C.check(lib.pd_backproject(B, H, W, C.ptr(depth), C.ptr(inv_K), C.ptr(cam), C.stream_handle(depth.device)),
                    "pd_backproject")
        ctx.save_for_backward(inv_K)
        ctx.hw = (H, W)
        return cam

    @staticmethod
    def backward(ctx, g_cam):
        lib = C.load()
        (inv_K,) = ctx.saved_tensors
        H, W = ctx.hw
        B = inv_K.shape[0]
        g_depth = torch.empty(B, 1, H, W, device=g_cam.device, dtype=torch.float32)
        with C.on_device(g_cam.device):
            C.check(lib.pd_backproject_bwd(B, H, W, C.ptr(inv_K), C.ptr(g_cam.contiguous()), C.ptr(g_depth),
                                           C.stream_handle(g_cam.device)), "pd_backproject_bwd")
        return g_depth, None


def backproject_depth(depth, inv_K):
    """BackprojectDepth.forward (layers.py:150-156): depth [B,1,H,W], inv_K [B,4,4] -> cam points [B,4,H*W]."""
    return _Backproject.apply(depth, inv_K)


class _Project3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam, P, H, W, eps):
        lib = C.load()
        B = cam.shape[0]
        C.require_gpu_tensor("points", cam, (B, 4, H * W))
        C.require_gpu_tensor("P", P, (B, 3, 4))
        cam, P = cam.contiguous(), P.contiguous()
        grid = torch.empty(B, H, W, 2, device=cam.device, dtype=torch.float32)
        with C.on_device(cam.device):
            C.check(lib.pd_project3d(B, H, W, eps, C.ptr(cam), C.ptr(P), C.ptr(grid), C.stream_handle(cam.device)),
                    "pd_project3d")
        ctx.save_for_backward(cam, P)
        ctx.cfg = (H, W, eps)
        return grid

    @staticmethod
    def backward(ctx, g_grid):
        lib = C.load()
        cam, P = ctx.saved_tensors
        H, W, eps = ctx.cfg
        B = cam.shape[0]
        g_cam = torch.empty_like(cam) if ctx.needs_input_grad[0] else None
        g_P = torch.empty_like(P) if ctx.needs_input_grad[1] else None
        ws = torch.empty(12 * B * ((H * W + 255) // 256), device=cam.device, dtype=torch.float32) if g_P is not None else None
        with C.on_device(cam.device):
            C.check(lib.pd_project3d_bwd(B, H, W, eps, C.ptr(cam), C.ptr(P), C.ptr(g_grid.contiguous()), C.ptr(g_cam),
                                         C.ptr(g_P), C.ptr(ws), C.stream_handle(cam.device)), "pd_project3d_bwd")
        return g_cam, g_P, None, None, None


def project_3d(points, K, T, height, width, eps=1e-7):
    """Project3D.forward (layers.py:169-182).  P = (K @ T)[:, :3, :] is formed in torch (B tiny 4x4 products)."""
    P = torch.matmul(K, T)[:, :3, :]
    return _Project3D.apply(points, P, int(height), int(width), float(eps))


class _HomographyGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H_t2s, Rn, inv_K3, H, W):
        lib = C.load()
        M = H_t2s.shape[0]
        C.require_gpu_tensor("H_t2s", H_t2s, (M, 3, 3))
        C.require_gpu_tensor("Rn", Rn, (M, 3))
        C.require_gpu_tensor("inv_K3", inv_K3, (M, 3, 3))
        H_t2s, Rn, inv_K3 = H_t2s.contiguous(), Rn.contiguous(), inv_K3.contiguous()
        grid = torch.empty(M, H, W, 2, device=H_t2s.device, dtype=torch.float32)
        mask = torch.empty(M, H, W, device=H_t2s.device, dtype=torch.uint8)
        with C.on_device(H_t2s.device):
            C.check(lib.pd_homography_grid(M, H, W, C.ptr(H_t2s), C.ptr(Rn), C.ptr(inv_K3), C.ptr(grid), C.ptr(mask),
                                           C.stream_handle(H_t2s.device)), "pd_homography_grid")
        ctx.save_for_backward(H_t2s)
        ctx.hw = (H, W)
        ctx.mark_non_differentiable(mask)
        return grid, mask

    @staticmethod
    def backward(ctx, g_grid, _g_mask):
        lib = C.load()
        (H_t2s,) = ctx.saved_tensors
        H, W = ctx.hw
        M = H_t2s.shape[0]
        g_H = torch.empty_like(H_t2s)
        ws = torch.empty(9 * M * ((H * W + 255) // 256), device=H_t2s.device, dtype=torch.float32)
        with C.on_device(H_t2s.device):
            C.check(lib.pd_homography_grid_bwd(M, H, W, C.ptr(H_t2s), C.ptr(g_grid.contiguous()), C.ptr(g_H), C.ptr(ws),
                                               C.stream_handle(H_t2s.device)), "pd_homography_grid_bwd")
        return g_H, None, None, None, None


def homography_grid(d, n, T, K, inv_K, height, width):
    """HomographyWarp.forward (layers.py:206-234) -> (pix_coords [BN,H,W,2], padding_mask bool [B,N,1,H,W])."""
    B, N = d.shape
    H_t2s, Rn = homography_matrices(d, n, T, K, inv_K)
    grid, mask = _HomographyGrid.apply(H_t2s, Rn.detach(), inv_K[:, :3, :3].detach(), int(height), int(width))
    return grid, mask.bool().reshape(B, N, 1, height, width)


# ---------------------------------------------------------------------------------------------------------------------
# grid_sample (bilinear, align_corners=True)
# ---------------------------------------------------------------------------------------------------------------------
class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, grid, padding_mode):
        lib = C.load()
        M, Cc, Hi, Wi = inp.shape
        _, Ho, Wo, two = grid.shape
        C.require_gpu_tensor("input", inp)
        C.require_gpu_tensor("grid", grid, (M, Ho, Wo, 2))
        inp, grid = inp.contiguous(), grid.contiguous()
        out = torch.empty(M, Cc, Ho, Wo, device=inp.device, dtype=torch.float32)
        with C.on_device(inp.device):
            C.check(lib.pd_grid_sample_fwd(M, Cc, Hi, Wi, Ho, Wo, padding_mode, C.ptr(inp), C.ptr(grid), C.ptr(out),
                                           C.stream_handle(inp.device)), "pd_grid_sample_fwd")
        ctx.save_for_backward(inp, grid)
        ctx.padding_mode = padding_mode
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = C.load()
        inp, grid = ctx.saved_tensors
        M, Cc, Hi, Wi = inp.shape
        _, Ho, Wo, _ = grid.shape
        g_in = torch.zeros_like(inp) if ctx.needs_input_grad[0] else None  # accumulated with atomics
        g_grid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        if g_in is None and g_grid is None:
            return None, None, None
        with C.on_device(inp.device):
            C.check(lib.pd_grid_sample_bwd(M, Cc, Hi, Wi, Ho, Wo, ctx.padding_mode, C.ptr(inp), C.ptr(grid),
                                           C.ptr(g_out.contiguous()), C.ptr(g_in), C.ptr(g_grid),
                                           C.stream_handle(inp.device)), "pd_grid_sample_bwd")
        return g_in, g_grid, None


def grid_sample(input, grid, padding_mode="zeros", align_corners=True, mode="bilinear"):
    """The subset of ``F.grid_sample`` the reference uses: bilinear, align_corners=True, zeros | border."""
    if mode != "bilinear" or not align_corners:
        raise NotImplementedError("PlaneDepth only calls grid_sample(mode='bilinear', align_corners=True)")
    pm = {"zeros": C.PD_PAD_ZEROS, "border": C.PD_PAD_BORDER}.get(padding_mode)
    if pm is None:
        raise NotImplementedError("padding_mode %r (the reference uses 'zeros' and 'border')" % (padding_mode,))
    return _GridSample.apply(input, grid, pm)

