"""Loss operators: SSIM / reprojection loss (layers.py:276-306, trainer.py:687-699), the mixture NLL (layers.py:454-466), the
photometric loss under mask_novel (trainer.py:724-742), the smoothness loss (layers.py:243-256).
"""
import ctypes
import os

from . import _capi as C
from . import _state as S
from ._buffers import torch, _timed, _desc, _contig, _zero_scalar, _zero_block, _plane_grad_buffer


# ---------------------------------------------------------------------------------------------------------------------
# SSIM / reprojection loss
# ---------------------------------------------------------------------------------------------------------------------
class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        lib = C.load()
        C.require_gpu_tensor("x", x)
        C.require_gpu_tensor("y", y, x.shape)
        x, y = x.contiguous(), y.contiguous()
        B, Cc, H, W = x.shape
        out = torch.empty_like(x)
        with C.on_device(x.device):
            C.check(lib.pd_ssim_fwd(B, Cc, H, W, C.ptr(x), C.ptr(y), C.ptr(out), C.stream_handle(x.device)), "pd_ssim_fwd")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = C.load()
        x, y = ctx.saved_tensors
        B, Cc, H, W = x.shape
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None
        with C.on_device(x.device):
            C.check(lib.pd_ssim_bwd(B, Cc, H, W, C.ptr(x), C.ptr(y), C.ptr(g.contiguous()), C.ptr(gx), C.ptr(gy),
                                    C.stream_handle(x.device)), "pd_ssim_bwd")
        return gx, gy


def ssim(x, y):
    """layers.py:292-306 — per-pixel, per-channel clamp((1 - SSIM)/2, 0, 1) with a 3x3 reflected box window."""
    return _SSIM.apply(x, y)


class _ReprojLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, use_ssim):
        lib = C.load()
        B, Cc, H, W = pred.shape
        if Cc != 3:
            raise ValueError("compute_reprojection_loss expects 3-channel images")
        C.require_gpu_tensor("pred", pred)
        C.require_gpu_tensor("target", target, pred.shape)
        pred, target = pred.contiguous(), target.contiguous()
        loss = torch.empty(B, 1, H, W, device=pred.device, dtype=torch.float32)
        with C.on_device(pred.device):
            C.check(lib.pd_reproj_loss_fwd(B, H, W, int(use_ssim), C.ptr(pred), C.ptr(target), C.ptr(loss),
                                           C.stream_handle(pred.device)), "pd_reproj_loss_fwd")
        ctx.save_for_backward(pred, target)
        ctx.use_ssim = int(use_ssim)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = C.load()
        pred, target = ctx.saved_tensors
        B, _, H, W = pred.shape
        gp = torch.empty_like(pred)
        gt = torch.empty_like(target) if ctx.needs_input_grad[1] else None
        with C.on_device(pred.device):
            C.check(lib.pd_reproj_loss_bwd(B, H, W, ctx.use_ssim, C.ptr(pred), C.ptr(target), C.ptr(g.contiguous()),
                                           C.ptr(gp), C.ptr(gt), C.stream_handle(pred.device)), "pd_reproj_loss_bwd")
        return gp, gt, None


def reprojection_loss(pred, target, use_ssim=True):
    """trainer.py:687-699 fused: 0.85 * mean_c SSIM(pred, target) + 0.15 * mean_c |target - pred|  -> [B,1,H,W]."""
    return _ReprojLoss.apply(pred, target, bool(use_ssim))


class _MixtureNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, error, sigma, pi, laplacian):
        lib = C.load()
        C.require_gpu_tensor("error", error)
        B, N, H, W = error.shape
        error, sigma, pi = (t.expand(B, N, H, W).contiguous() for t in (error, sigma, pi))
        C.require_gpu_tensor("sigma", sigma)
        C.require_gpu_tensor("pi", pi)
        out = torch.empty(B, 1, H, W, device=error.device, dtype=torch.float32)
        with C.on_device(error.device):
            C.check(lib.pd_mixture_nll_fwd(B, N, H, W, int(laplacian), C.ptr(error), C.ptr(sigma), C.ptr(pi), C.ptr(out),
                                           C.stream_handle(error.device)), "pd_mixture_nll_fwd")
        ctx.save_for_backward(error, sigma, pi)
        ctx.lap = int(laplacian)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = C.load()
        error, sigma, pi = ctx.saved_tensors
        B, N, H, W = error.shape
        ge = torch.empty_like(error) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(sigma) if ctx.needs_input_grad[1] else None
        gp = torch.empty_like(pi) if ctx.needs_input_grad[2] else None
        if ge is None and gs is None and gp is None:
            return None, None, None, None
        with C.on_device(error.device):
            C.check(lib.pd_mixture_nll_bwd(B, N, H, W, ctx.lap, C.ptr(error), C.ptr(sigma), C.ptr(pi),
                                           C.ptr(g.contiguous()), C.ptr(ge), C.ptr(gs), C.ptr(gp),
                                           C.stream_handle(error.device)), "pd_mixture_nll_bwd")
        return ge, gs, gp, None


def multimodal_loss(error, sigma, pi, dist="gaussian"):
    """layers.py:465-466 on materialised [B,N,H,W] tensors -> [B,1,H,W] (one kernel each way instead of ~10 passes)."""
    return _MixtureNLL.apply(error, sigma, pi, dist != "gaussian")




# ---------------------------------------------------------------------------------------------------------------------
# Photometric loss under mask_novel (trainer.py:724-742)
# ---------------------------------------------------------------------------------------------------------------------
class _MaskedPhotometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_rec, ph_map, target, source, mask):
        lib = C.load()
        B, _, H, W = rgb_rec.shape
        dev = rgb_rec.device
        mix = ph_map is not None
        rgb_rec, target = _contig(rgb_rec.detach()), _contig(target)
        source = _contig(source) if source is not None else None
        mask = _contig(mask.float()) if mask is not None else None
        C.require_gpu_tensor("rgb_rec", rgb_rec, (B, 3, H, W))
        C.require_gpu_tensor("target", target, (B, 3, H, W))
        if mask is not None:
            C.require_gpu_tensor("mask_novel", mask, (B, 1, H, W))
        pm = _contig(ph_map.detach()) if mix else None
        pred = torch.empty_like(rgb_rec)
        partials = torch.empty(B * ((H * W + 255) // 256), device=dev)
        mean = torch.empty(1, device=dev)
        with C.on_device(dev):
            C.check(lib.pd_masked_photometric_fwd(B, H, W, int(mix), C.ptr(rgb_rec), C.ptr(target), C.ptr(source),
                                                  C.ptr(mask), C.ptr(pm), C.ptr(pred), C.ptr(partials), C.ptr(mean),
                                                  C.stream_handle(dev)), "pd_masked_photometric_fwd")
        ctx.save_for_backward(rgb_rec, target, source, mask)
        ctx.mix = mix
        return pred, mean.reshape(())

    @staticmethod
    def backward(ctx, g_pred, g_mean):
        lib = C.load()
        rgb_rec, target, source, mask = ctx.saved_tensors
        B, _, H, W = rgb_rec.shape
        dev = rgb_rec.device
        g_pred = _contig(g_pred) if g_pred is not None else None
        g_mean = _contig(g_mean.reshape(1)) if g_mean is not None else None
        g_rgb = torch.empty_like(rgb_rec) if ctx.needs_input_grad[0] else None
        g_ph = torch.empty(B, 1, H, W, device=dev) if (ctx.mix and ctx.needs_input_grad[1]) else None
        if g_rgb is not None or g_ph is not None:
            with C.on_device(dev):
                C.check(lib.pd_masked_photometric_bwd(B, H, W, int(ctx.mix), C.ptr(rgb_rec), C.ptr(target), C.ptr(source),
                                                      C.ptr(mask), C.ptr(g_mean), C.ptr(g_pred), C.ptr(g_rgb),
                                                      C.ptr(g_ph), C.stream_handle(dev)), "pd_masked_photometric_bwd")
        return g_rgb, g_ph, None, None, None


def masked_photometric(rgb_rec, target, mask, *, source=None, ph_map=None):
    """trainer.py:724-742 under ``outputs["mask_novel"]``: returns ``(pred, ph_loss)`` with
    ``pred = rgb_rec * mask + target * (1 - mask)`` (what the perceptual net is fed) and the scalar photometric loss —
    ``ph_map`` given (mixture): ``(ph_map * mask).mean()``; otherwise L1 on ``pred`` with the automask's ``min`` against
    ``source`` when that is given.  One kernel each way (pd_masked_loss.hip)."""
    return _MaskedPhotometric.apply(rgb_rec, ph_map, target, source, mask)


# ---------------------------------------------------------------------------------------------------------------------
# Smoothness loss
# ---------------------------------------------------------------------------------------------------------------------
def _row_strided(name, t):
    """A [B,C,H,W] fp32 GPU tensor whose columns are unit-stride (e.g. the crop t[..., k:]) as it is, else a copy."""
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError("%s must be a float32 GPU tensor (got %s on %s)" % (name, t.dtype, t.device))
    return t if (t.stride(3) == 1 and min(t.stride()[:3]) >= 0) else t.contiguous()


class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img, gamma, x0):
        lib = C.load()
        B, Cn, H, Wf = img.shape
        if tuple(disp.shape) != (B, 1, H, Wf):
            raise ValueError("disp must be [B,1,H,W] matching img, got %s vs %s" % (tuple(disp.shape), tuple(img.shape)))
        disp, img = _row_strided("disp", disp), _row_strided("img", img)
        W = Wf - x0
        out = torch.empty(1, device=disp.device, dtype=torch.float32)
        # the crop [..., x0:] is an offset on the two base pointers: same strides, W - x0 columns
        dptr, iptr = ctypes.c_void_p(disp.data_ptr() + 4 * x0), ctypes.c_void_p(img.data_ptr() + 4 * x0)
        with C.on_device(disp.device):
            C.check(lib.pd_smooth_loss_fwd(B, Cn, H, W, dptr, disp.stride(0), disp.stride(2), iptr,
                                           img.stride(0), img.stride(1), img.stride(2), float(gamma), C.ptr(out),
                                           C.stream_handle(disp.device)), "pd_smooth_loss_fwd")
        ctx.save_for_backward(disp, img)
        ctx.gamma, ctx.x0 = float(gamma), int(x0)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = C.load()
        disp, img = ctx.saved_tensors
        B, Cn, H, Wf = img.shape
        x0 = ctx.x0
        g_disp = torch.empty(B, 1, H, Wf, device=disp.device, dtype=torch.float32)
        g = g.reshape(1).contiguous().float()
        dptr, iptr = ctypes.c_void_p(disp.data_ptr() + 4 * x0), ctypes.c_void_p(img.data_ptr() + 4 * x0)
        with C.on_device(disp.device):
            # one kernel writes the whole [B,1,H,W] gradient, zeros in the cropped-away columns included
            C.check(lib.pd_smooth_loss_bwd_padded(B, Cn, H, Wf - x0, x0, dptr, disp.stride(0), disp.stride(2), iptr,
                                                  img.stride(0), img.stride(1), img.stride(2), ctx.gamma, C.ptr(g),
                                                  C.ptr(g_disp), C.stream_handle(disp.device)), "pd_smooth_loss_bwd_padded")
        return g_disp, None, None, None


def smooth_loss_disp(disp, img, gamma=1.0, x0=0):
    """get_smooth_loss_disp (reference layers.py:243-256) as one kernel each way.  ``x0``: evaluate on the crop
    ``[..., x0:]`` of both tensors (trainer.py:768 passes ``disp[..., int(0.2 * W):]``) WITHOUT slicing them in the autograd
    graph: the crop is a pointer offset in the forward, and the backward writes the gradient of the uncropped ``disp``
    directly (zeros left of the crop) — no slice node, i.e. no zero-fill, strided copy and three operator calls per step.
    Tensors that already are crops (``x0 = 0``) are read in place through their strides as before."""
    x0 = int(x0)
    if not 0 <= x0 <= disp.shape[-1] - 2:   # the crop is a pointer offset: a bad one would read past every row
        raise ValueError("smooth_loss_disp: x0 = %d is not a crop of a width-%d tensor (need 0 <= x0 <= W - 2)" % (x0, disp.shape[-1]))
    return _SmoothLoss.apply(disp, img, gamma, x0)


