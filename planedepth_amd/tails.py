"""Network tails: DepthDecoder's (networks/depth_decoder.py:258-291) and PladeNet's compositing tail
(networks/plade_net.py:309-341).
"""
import ctypes
import os

from . import _capi as C
from . import _state as S
from ._buffers import torch, _timed, _desc, _contig, _zero_scalar, _zero_block, _plane_grad_buffer
from .sweep import TailLink, _per_plane_view

# ---------------------------------------------------------------------------------------------------------------------
# Decoder tail
# ---------------------------------------------------------------------------------------------------------------------
class _DecoderTail(torch.autograd.Function):
    """(raw_logits, raw_sigma, disp_layered[, padding_mask]) -> (logits, sigma, disp, depth, stash)."""

    @staticmethod
    def forward(ctx, raw_logits, raw_sigma, disp_layered, padding_mask, flags, link=None):
        lib = C.load()
        B, N, H, W = raw_logits.shape
        mix = bool(flags & C.PD_TAIL_MIXTURE)
        ctx.link = link
        C.require_gpu_tensor("raw_logits", raw_logits)
        if mix:
            C.require_gpu_tensor("raw_sigma", raw_sigma, (B, N, H, W))
        C.require_gpu_tensor("disp_layered", disp_layered, (B, N, H, W) if flags & C.PD_TAIL_DISP_DENSE else (B, N))
        if padding_mask is not None:
            C.require_gpu_tensor("padding_mask", padding_mask, (B, N, H, W))
        raw_logits, raw_sigma, disp_layered, padding_mask = map(_contig, (raw_logits, raw_sigma, disp_layered, padding_mask))
        dev = raw_logits.device
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)  # noqa: E731
        logits = new(B, N, H, W) if padding_mask is not None else None
        sigma = new(B, N, H, W) if mix else None
        disp, depth, stash = new(B, 1, H, W), new(B, 1, H, W), new(B, 2, H, W)
        with C.on_device(dev), _timed("tail_fwd"):
            C.check(lib.pd_decoder_tail_fwd(B, N, H, W, flags, C.ptr(raw_logits), C.ptr(raw_sigma), C.ptr(padding_mask),
                                            C.ptr(disp_layered), C.ptr(logits), C.ptr(sigma), C.ptr(disp), C.ptr(depth),
                                            C.ptr(stash), C.stream_handle(dev)), "pd_decoder_tail_fwd")
        ctx.save_for_backward(raw_logits, raw_sigma, disp_layered, padding_mask, stash, disp)
        ctx.flags = flags
        ctx.mark_non_differentiable(stash)
        ctx.set_materialize_grads(False)   # an output nobody differentiates (depth, usually) arrives as None, not as a zero tensor
        if link is not None:
            link.raw_sigma, link.stash, link.disp = raw_sigma, stash, disp.detach()
        if logits is None:       # no mask: the logits ARE the conv output (reference: logits * ones)
            logits = raw_logits.view_as(raw_logits)
        if sigma is None:
            sigma = new(0)
            ctx.mark_non_differentiable(sigma)
        return logits, sigma, disp, depth, stash

    @staticmethod
    def backward(ctx, g_logits, g_sigma, g_disp, g_depth, _g_stash):
        lib = C.load()
        raw_logits, raw_sigma, disp_layered, padding_mask, stash, disp = ctx.saved_tensors
        B, N, H, W = raw_logits.shape
        flags = ctx.flags
        mix = bool(flags & C.PD_TAIL_MIXTURE)
        need_l, need_s, need_d = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and mix, ctx.needs_input_grad[2]
        if not (need_l or need_s or need_d):
            return None, None, None, None, None, None
        link = ctx.link
        extra = None
        applied = None
        if link is not None:
            applied, link.applied = link.applied, None   # per-pass state: consumed here (a second backward over the graph)
            link.seen.clear()
        if applied is not None:
            # the sweep's backward kernel applied this node's backward already (pd_plane_sweep_bwd_tail): g_logits / g_sigma ARE
            # the conv outputs' gradients, the disparity share went into the sweep's g_plane.  Only an upstream gradient of
            # disp / depth that the sweep did not see is still owed: the plain kernel on that remainder alone, added on top.
            def rest(got, used):
                if got is None:
                    return None
                if used is None:
                    return got
                if got.data_ptr() == used.data_ptr() and got.shape == used.shape:
                    return None
                return got - used
            r_disp, r_depth = rest(g_disp, applied["disp"]), rest(g_depth, applied["depth"])
            if r_disp is None and r_depth is None:
                return (g_logits if need_l else None), (g_sigma if need_s else None), None, None, None, None
            extra = (g_logits, g_sigma)
            g_logits, g_sigma, g_disp, g_depth = None, None, r_disp, r_depth
        g_raw_logits = torch.empty_like(raw_logits) if need_l else None
        g_raw_sigma = torch.empty_like(raw_sigma) if need_s else None
        g_dl = torch.empty_like(disp_layered) if need_d else None
        ws = None
        if need_d and not (flags & C.PD_TAIL_DISP_DENSE):
            ws = torch.empty(lib.pd_decoder_tail_bwd_workspace_floats(B, N, H, W), device=raw_logits.device,
                             dtype=torch.float32)
        g_logits, g_sigma, g_disp, g_depth = map(_contig, (g_logits, g_sigma if mix else None, g_disp, g_depth))
        with C.on_device(raw_logits.device), _timed("tail_bwd"):
            C.check(lib.pd_decoder_tail_bwd(B, N, H, W, flags, C.ptr(raw_logits), C.ptr(raw_sigma), C.ptr(padding_mask),
                                            C.ptr(disp_layered), C.ptr(stash), C.ptr(disp), C.ptr(g_logits),
                                            C.ptr(g_sigma), C.ptr(g_disp), C.ptr(g_depth), C.ptr(g_raw_logits),
                                            C.ptr(g_raw_sigma), C.ptr(g_dl), C.ptr(ws),
                                            C.stream_handle(raw_logits.device)), "pd_decoder_tail_bwd")
        if extra is not None:
            if g_raw_logits is not None and extra[0] is not None:
                g_raw_logits += extra[0]
            if g_raw_sigma is not None and extra[1] is not None:
                g_raw_sigma += extra[1]
        return g_raw_logits, g_raw_sigma, g_dl, None, None, None


def decoder_tail(raw_logits, raw_sigma, padding_mask, disp_layered, use_mixture_loss=True, fuse_sweep_backward=False):
    """Tail of DepthDecoder.forward (networks/depth_decoder.py:256-291, softmax branch) in one fused pass.

    Returns (logits, sigma | None, disp, depth, layers) where ``layers()`` materialises ``(pi, probability)`` on demand
    (no gradient: nothing in the reference's losses reads them).  ``disp_layered`` may be the decoder's expanded view of
    per-plane scalars or a dense map; ``padding_mask=None`` means all ones (xy planes only).
    """
    B, N, H, W = raw_logits.shape
    if tuple(disp_layered.shape) != (B, N, H, W):
        disp_layered = disp_layered.expand(B, N, H, W)
    per_plane = disp_layered.stride(2) == 0 and disp_layered.stride(3) == 0
    plane = _per_plane_view(disp_layered) if per_plane else disp_layered
    flags = (C.PD_TAIL_MIXTURE if use_mixture_loss else 0) | (0 if per_plane else C.PD_TAIL_DISP_DENSE)
    if padding_mask is not None:
        if padding_mask.dtype != torch.float32:
            padding_mask = padding_mask.float()
        if tuple(padding_mask.shape) != (B, N, H, W):
            padding_mask = padding_mask.expand(B, N, H, W)
    # fuse_sweep_backward: the caller's promise that logits / sigma feed (with gradient) exactly ONE plane sweep — the trainer's
    # single-view pred_novel_images — whose backward kernel then applies this tail's backward too (TailLink).  Sweeps are
    # counted (a second one, or one the fused form does not serve, switches the fusion off); any OTHER differentiable consumer
    # of ``sigma`` (a regulariser on outputs["sigma"]) is NOT detected: its gradient would arrive in sigma space on top of one
    # the sweep already wrote in conv-output space, without the sigmoid' factor and the clamp gate.  (``logits`` are safe:
    # d logits / d raw_logits is the identity here.)  Leave the flag off for such a graph.
    link = TailLink(None, None, None) if (fuse_sweep_backward and use_mixture_loss and padding_mask is None and per_plane
                                           and torch.is_grad_enabled()) else None
    logits, sigma, disp, depth, stash = _DecoderTail.apply(raw_logits, raw_sigma if use_mixture_loss else None, plane,
                                                           padding_mask, flags, link)
    if link is not None:
        logits._pd_tail_link = link
        sigma._pd_tail_link = link

    def layers(want_pi=True, want_probability=True):
        lib = C.load()
        with torch.no_grad():
            pi = torch.empty_like(raw_logits) if want_pi else None
            prob = torch.empty_like(raw_logits) if want_probability else None
            rl, rs, pm = map(_contig, (raw_logits.detach(), raw_sigma.detach() if use_mixture_loss else None, padding_mask))
            with C.on_device(raw_logits.device):
                C.check(lib.pd_decoder_tail_layers(B, N, H, W, flags, C.ptr(rl), C.ptr(rs), C.ptr(pm), C.ptr(stash),
                                                   C.ptr(pi), C.ptr(prob), C.stream_handle(raw_logits.device)),
                        "pd_decoder_tail_layers")
        return pi, prob

    return logits, (sigma if use_mixture_loss else None), disp, depth, layers


class _PladeTail(torch.autograd.Function):
    """(raw_logits [B,N-1,H,W], raw_sigma, disp_layered, ray_norm) -> (logits, dists, sigma, disp, depth, stash)."""

    @staticmethod
    def forward(ctx, raw_logits, raw_sigma, disp_layered, ray_norm, flags):
        lib = C.load()
        B, Nm1, H, W = raw_logits.shape
        N = Nm1 + 1
        mix = bool(flags & C.PD_TAIL_MIXTURE)
        C.require_gpu_tensor("raw_logits", raw_logits)
        if mix:
            C.require_gpu_tensor("raw_sigma", raw_sigma, (B, N, H, W))
        C.require_gpu_tensor("disp_layered", disp_layered, (B, N, H, W) if flags & C.PD_TAIL_DISP_DENSE else (B, N))
        C.require_gpu_tensor("ray_norm", ray_norm, (H, W))
        raw_logits, raw_sigma, disp_layered, ray_norm = map(_contig, (raw_logits, raw_sigma, disp_layered, ray_norm))
        dev = raw_logits.device
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)  # noqa: E731
        logits, dists = new(B, N, H, W), new(B, N - 1, H, W)
        sigma = new(B, N, H, W) if mix else None
        disp, depth, stash = new(B, 1, H, W), new(B, 1, H, W), new(B, 1, H, W)
        with C.on_device(dev):
            C.check(lib.pd_plade_tail_fwd(B, N, H, W, flags, C.ptr(raw_logits), C.ptr(raw_sigma), C.ptr(disp_layered),
                                          C.ptr(ray_norm), C.ptr(logits), C.ptr(dists), C.ptr(sigma), C.ptr(disp), C.ptr(depth),
                                          C.ptr(stash), C.stream_handle(dev)), "pd_plade_tail_fwd")
        ctx.save_for_backward(raw_logits, raw_sigma, disp_layered, ray_norm, stash, disp)
        ctx.flags = flags
        ctx.mark_non_differentiable(stash)
        if sigma is None:
            sigma = new(0)
            ctx.mark_non_differentiable(sigma)
        return logits, dists, sigma, disp, depth, stash

    @staticmethod
    def backward(ctx, g_logits, g_dists, g_sigma, g_disp, g_depth, _g_stash):
        lib = C.load()
        raw_logits, raw_sigma, disp_layered, ray_norm, stash, disp = ctx.saved_tensors
        B, Nm1, H, W = raw_logits.shape
        N = Nm1 + 1
        flags = ctx.flags
        mix = bool(flags & C.PD_TAIL_MIXTURE)
        need_l, need_s, need_d = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and mix, ctx.needs_input_grad[2]
        if not (need_l or need_s or need_d):
            return None, None, None, None, None, None
        g_raw_logits = torch.empty_like(raw_logits) if need_l else None
        g_raw_sigma = torch.empty_like(raw_sigma) if need_s else None
        g_dl = torch.empty_like(disp_layered) if need_d else None
        ws = None
        if need_d and not (flags & C.PD_TAIL_DISP_DENSE):
            ws = torch.empty(lib.pd_decoder_tail_bwd_workspace_floats(B, N, H, W), device=raw_logits.device, dtype=torch.float32)
        g_logits, g_dists, g_sigma, g_disp, g_depth = map(_contig, (g_logits, g_dists, g_sigma if mix else None, g_disp, g_depth))
        with C.on_device(raw_logits.device):
            C.check(lib.pd_plade_tail_bwd(B, N, H, W, flags, C.ptr(raw_logits), C.ptr(raw_sigma), C.ptr(disp_layered),
                                          C.ptr(ray_norm), C.ptr(stash), C.ptr(disp), C.ptr(g_logits), C.ptr(g_dists),
                                          C.ptr(g_sigma), C.ptr(g_disp), C.ptr(g_depth), C.ptr(g_raw_logits), C.ptr(g_raw_sigma),
                                          C.ptr(g_dl), C.ptr(ws), C.stream_handle(raw_logits.device)), "pd_plade_tail_bwd")
        return g_raw_logits, g_raw_sigma, g_dl, None, None


_RAY_NORM = {}   # (H, W, device) -> [H, W]: the ray lengths depend on the image size only (plade_net.py:314 rebuilds them per call)


def camera_ray_norm(height, width, device):
    """|K^-1 [x, y, 1]| per pixel, [H, W]: torch.linalg.norm(create_camera_plane(H, W), dim=1) of the reference
    (layers.py:468-492, plade_net.py:314-315) — the same fp32 torch.inverse / matmul chain on the host, once per image
    size and device (cached)."""
    key = (height, width, str(device))
    if key not in _RAY_NORM:
        _RAY_NORM[key] = _camera_ray_norm(height, width).to(device)
    return _RAY_NORM[key]


def _camera_ray_norm(height, width):
    K = torch.tensor([[0.58 * width, 0, 0.5 * width], [0, 1.92 * height, 0.5 * height], [0, 0, 1]], dtype=torch.float32)
    K_inv = torch.inverse(K)
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width)], 0)
    return torch.linalg.norm(torch.matmul(K_inv, pix).reshape(3, height, width), dim=0).contiguous()


def plade_tail(raw_logits, raw_sigma, disp_layered, ray_norm=None, use_mixture_loss=True):
    """Tail of PladeNet.forward with --render_probability (networks/plade_net.py:309-341) in one fused pass.

    ``raw_logits`` [B,N-1,H,W] = conv0's output, ``raw_sigma`` [B,N,H,W] = conv_sigma's (mixture only), ``disp_layered`` the
    network's expanded view of per-plane scalars or a dense map (ground planes).  Returns (logits [B,N,H,W], dists
    [B,N-1,H,W], sigma | None, disp, depth, layers) where ``layers()`` materialises ``(pi, probability)`` on demand (no
    gradient: nothing in the reference's losses reads them)."""
    B, Nm1, H, W = raw_logits.shape
    N = Nm1 + 1
    if tuple(disp_layered.shape) != (B, N, H, W):
        disp_layered = disp_layered.expand(B, N, H, W)
    per_plane = disp_layered.stride(2) == 0 and disp_layered.stride(3) == 0
    plane = _per_plane_view(disp_layered) if per_plane else disp_layered
    flags = (C.PD_TAIL_MIXTURE if use_mixture_loss else 0) | (0 if per_plane else C.PD_TAIL_DISP_DENSE)
    if ray_norm is None:
        ray_norm = camera_ray_norm(H, W, raw_logits.device)
    logits, dists, sigma, disp, depth, stash = _PladeTail.apply(raw_logits, raw_sigma if use_mixture_loss else None, plane,
                                                               ray_norm, flags)

    def layers(want_pi=True, want_probability=True):
        lib = C.load()
        rl, rs, pl = _contig(raw_logits.detach()), _contig(raw_sigma.detach()) if use_mixture_loss else None, _contig(plane.detach())
        pi = torch.empty(B, N, H, W, device=rl.device) if want_pi else None
        prob = torch.empty(B, N, H, W, device=rl.device) if want_probability else None
        with C.on_device(rl.device):
            C.check(lib.pd_plade_tail_layers(B, N, H, W, flags, C.ptr(rl), C.ptr(rs), C.ptr(pl), C.ptr(ray_norm), C.ptr(stash),
                                             C.ptr(pi), C.ptr(prob), C.stream_handle(rl.device)), "pd_plade_tail_layers")
        return pi, prob

    return logits, dists, (sigma if use_mixture_loss else None), disp, depth, layers


# ---------------------------------------------------------------------------------------------------------------------
# The decoders' disparity levels (networks/depth_decoder.py:147-152, networks/plade_net.py:280-285)
# ---------------------------------------------------------------------------------------------------------------------
class _PlaneLevels(torch.autograd.Function):
    """levels [B,N] -> (disp [B,N], distance [B,N]); one launch each way instead of ~5 + ~8 elementwise ones."""

    @staticmethod
    def forward(ctx, levels, no_levels, disp_min, disp_max, dist_num):
        lib = C.load()
        C.require_gpu_tensor("levels", levels)
        levels = _contig(levels)
        disp, distance = torch.empty_like(levels), torch.empty_like(levels)
        with C.on_device(levels.device):
            C.check(lib.pd_plane_levels_fwd(levels.numel(), int(no_levels), float(disp_min), float(disp_max), float(dist_num),
                                            C.ptr(levels), C.ptr(disp), C.ptr(distance), C.stream_handle(levels.device)),
                    "pd_plane_levels_fwd")
        ctx.save_for_backward(disp)
        ctx.cfg = (int(no_levels), float(disp_min), float(disp_max), float(dist_num))
        return disp, distance

    @staticmethod
    def backward(ctx, g_disp, g_distance):
        lib = C.load()
        disp, = ctx.saved_tensors
        g = torch.empty_like(disp)
        g_disp, g_distance = _contig(g_disp), _contig(g_distance)
        with C.on_device(disp.device):
            C.check(lib.pd_plane_levels_bwd(disp.numel(), *ctx.cfg, C.ptr(disp), C.ptr(g_disp), C.ptr(g_distance), C.ptr(g),
                                            C.stream_handle(disp.device)), "pd_plane_levels_bwd")
        return g, None, None, None, None


def plane_disparities(levels, disp_min, disp_max, width, no_levels=None):
    """``disp_max * (disp_min / disp_max) ** (levels / (no_levels - 1))`` and ``0.1 * 0.58 * W / that`` of the decoders
    (networks/depth_decoder.py:150-152): ``levels`` [B,N,1,1] or [B,N] = arange(no_levels) (+ the plane residual).  Returns
    (disp_layered [B,N,1,1] — expand it over H, W as the decoder does —, distance [B,N]); gradients flow into ``levels``."""
    B, N = levels.shape[:2]
    disp, distance = _PlaneLevels.apply(levels.reshape(B, N), N if no_levels is None else no_levels, disp_min, disp_max,
                                        0.1 * 0.58 * width)
    return disp.reshape(B, N, 1, 1), distance

