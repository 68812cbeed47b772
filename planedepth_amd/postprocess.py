"""Forward-only operators around the batch: the self-distillation post-process (trainer.py:404-466), the batch doubling of
add_flip_right_inputs (trainer.py:252-276), the dataset's crop grid (datasets/pair_transforms.py:27-56).
"""
import ctypes
import os

from . import _capi as C
from . import _state as S
from ._buffers import torch, _timed, _desc, _contig, _zero_scalar, _zero_block, _plane_grad_buffer


# ---------------------------------------------------------------------------------------------------------------------
# Post-process warps — forward only, as in the reference (no_grad networks, detached result)
# ---------------------------------------------------------------------------------------------------------------------
def _pp_disp(disp_layered, B, N, H, W, row_uniform=False):
    """(tensor, flags): per-plane [B,N] when the map is an H/W-expanded view; [B,N,H] + PD_PP_DISP_ROWS when it is constant
    along x — an x-expanded view, or a dense map under the caller's ``row_uniform`` promise (xy and xz planes:
    networks/depth_decoder.py:153-181; yz planes are not) —; else the dense [B,N,H,W] map."""
    if tuple(disp_layered.shape) != (B, N, H, W):
        disp_layered = disp_layered.expand(B, N, H, W)
    if disp_layered.stride(2) == 0 and disp_layered.stride(3) == 0:
        return disp_layered[:, :, 0, 0].contiguous(), 0
    if disp_layered.stride(3) == 0 or row_uniform:
        return disp_layered[:, :, :, 0].contiguous(), C.PD_PP_DISP_ROWS
    return disp_layered.contiguous(), C.PD_PP_DISP_DENSE


def warp_softmax(planes, disp_layered, sign, flip_src=False, row_uniform=False):
    """softmax over the planes of ``planes`` sampled at x + sign * disp (trainer.py:443-446 / 451-453)."""
    lib = C.load()
    C.require_gpu_tensor("planes", planes)
    B, N, H, W = planes.shape
    with torch.no_grad():
        planes = planes.detach().contiguous()
        disp, flags = _pp_disp(disp_layered.detach(), B, N, H, W, row_uniform)
        out = torch.empty_like(planes)
        with C.on_device(planes.device):
            C.check(lib.pd_warp_softmax(B, N, H, W, float(sign), flags | (C.PD_PP_FLIP_SRC if flip_src else 0),
                                        C.ptr(planes), C.ptr(disp), C.ptr(out), C.stream_handle(planes.device)),
                    "pd_warp_softmax")
    return out


def warp_sum(planes, disp_layered, sign, cap=1.0, flip_src=False, row_uniform=False):
    """min(cap, sum over the planes of ``planes`` sampled at x + sign * disp) (trainer.py:447-449, 454-456, 463-465)."""
    lib = C.load()
    C.require_gpu_tensor("planes", planes)
    B, N, H, W = planes.shape
    with torch.no_grad():
        planes = planes.detach().contiguous()
        disp, flags = _pp_disp(disp_layered.detach(), B, N, H, W, row_uniform)
        out = torch.empty(B, 1, H, W, device=planes.device, dtype=torch.float32)
        with C.on_device(planes.device):
            C.check(lib.pd_warp_sum(B, N, H, W, float(sign), flags | (C.PD_PP_FLIP_SRC if flip_src else 0),
                                    C.ptr(planes), C.ptr(disp), float(cap), C.ptr(out),
                                    C.stream_handle(planes.device)), "pd_warp_sum")
    return out


def pp_combine(disp, o_fr, o_l):
    """disp_pp of trainer.py:458-461 in one launch: ``disp`` [2B,1,H,W] (image, mirrored image), the occlusion masks
    ``o_fr`` / ``o_l`` [B,1,H,W] -> mean-of-both where o_fr says so, the mirrored pass's disparity where o_l is 0."""
    lib = C.load()
    C.require_gpu_tensor("disp", disp)
    B2, _, H, W = disp.shape
    B = B2 // 2
    C.require_gpu_tensor("o_fr", o_fr, (B, 1, H, W))
    C.require_gpu_tensor("o_l", o_l, (B, 1, H, W))
    with torch.no_grad():
        disp, o_fr, o_l = (_contig(t.detach()) for t in (disp, o_fr, o_l))
        out = torch.empty(B, 1, H, W, device=disp.device, dtype=torch.float32)
        with C.on_device(disp.device):
            C.check(lib.pd_pp_combine(B, H, W, C.ptr(disp), C.ptr(o_fr), C.ptr(o_l), C.ptr(out), C.stream_handle(disp.device)),
                    "pd_pp_combine")
    return out


def post_process_disp(logits, probability, disp, disp_layered, row_uniform=False):
    """trainer.py:421-466 given the fixed model's outputs for cat([image, mirrored image]) -> (disp_pp, mask_novel): ONE C-ABI
    call (pd_post_process: row chains where a row's softmax fits the CU's LDS, else two warp-softmaxes, three warp-sums and the
    blend).  ``row_uniform=True`` promises a dense ``disp_layered`` that is constant along x (no yz planes): it is then read as
    one disparity per (plane, row) and the row kernels / chains serve it instead of the per-pixel gather form."""
    lib = C.load()
    C.require_gpu_tensor("logits", logits)
    B2, N, H, W = logits.shape
    B = B2 // 2
    with torch.no_grad():
        prob = probability.tensor() if hasattr(probability, "tensor") else probability
        logits, prob, disp = (_contig(t.detach()) for t in (logits, prob, disp))
        dl, flags = _pp_disp(disp_layered.detach(), B2, N, H, W, row_uniform)
        dev = logits.device
        ws = torch.empty(lib.pd_post_process_workspace_floats(B, N, H, W), device=dev, dtype=torch.float32)
        disp_pp = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
        mask_novel = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
        with C.on_device(dev):
            C.check(lib.pd_post_process(B, N, H, W, flags, C.ptr(logits), C.ptr(prob), C.ptr(disp), C.ptr(dl), C.ptr(ws),
                                        C.ptr(disp_pp), C.ptr(mask_novel), C.stream_handle(dev)), "pd_post_process")
    return disp_pp, mask_novel


def post_process_disp_stepwise(logits, probability, disp, disp_layered, row_uniform=False):
    """The same through the single operators (cross-check of pd_post_process; what round 5 ran)."""
    B = probability.shape[0] // 2
    with torch.no_grad():
        dl_r, dl_l = disp_layered[:B], disp_layered[B:]
        ru = dict(row_uniform=row_uniform)
        plr = warp_softmax(logits[:B], dl_r, +1.0, **ru)                 # :443-446
        o_l = warp_sum(plr, dl_l, -1.0, **ru)                            # :447-449
        pfrl = warp_softmax(logits[B:], dl_l, -1.0, flip_src=True, **ru)  # :451-453 (the flip is folded into the read)
        o_fr = warp_sum(pfrl, dl_r, +1.0, **ru)                          # :454-456
        disp_pp = pp_combine(disp, o_fr, o_l)                            # :458-461
        prob = probability.tensor() if hasattr(probability, "tensor") else probability
        mask_novel = warp_sum(prob[:B], dl_r, +1.0, **ru)                # :463-465
    return disp_pp, mask_novel


# ---------------------------------------------------------------------------------------------------------------------
# Batch doubling of add_flip_right_inputs
# ---------------------------------------------------------------------------------------------------------------------
def cat_flip(own, other, negate_c0=False):
    """cat([own, other.flip(-1)], dim=0) in one kernel (trainer.py:253-262); ``negate_c0`` for the grid tensor."""
    lib = C.load()
    C.require_gpu_tensor("own", own)
    C.require_gpu_tensor("other", other, tuple(own.shape))
    B, Cn, H, W = own.shape
    with torch.no_grad():
        own, other = own.contiguous(), other.contiguous()
        out = torch.empty(2 * B, Cn, H, W, device=own.device, dtype=torch.float32)
        with C.on_device(own.device):
            C.check(lib.pd_cat_flip(B, Cn, H, W, C.ptr(own), C.ptr(other), int(bool(negate_c0)), C.ptr(out),
                                    C.stream_handle(own.device)), "pd_cat_flip")
    return out


def crop_grid(params, height, width):
    """``inputs["grid"]`` [B,2,H,W] on the device from per-sample crop parameters [B,4] int32 = (full_w, full_h, w0, h0)
    (datasets/pair_transforms.py:27-37: the RandomResizeCrop grid; Resize is full = (W, H), origin 0) — the reference's
    ``torch.linspace`` / ``meshgrid`` / crop to one ulp (torch's vectorised linspace itself differs in the last bit between
    host CPUs: tests/test_gpu_parity.py::test_on_device_grid_matches_the_reference_pipeline_to_one_ulp)."""
    lib = C.load()
    C.require_gpu_tensor("params", params, dtype=torch.int32)
    if params.dim() != 2 or params.shape[1] != 4:
        raise ValueError("params must be [B,4] int32 (full_w, full_h, w0, h0), got %s" % (tuple(params.shape),))
    B = params.shape[0]
    params = params.contiguous()
    grid = torch.empty(B, 2, int(height), int(width), device=params.device, dtype=torch.float32)
    with C.on_device(params.device):
        C.check(lib.pd_crop_grid(B, int(height), int(width), C.ptr(params), C.ptr(grid), C.stream_handle(params.device)),
                "pd_crop_grid")
    return grid


