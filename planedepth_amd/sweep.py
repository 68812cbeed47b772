"""The fused plane sweep + photometric loss (reference trainer.py:523-603, 717-742): autograd nodes over pd_plane_sweep_*,
routing between the kernel families, the homography algebra, the per-plane layer tensors on demand.

Every function launches hand-written HIP kernels through ctypes on torch's current stream; there is no eager / CPU
implementation behind these operators.
"""
import ctypes
import os

from . import _capi as C
from . import _state as S
from ._state import _env_int
from ._buffers import torch, _timed, _desc, _contig, _zero_scalar, _zero_block, _plane_grad_buffer


# ---------------------------------------------------------------------------------------------------------------------
# Fused plane sweep + photometric loss
# ---------------------------------------------------------------------------------------------------------------------
def _sweep_forward(src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, mode, flags, sign):
    """One target view through pd_plane_sweep_fwd -> ((rgb_rec, ph_map, ph_mean[1]), tensors the backward needs)."""
    S.LAST_SWEEP_FLAGS = flags
    lib = C.load()
    B, N, H, W = logits.shape
    C.require_gpu_tensor("logits", logits)
    C.require_gpu_tensor("src", src, (B, 3, H, W))
    C.require_gpu_tensor("tgt", tgt, (B, 3, H, W))
    if flags & C.PD_MIXTURE:
        C.require_gpu_tensor("sigma", sigma, (B, N, H, W))
    if mode == C.PD_WARP_DISP:
        C.require_gpu_tensor("disp", plane, (B, N, H, W) if flags & C.PD_DISP_DENSE else
                             ((B, N, H) if flags & C.PD_DISP_ROWS else (B, N)))
        if padding_mask is not None:
            C.require_gpu_tensor("padding_mask", padding_mask, (B, N, H) if flags & C.PD_MASK_ROWS else (B, N, H, W))
    else:
        C.require_gpu_tensor("H_t2s", plane, (B, 4, 3, 3) if flags & C.PD_HOMO_UNIFORM else (B * N, 3, 3))
        if flags & C.PD_HOMO_UNIFORM and padding_mask is not None:
            C.require_gpu_tensor("translation weights", padding_mask, (B, N, 3))
        C.require_gpu_tensor("Rn", plane_aux, (B * N, 3))
        C.require_gpu_tensor("inv_K3", inv_K3, (B, 3, 3))
    if flags & C.PD_RENDER_PROB:
        C.require_gpu_tensor("dists", dists, (B, N - 1, H, W))
    else:
        dists = None
    src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists = map(
        _contig, (src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists))
    d = _desc(B, N, H, W, mode, flags, sign)
    k = lib.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
    rgb_rec = torch.empty(B, 3, H, W, device=logits.device, dtype=torch.float32)
    ph_map = torch.empty(B, 1, H, W, device=logits.device, dtype=torch.float32)
    if S.ZERO_POOL:
        ph_mean = _zero_scalar(logits.device)   # a pre-zeroed slot: the entry point then launches no memset (PD_PH_MEAN_ZEROED)
        d.flags |= C.PD_PH_MEAN_ZEROED
    else:
        ph_mean = torch.empty(1, device=logits.device, dtype=torch.float32)
    stash = torch.empty(B, k, H, W, device=logits.device, dtype=torch.float32)
    with C.on_device(logits.device), _timed("fwd"):
        rc = lib.pd_plane_sweep_fwd(ctypes.byref(d), C.ptr(src), C.ptr(tgt), C.ptr(logits), C.ptr(sigma),
                                    C.ptr(plane), C.ptr(plane_aux), C.ptr(inv_K3), C.ptr(padding_mask), C.ptr(dists),
                                    C.ptr(rgb_rec), C.ptr(ph_map), C.ptr(ph_mean), C.ptr(stash),
                                    C.stream_handle(logits.device))
    C.check(rc, "pd_plane_sweep_fwd")
    if S.DEBUG_STASH is not None:
        S.DEBUG_STASH.append(stash)
    return (rgb_rec, ph_map, ph_mean), (src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, rgb_rec, stash)


def _sweep_forward_pair(src, logits, sigma, side_a, side_b):
    """pd_uniform_fwd_pair: two plane-uniform target views (``side_*`` = (tgt, plane, plane_aux, inv_K3, padding_mask, dists,
    mode, flags, sign) with equal mode / flags / sign) of the same src / logits / sigma in one launch.  Returns what two
    ``_sweep_forward`` calls return."""
    lib = C.load()
    mode, flags, sign = side_a[6:9]
    S.LAST_SWEEP_FLAGS = flags
    B, N, H, W = logits.shape
    C.require_gpu_tensor("logits", logits)
    C.require_gpu_tensor("src", src, (B, 3, H, W))
    if flags & C.PD_MIXTURE:
        C.require_gpu_tensor("sigma", sigma, (B, N, H, W))
    src, logits, sigma = _contig(src), _contig(logits), _contig(sigma)
    d = _desc(B, N, H, W, mode, flags, sign)
    k = lib.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
    if S.ZERO_POOL:
        d.flags |= C.PD_PH_MEAN_ZEROED
    views, results = [], []
    for tgt, plane, plane_aux, inv_K3, padding_mask, dists, _, _, _ in (side_a, side_b):
        C.require_gpu_tensor("tgt", tgt, (B, 3, H, W))
        C.require_gpu_tensor("H_t2s", plane, (B, 4, 3, 3))
        C.require_gpu_tensor("Rn", plane_aux, (B * N, 3))
        C.require_gpu_tensor("inv_K3", inv_K3, (B, 3, 3))
        if padding_mask is not None:
            C.require_gpu_tensor("translation weights", padding_mask, (B, N, 3))
        if flags & C.PD_RENDER_PROB:
            C.require_gpu_tensor("dists", dists, (B, N - 1, H, W))
        else:
            dists = None
        tgt, plane, plane_aux, inv_K3, padding_mask, dists = map(_contig, (tgt, plane, plane_aux, inv_K3, padding_mask, dists))
        rgb_rec = torch.empty(B, 3, H, W, device=logits.device, dtype=torch.float32)
        ph_map = torch.empty(B, 1, H, W, device=logits.device, dtype=torch.float32)
        ph_mean = _zero_scalar(logits.device) if S.ZERO_POOL else torch.empty(1, device=logits.device, dtype=torch.float32)
        stash = torch.empty(B, k, H, W, device=logits.device, dtype=torch.float32)
        views.append(C.sweep_view(tgt=tgt, plane=plane, plane_aux=plane_aux, inv_K3=inv_K3, dists=dists, rgb_rec=rgb_rec,
                                  ph_map=ph_map, ph_mean=ph_mean, stash=stash))
        results.append(((rgb_rec, ph_map, ph_mean),
                        (src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, rgb_rec, stash)))
        if S.DEBUG_STASH is not None:
            S.DEBUG_STASH.append(stash)
    with C.on_device(logits.device), _timed("fwd"):
        rc = lib.pd_uniform_fwd_pair(ctypes.byref(d), C.ptr(src), C.ptr(logits), C.ptr(sigma), ctypes.byref(views[0]),
                                     ctypes.byref(views[1]), C.stream_handle(logits.device))
    C.check(rc, "pd_uniform_fwd_pair")
    return results


def _sweep_backward_pair(view_a, view_b, cfg, need_a, need_b, g_logits, g_sigma, accumulate):
    """pd_uniform_bwd_pair: the backward of two plane-uniform views (``view_*`` = (saved tensors, upstream gradients)) of the
    same logits / sigma — both first passes in one launch, then the pair gather into (``accumulate``: added to)
    g_logits / g_sigma (None: only the views' own gradients).  Returns ((g_plane_a, g_dists_a), (g_plane_b, g_dists_b))."""
    lib = C.load()
    mode, flags, sign = cfg
    logits = view_a[0][2]
    B, N, H, W = logits.shape
    mix = bool(flags & C.PD_MIXTURE)
    d = _desc(B, N, H, W, mode, flags | C.PD_BWD_DEFER_GATHER | (C.PD_BWD_ACCUMULATE if accumulate else 0), sign)
    nws = max(int(lib.pd_sweep_bwd_workspace_floats(ctypes.byref(d))), 1)
    views, outs, keep = [], [], []
    for (saved, grads), need in ((view_a, need_a), (view_b, need_b)):
        src, tgt, _, sigma, plane, plane_aux, inv_K3, padding_mask, dists, rgb_rec, stash = saved
        g_rgb_rec, g_ph_map, g_ph_mean = grads
        g_rgb_rec, g_ph_map = _contig(g_rgb_rec), _contig(g_ph_map)
        if g_ph_mean is not None:
            g_ph_mean = g_ph_mean.reshape(1).to(torch.float32).contiguous()
        g_plane = torch.empty_like(plane) if need[2] else None
        g_dists = torch.empty_like(dists) if (dists is not None and need[3]) else None
        ws = torch.empty(nws, device=logits.device, dtype=torch.float32)
        views.append(C.sweep_view(tgt=tgt, plane=plane, plane_aux=plane_aux, inv_K3=inv_K3, padding_mask=padding_mask,
                                  dists=dists, rgb_rec=rgb_rec, stash=stash, g_rgb_rec=g_rgb_rec, g_ph_map=g_ph_map,
                                  g_ph_mean=g_ph_mean, g_plane=g_plane, g_dists=g_dists, workspace=ws))
        outs.append((g_plane, g_dists))
        keep.append((g_rgb_rec, g_ph_map, g_ph_mean, ws))   # alive until the call is enqueued
        if S.DEBUG_WORKSPACE is not None:
            S.DEBUG_WORKSPACE.append((d, ws))
    src, sigma = view_a[0][0], view_a[0][3]
    with C.on_device(logits.device), _timed("bwd"):
        rc = lib.pd_uniform_bwd_pair(ctypes.byref(d), C.ptr(src), C.ptr(logits), C.ptr(sigma), ctypes.byref(views[0]),
                                     ctypes.byref(views[1]), C.ptr(g_logits), C.ptr(g_sigma if mix else None),
                                     C.stream_handle(logits.device))
    C.check(rc, "pd_uniform_bwd_pair")
    del keep
    return outs


def _sweep_backward(saved, cfg, grads, need, into=None, accumulate=False, defer=False):
    """pd_plane_sweep_bwd of one target view.  ``need`` = (logits, sigma, plane, dists) gradients wanted; ``into`` =
    (g_logits, g_sigma) buffers to write (or, ``accumulate``: add) into instead of fresh ones.
    Returns (g_logits, g_sigma, g_plane, g_dists).  ``defer`` (plane-uniform views only): the first pass only
    (PD_BWD_DEFER_GATHER) -> (g_plane, g_dists, workspace); ``_gather_pair`` finishes two such views in one kernel."""
    lib = C.load()
    src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, rgb_rec, stash = saved
    mode, flags, sign = cfg
    g_rgb_rec, g_ph_map, g_ph_mean = grads
    B, N, H, W = logits.shape
    need_logits, need_sigma, need_plane, need_dists = need
    g_plane, plane_flag = _plane_grad_buffer(plane, mode, flags) if need_plane else (None, 0)
    d = _desc(B, N, H, W, mode, flags | plane_flag | (C.PD_BWD_ACCUMULATE if accumulate else 0) |
              (C.PD_BWD_DEFER_GATHER if defer else 0), sign)
    mix = bool(flags & C.PD_MIXTURE)
    if defer:
        g_logits = g_sigma = None
    elif into is not None:
        g_logits, g_sigma = into
    else:
        g_logits = torch.empty_like(logits) if need_logits else None
        g_sigma = torch.empty_like(sigma) if (need_sigma and mix) else None
    g_dists = torch.empty_like(dists) if (dists is not None and need_dists) else None
    # scratch: partial sums of the plane-parameter gradient and the row-shift kernels' boundary spill
    ws = torch.empty(max(int(lib.pd_sweep_bwd_workspace_floats(ctypes.byref(d))), 1), device=logits.device,
                     dtype=torch.float32)
    g_rgb_rec, g_ph_map = _contig(g_rgb_rec), _contig(g_ph_map)
    if g_ph_mean is not None:
        g_ph_mean = g_ph_mean.reshape(1).to(torch.float32).contiguous()
    with C.on_device(logits.device), _timed("bwd"):
        rc = lib.pd_plane_sweep_bwd(ctypes.byref(d), C.ptr(src), C.ptr(tgt), C.ptr(logits), C.ptr(sigma),
                                    C.ptr(plane), C.ptr(plane_aux), C.ptr(inv_K3), C.ptr(padding_mask), C.ptr(dists),
                                    C.ptr(rgb_rec), C.ptr(stash), C.ptr(g_rgb_rec), C.ptr(g_ph_map), C.ptr(g_ph_mean),
                                    C.ptr(g_logits), C.ptr(g_sigma if mix else None), C.ptr(g_plane), C.ptr(g_dists),
                                    C.ptr(ws), C.stream_handle(logits.device))
    C.check(rc, "pd_plane_sweep_bwd")
    if S.DEBUG_WORKSPACE is not None:
        S.DEBUG_WORKSPACE.append((d, ws))
    if defer:
        return g_plane, g_dists, ws
    return g_logits, (g_sigma if mix else None), g_plane, g_dists


def _sweep_backward_tail(saved, cfg, grads, need, link):
    """pd_plane_sweep_bwd_tail: the sweep's backward with the linked decoder tail's backward riding along.  Returns
    (g_raw_logits, g_raw_sigma, g_plane) — handed to autograd as the gradients of logits / sigma; the tail's node passes them
    through (TailLink)."""
    lib = C.load()
    src, tgt, logits, sigma, plane, _, _, _, _, rgb_rec, stash = saved
    mode, flags, sign = cfg
    g_rgb_rec, g_ph_map, g_ph_mean = grads
    B, N, H, W = logits.shape
    g_plane, plane_flag = _plane_grad_buffer(plane, mode, flags) if need[2] else (None, 0)
    d = _desc(B, N, H, W, mode, flags | plane_flag, sign)
    g_disp, g_depth = link.seen.pop("disp", None), link.seen.pop("depth", None)   # (taken: state of THIS backward pass only)
    gl, gs = torch.empty_like(logits), torch.empty_like(sigma)
    ws = torch.empty(max(int(lib.pd_sweep_bwd_workspace_floats(ctypes.byref(d))), 1), device=logits.device, dtype=torch.float32)
    g_rgb_rec, g_ph_map, gd, gz = map(_contig, (g_rgb_rec, g_ph_map, g_disp, g_depth))
    if g_ph_mean is not None:
        g_ph_mean = g_ph_mean.reshape(1).to(torch.float32).contiguous()
    with C.on_device(logits.device), _timed("bwd"):
        rc = lib.pd_plane_sweep_bwd_tail(ctypes.byref(d), C.ptr(src), C.ptr(tgt), C.ptr(logits), C.ptr(sigma), C.ptr(plane),
                                         C.ptr(rgb_rec), C.ptr(stash), C.ptr(g_rgb_rec), C.ptr(g_ph_map), C.ptr(g_ph_mean),
                                         C.ptr(link.raw_sigma), C.ptr(link.stash), C.ptr(link.disp), C.ptr(gd), C.ptr(gz),
                                         C.ptr(gl), C.ptr(gs), C.ptr(g_plane), C.ptr(ws), C.stream_handle(logits.device))
    C.check(rc, "pd_plane_sweep_bwd_tail")
    link.applied = {"disp": g_disp, "depth": g_depth}   # until the tail's node of this pass has consumed it
    link.fused_passes += 1
    return gl, gs, g_plane


def _gather_pair(view_a, view_b, cfg, g_logits, g_sigma, accumulate):
    """pd_uniform_gather_pair: the second pass of two deferred plane-uniform backward calls (``view_*`` = (saved tensors,
    workspace)) into (or, ``accumulate``: added to) g_logits / g_sigma."""
    lib = C.load()
    (saved_a, ws_a), (saved_b, ws_b) = view_a, view_b
    logits = saved_a[2]
    B, N, H, W = logits.shape
    mode, flags, sign = cfg
    mix = bool(flags & C.PD_MIXTURE)
    d = _desc(B, N, H, W, mode, flags | C.PD_BWD_DEFER_GATHER | (C.PD_BWD_ACCUMULATE if accumulate else 0), sign)
    with C.on_device(logits.device), _timed("bwd"):
        rc = lib.pd_uniform_gather_pair(ctypes.byref(d), C.ptr(saved_a[4]), C.ptr(saved_a[6]), C.ptr(ws_a),
                                        C.ptr(saved_b[4]), C.ptr(saved_b[6]), C.ptr(ws_b), C.ptr(g_logits),
                                        C.ptr(g_sigma if mix else None), C.stream_handle(logits.device))
    C.check(rc, "pd_uniform_gather_pair")


class TailLink:
    """What ties a fused decoder tail (``decoder_tail(..., fuse_sweep_backward=True)``) to the ONE plane sweep that consumes
    its logits / sigma, so that the sweep's backward kernel can apply the tail's backward as well
    (``pd_plane_sweep_bwd_tail``: the [B,N,H,W]-sized g_logits / g_sigma are never re-read by a tail kernel).

    Autograd runs the sweep's node before the tail's, and the tail's other upstream gradients (d loss / d disp from the
    smoothness term, d / d depth) reach the tail's node only — so ``pred_novel_images`` routes ``outputs["disp"]`` /
    ``["depth"]`` through gradient taps created AFTER the sweep's node: nodes created later run earlier, the taps have
    handed their gradients over by the time the sweep's backward runs.  The tail's own backward then passes g_logits /
    g_sigma through, and runs its kernel only on whatever upstream gradient of disp / depth the sweep did NOT see (none in
    the trainer's graph; a consumer that took ``disp`` before the tap existed, for example) — correct in any order."""

    def __init__(self, raw_sigma, stash, disp):
        self.raw_sigma, self.stash, self.disp = raw_sigma, stash, disp
        self.consumers = 0        # sweeps that registered as consumers of this tail's logits / sigma
        self.seen = {}            # "disp" / "depth" -> gradient handed over by its tap (taken by the sweep's backward of the pass)
        self.applied = None       # {"disp": g or None, "depth": g or None}: a sweep's backward has applied the tail's terms in THIS
                                  # backward pass; the tail's node consumes it and resets it — a second pass over a retained graph
                                  # (retain_graph=True, a second torch.autograd.grad) starts clean
        self.fused_passes = 0     # backward passes in which the sweep's kernel applied the tail's backward (diagnostics / tests)


class _GradTap(torch.autograd.Function):
    """Identity whose backward leaves the gradient with the TailLink on its way through."""

    @staticmethod
    def forward(ctx, x, link, which):
        ctx.link, ctx.which = link, which
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.link.seen[ctx.which] = g
        return g, None, None


def tail_taps(outputs):
    """Called by ``pred_novel_images`` right after the sweep's node exists: ``outputs["disp"]`` / ``["depth"]`` of a linked
    fused decoder tail go through gradient taps (see TailLink).  No-op without a link or with more than one consumer."""
    link = getattr(outputs.get("logits"), "_pd_tail_link", None)
    if link is None or link.consumers != 1:
        return
    for k in ("disp", "depth"):
        t = outputs.get(k)
        if torch.is_tensor(t) and t.requires_grad:
            outputs[k] = _GradTap.apply(t, link, k)


class _PlaneSweep(torch.autograd.Function):
    """(src, tgt, logits, sigma, plane, ...) -> (rgb_rec [B,3,H,W], ph_map [B,1,H,W], ph_mean []).

    ``ph_mean`` is ``ph_map.mean()`` accumulated inside the sweep kernel (the `.mean()` of trainer.py:742 without a
    reduction kernel of its own); its upstream gradient is a device scalar that the backward kernel applies per pixel.

    Gradients: logits, sigma, plane (disp_layered or H_t2s).  src / tgt are images (no gradient, as in the reference
    where they are dataset tensors).
    """

    @staticmethod
    def forward(ctx, src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, mode, flags, sign, link=None):
        (rgb_rec, ph_map, ph_mean), saved = _sweep_forward(src, tgt, logits, sigma, plane, plane_aux, inv_K3,
                                                           padding_mask, dists, mode, flags, sign)
        ctx.save_for_backward(*saved)
        ctx.cfg = (mode, flags, sign)
        ctx.link = link
        ctx.set_materialize_grads(False)  # unused outputs arrive as None in backward, not as zero tensors
        return rgb_rec, ph_map, ph_mean.reshape(())

    @staticmethod
    def backward(ctx, g_rgb_rec, g_ph_map, g_ph_mean):
        need = (ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4], ctx.needs_input_grad[8])
        link = ctx.link
        if link is not None and link.consumers == 1 and need[0] and need[1]:
            g_logits, g_sigma, g_plane = _sweep_backward_tail(ctx.saved_tensors, ctx.cfg, (g_rgb_rec, g_ph_map, g_ph_mean),
                                                              need, link)
            return None, None, g_logits, g_sigma, g_plane, None, None, None, None, None, None, None, None
        g_logits, g_sigma, g_plane, g_dists = _sweep_backward(ctx.saved_tensors, ctx.cfg,
                                                              (g_rgb_rec, g_ph_map, g_ph_mean), need)
        return None, None, g_logits, g_sigma, g_plane, None, None, None, g_dists, None, None, None, None


_PER_SIDE = 9   # tgt, plane, plane_aux, inv_K3, padding_mask, dists, mode, flags, sign


class _MultiPlaneSweep(torch.autograd.Function):
    """Every target view of one step (trainer.py:532: ``for target_side in self.target_sides``) over the SAME source
    image, logits and sigma as ONE autograd node: the views' gradients into logits / sigma are summed inside the backward
    kernels (PD_BWD_ACCUMULATE) instead of by [B,N,H,W]-sized add kernels between separate nodes (at 8x49x192x640 each
    such add moves 0.58 GB; three views need four of them).

    apply(src, logits, sigma, *flat) with ``flat`` = per view (tgt, plane, plane_aux, inv_K3, padding_mask, dists, mode,
    flags, sign) -> per view (rgb_rec, ph_map, ph_mean)."""

    @staticmethod
    def forward(ctx, src, logits, sigma, *flat):
        n = len(flat) // _PER_SIDE
        outs, tensors, cfgs, layout = [], [], [], []
        sides = [flat[i * _PER_SIDE:(i + 1) * _PER_SIDE] for i in range(n)]
        done = {}   # plane-uniform views of equal configuration go through the forward two at a time (pd_uniform_fwd_pair)
        if S.PAIR_FORWARD:
            uni = [i for i in range(n) if sides[i][6] == C.PD_WARP_HOMOGRAPHY and sides[i][7] & C.PD_HOMO_UNIFORM]
            while len(uni) >= 2:
                i = uni.pop(0)
                j = next((q for q in uni if tuple(sides[q][6:9]) == tuple(sides[i][6:9])), None)
                if j is None:
                    continue
                uni.remove(j)
                done[i], done[j] = _sweep_forward_pair(src, logits, sigma if sides[i][7] & C.PD_MIXTURE else None,
                                                       sides[i], sides[j])
        for i in range(n):
            tgt, plane, plane_aux, inv_K3, padding_mask, dists, mode, flags, sign = sides[i]
            if i in done:
                (rgb_rec, ph_map, ph_mean), saved = done[i]
            else:
                (rgb_rec, ph_map, ph_mean), saved = _sweep_forward(src, tgt, logits, sigma if flags & C.PD_MIXTURE else None,
                                                                   plane, plane_aux, inv_K3, padding_mask, dists, mode, flags, sign)
            outs += [rgb_rec, ph_map, ph_mean.reshape(())]
            cfgs.append((mode, flags, sign))
            idx = []
            for t in saved:     # save_for_backward takes tensors only: remember where the Nones were
                if t is None:
                    idx.append(-1)
                else:
                    idx.append(len(tensors))
                    tensors.append(t)
            layout.append(idx)
        ctx.save_for_backward(*tensors)
        ctx.cfgs, ctx.layout, ctx.n = cfgs, layout, n
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = C.load()
        tensors = ctx.saved_tensors
        n = ctx.n
        need_logits, need_sigma = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        views = []
        for i in range(n):
            g = grads[3 * i:3 * i + 3]
            if all(x is None for x in g):
                continue   # this view took no part in the loss
            saved = tuple(None if j < 0 else tensors[j] for j in ctx.layout[i])
            logits = saved[2]
            B, N, H, W = logits.shape
            mode, flags, sign = ctx.cfgs[i]
            can = bool(lib.pd_sweep_bwd_accumulates(ctypes.byref(_desc(B, N, H, W, mode, flags, sign))))
            views.append((i, saved, g, can))
        views.sort(key=lambda v: v[3])   # kernels that cannot add in place (the row-shift ones) first: one of them starts the sum
        g_logits = g_sigma = None
        per_view = {}

        def pairable(v):   # plane-uniform views with the same kernel configuration gather together (pd_uniform_gather_pair)
            mode, flags, sign = ctx.cfgs[v[0]]
            # (PD_UNI_CHUNK, the library's chunked plane-uniform passes, does not serve the deferred gather: sequential views then)
            return (S.PAIR_GATHER and not _env_int("PD_UNI_CHUNK") and mode == C.PD_WARP_HOMOGRAPHY and
                    bool(flags & C.PD_HOMO_UNIFORM) and (need_logits or need_sigma))
        k = 0
        while k < len(views):
            i, saved, g, can = views[k]
            base = 3 + i * _PER_SIDE
            need = (need_logits, need_sigma, ctx.needs_input_grad[base + 1], ctx.needs_input_grad[base + 5])
            nxt = views[k + 1] if k + 1 < len(views) else None
            if nxt is not None and pairable(views[k]) and pairable(nxt) and ctx.cfgs[i] == ctx.cfgs[nxt[0]]:
                j, saved_j, g_j, _ = nxt
                base_j = 3 + j * _PER_SIDE
                need_j = (need_logits, need_sigma, ctx.needs_input_grad[base_j + 1], ctx.needs_input_grad[base_j + 5])
                started = g_logits is not None or g_sigma is not None
                logits = saved[2]
                mix = bool(ctx.cfgs[i][1] & C.PD_MIXTURE)
                if g_logits is None:
                    g_logits = torch.zeros_like(logits) if started else torch.empty_like(logits)
                if mix and g_sigma is None:
                    g_sigma = torch.zeros_like(logits) if started else torch.empty_like(logits)
                if S.PAIR_FORWARD:   # both first passes in one launch, the pair gather, the reductions: one call
                    (gp, gd), (gp_j, gd_j) = _sweep_backward_pair((saved, g), (saved_j, g_j), ctx.cfgs[i], need, need_j,
                                                                  g_logits, g_sigma, accumulate=started)
                else:
                    gp, gd, ws = _sweep_backward(saved, ctx.cfgs[i], g, need, defer=True)
                    gp_j, gd_j, ws_j = _sweep_backward(saved_j, ctx.cfgs[j], g_j, need_j, defer=True)
                    _gather_pair((saved, ws), (saved_j, ws_j), ctx.cfgs[i], g_logits, g_sigma, accumulate=started)
                    # the two (g_l, g_s) scratch workspaces (2 x [B,N,H,W,2] floats: 770 MB at 8x49x192x640, twice what
                    # sequential views hold at a time) go back to the allocator now, not when the node's frame dies
                    del ws, ws_j
                per_view[i], per_view[j] = (gp, gd), (gp_j, gd_j)
                k += 2
                continue
            if g_logits is None and g_sigma is None:
                g_logits, g_sigma, gp, gd = _sweep_backward(saved, ctx.cfgs[i], g, need)
            elif can:
                gl, gs, gp, gd = _sweep_backward(saved, ctx.cfgs[i], g, need, into=(g_logits, g_sigma), accumulate=True)
                g_sigma = g_sigma if g_sigma is not None else gs
            else:
                gl, gs, gp, gd = _sweep_backward(saved, ctx.cfgs[i], g, need)
                if gl is not None:
                    g_logits = gl if g_logits is None else g_logits.add_(gl)
                if gs is not None:
                    g_sigma = gs if g_sigma is None else g_sigma.add_(gs)
            per_view[i] = (gp, gd)
            k += 1
        out = [None, g_logits, g_sigma]
        for i in range(n):
            gp, gd = per_view.get(i, (None, None))
            out += [None, gp, None, None, None, gd, None, None, None]
        return tuple(out)


def plane_sweep_multi(deferred):
    """``deferred``: one argument tuple per target view as returned by ``plane_sweep_disp(..., defer=True)`` /
    ``plane_sweep_homography(..., defer=True)`` — all over the same (src, logits, sigma).  Returns a list of
    ``(rgb_rec, ph_map, ph_mean)`` per view; see _MultiPlaneSweep."""
    src, _, logits = deferred[0][0], deferred[0][1], deferred[0][2]
    sigma = next((d[3] for d in deferred if d[3] is not None), None)
    flat = []
    for d in deferred:
        if d[0] is not src or d[2] is not logits or (d[3] is not None and d[3] is not sigma):
            raise ValueError("plane_sweep_multi: every view must sweep the same src / logits / sigma tensors")
        flat += [d[1]] + list(d[4:12])   # (a 13th element, the decoder tail's link, serves single-view nodes only)
    outs = _MultiPlaneSweep.apply(src, logits, sigma, *flat)
    return [tuple(outs[3 * i:3 * i + 3]) for i in range(len(deferred))]


def _flags(use_mixture_loss, automask, dense=False, render=False, rows=False):
    return ((C.PD_MIXTURE if use_mixture_loss else 0) | (C.PD_AUTOMASK if automask else 0) |
            (C.PD_DISP_DENSE if dense else 0) | (C.PD_RENDER_PROB if render else 0) | (C.PD_DISP_ROWS if rows else 0))


_SIGN = {"r": 1.0, "l": -1.0}


def _per_plane_view(disp_layered):
    """[B,N] view of an H/W-expanded disparity tensor, taken from the tensor it was expanded FROM when possible.

    ``disp_layered[:, :, 0, 0]`` would be correct but makes autograd materialise a zero [B,N,H,W] gradient and then
    reduce it again (ExpandBackward): ~0.1 ms per step of pure overhead at 8x49x192x640.  When the view's base is the
    decoder's [B,N,1,1] tensor (networks/depth_decoder.py:153-156) the gradient is handed to that tensor directly.
    """
    B, N = disp_layered.shape[:2]
    base = disp_layered._base
    if (base is not None and base.dim() == 4 and tuple(base.shape) == (B, N, 1, 1)
            and base.storage_offset() == disp_layered.storage_offset()
            and base.stride()[:2] == disp_layered.stride()[:2]
            and base.requires_grad == disp_layered.requires_grad):
        return base.reshape(B, N)
    return disp_layered[:, :, 0, 0]


class _FirstColumn(torch.autograd.Function):
    """``dense[..., 0]`` of a [B,N,H,W] map that is constant along x by the caller's promise (``row_uniform``: xy and xz
    planes, networks/depth_decoder.py:153-181) -> contiguous [B,N,H].

    Backward: the row's gradient goes back as ``g / W`` on EVERY column, as an expanded (stride-0) view — whatever built
    the map from x-independent quantities (the decoder's ``expand`` / its y-grid formula) sums over x and receives exactly
    ``g``.  A plain ``dense[..., 0]`` hands autograd a SelectBackward that zero-fills a [B,N,H,W] tensor per step to carry one
    column (248 MB at 8x63x192x640: 0.037 ms next to a 0.38 ms path) and makes that expand-backward read it all."""

    @staticmethod
    def forward(ctx, dense):
        ctx.W = dense.shape[-1]
        return dense[..., 0].contiguous()

    @staticmethod
    def backward(ctx, g):
        return (g * (1.0 / ctx.W)).unsqueeze(-1).expand(*g.shape, ctx.W)


def plane_sweep_disp(src, tgt, logits, sigma, disp_layered, padding_mask=None, *, target_side="r",
                     use_mixture_loss=True, automask=False, render_probability=False, dists=None, row_uniform=False,
                     return_mean=False, defer=False, _rows=None):
    """``disp_warp`` sweep (reference trainer.py:540-554 + 567-603 + 728-742) -> (rgb_rec, ph_map).

    ``disp_layered`` is the decoder's ``outputs["disp_layered"]``: either an expanded view of per-plane scalars
    ``[B,N,1,1] -> [B,N,H,W]`` (xy planes only; detected from its strides and passed as ``[B,N]`` without ever
    being materialised) or a dense ``[B,N,H,W]`` map (xz / yz planes present).  ``row_uniform=True`` promises that a
    dense map is constant along x (true for xy and xz planes: networks/depth_decoder.py:153-181 build them from the
    y-grid only; false once yz planes exist): its first column is then used as ``[B,N,H]`` per-row disparities, which
    keeps the row-shift kernels applicable.

    Gradient of a dense ``row_uniform`` map.  The reference's autograd hands ``disp_layered`` a dense [B,N,H,W] gradient
    (every column its own share).  Here the row's total ``g[b,n,y]`` comes back SPREAD EVENLY, ``g / W`` on every column, as
    a stride-0 view (``_FirstColumn``): anything that built the map from x-independent quantities — the decoder's
    ``expand`` and its y-grid formula, depth_decoder.py:153-181 — sums over x and receives exactly the reference's
    gradient, and nothing [B,N,H,W]-sized is written.  Per-column values differ from the reference's (their sum over x does
    not): a hook or a consumer that reads individual columns of ``disp_layered.grad`` must not pass ``row_uniform=True``.  A
    map that is a LEAF (``disp_layered.is_leaf``: somebody wants ``.grad`` itself) gets the plain select gradient instead —
    the row totals on column 0, zeros elsewhere.
    """
    B, N, H, W = logits.shape
    if _rows is not None:
        # internal (the stereo view of homography_warp): per-row shifts [B,N,H] and per-row mask [B,N,H] as they are — no
        # [B,N,H,W] view whose slice-backward would zero-fill and reduce 190 MB per step
        probe = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, C.PD_DISP_ROWS | C.PD_MASK_ROWS, 1.0, S.SWEEP_IMPL)
        if C.load().pd_sweep_uses_rowshift(ctypes.byref(probe)):
            shift, mask = _rows
            flags = _flags(use_mixture_loss, automask, rows=True, render=render_probability) | C.PD_MASK_ROWS
            call = (src, tgt, logits, sigma if use_mixture_loss else None, shift, None, None, mask,
                    dists if render_probability else None, C.PD_WARP_DISP, flags, _SIGN.get(target_side, 0.0))
            if defer:
                return call
            out = _PlaneSweep.apply(*call)
            return out if return_mean else out[:2]
        disp_layered, padding_mask = (t[..., None].expand(B, N, H, W) for t in _rows)   # PD_IMPL_GENERAL & co.
    if tuple(disp_layered.shape) != (B, N, H, W):
        disp_layered = disp_layered.expand(B, N, H, W)
    per_plane = disp_layered.stride(2) == 0 and disp_layered.stride(3) == 0
    rows = False
    if per_plane:
        plane = _per_plane_view(disp_layered)
    elif row_uniform:
        probe = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, C.PD_DISP_ROWS, 1.0, S.SWEEP_IMPL)
        rows = bool(C.load().pd_sweep_uses_rowshift(ctypes.byref(probe)))
        if rows:   # a LEAF map keeps the exact select gradient (g on column 0, zeros elsewhere); see the docstring
            plane = disp_layered[..., 0].contiguous() if disp_layered.is_leaf else _FirstColumn.apply(disp_layered)
        else:
            plane = disp_layered
    else:
        plane = disp_layered
    if padding_mask is not None and padding_mask.dtype != torch.float32:
        padding_mask = padding_mask.float()
    if padding_mask is not None and tuple(padding_mask.shape) != (B, N, H, W):
        padding_mask = padding_mask.expand(B, N, H, W)
    flags = _flags(use_mixture_loss, automask, dense=not (per_plane or rows), render=render_probability, rows=rows)
    if padding_mask is not None and row_uniform and (per_plane or rows):
        # the mask of xy / xz planes is constant along x as well (depth_decoder.py:157, 166): hand over its first column
        probe = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, flags, 1.0, S.SWEEP_IMPL)
        if C.load().pd_sweep_uses_rowshift(ctypes.byref(probe)):
            padding_mask = padding_mask[..., 0]
            flags |= C.PD_MASK_ROWS
    sign = _SIGN.get(target_side, 0.0)  # any other key leaves the grid untouched (trainer.py:546-549)
    call = (src, tgt, logits, sigma if use_mixture_loss else None, plane, None, None, padding_mask,
            dists if render_probability else None, C.PD_WARP_DISP, flags, sign)
    # a fused decoder tail that asked for it (decoder_tail(..., fuse_sweep_backward=True)) gets its backward applied by this
    # sweep's backward kernel — where the library serves that form for this descriptor
    link = getattr(logits, "_pd_tail_link", None)
    if (link is not None and per_plane and use_mixture_loss and padding_mask is None and not render_probability
            and sigma is not None and getattr(sigma, "_pd_tail_link", None) is link
            and C.load().pd_sweep_bwd_tail_fuses(ctypes.byref(_desc(B, N, H, W, C.PD_WARP_DISP, flags, sign)))):
        link.consumers += 1
        call = call + (link,)
    elif link is not None:
        link.consumers += 2   # a consumer the fused form does not serve: nobody fuses
    if defer:      # the argument tuple for plane_sweep_multi (several target views as one autograd node)
        return call
    out = _PlaneSweep.apply(*call)
    return out if return_mean else out[:2]  # (rgb_rec, ph_map[, ph_map.mean() fused into the kernel])


def homography_matrices(d, n, T, K, inv_K):
    """The O(B*N) 3x3 algebra of HomographyWarp.forward (layers.py:206-219, 223) in stock torch.

    Stays in torch on purpose: it keeps ``torch.inverse``'s rounding and lets autograd carry the
    gradient of ``H_t2s`` on to the pose network / plane distances.  Returns (H_t2s [BN,3,3], R·n [BN,3]).
    """
    B, N = d.shape
    Rm = T[:, :3, :3]
    t = T[:, :3, 3:4]
    nn_ = n.reshape(B * N, 1, 3)
    Rtnd = Rm + torch.matmul(t, nn_) / d.reshape(B * N, 1, 1)
    H_s2t = torch.matmul(K[:, :3, :3], torch.matmul(Rtnd, inv_K[:, :3, :3]))
    H_t2s = torch.inverse(H_s2t)
    Rn = torch.matmul(Rm, nn_.transpose(1, 2))[:, :, 0]
    return H_t2s, Rn


class _HomographyMatrices(torch.autograd.Function):
    """pd_homography_matrices_fwd/bwd: (distance [B,N], norm [B,N,3], T, K, inv_K [B,4,4]) -> per ``mode``
    (H_t2s, Rn) or (shift, mask, Rn).  Gradients to distance, norm and T."""

    @staticmethod
    def forward(ctx, distance, norm, T, K, inv_K, mode, rows):
        lib = C.load()
        B, N = distance.shape
        dev = distance.device
        distance, norm, T, K, inv_K = (_contig(t.detach().float()) for t in (distance, norm, T, K, inv_K))
        for name, t, shape in (("distance", distance, (B, N)), ("norm", norm, (B, N, 3)), ("T", T, (B, 4, 4)),
                               ("K", K, (B, 4, 4)), ("inv_K", inv_K, (B, 4, 4))):
            C.require_gpu_tensor(name, t, shape)
        Rn = torch.empty(B, N, 3, device=dev)
        Hm = shift = mask = None
        if mode == C.PD_HMAT_STEREO_ROWS:
            shift, mask = torch.empty(B, N, rows, device=dev), torch.empty(B, N, rows, device=dev)
        else:
            Hm = torch.empty(B, 4 if mode == C.PD_HMAT_UNIFORM else N, 3, 3, device=dev)
        with C.on_device(dev):
            C.check(lib.pd_homography_matrices_fwd(B, N, mode, rows, C.ptr(distance), C.ptr(norm), C.ptr(T), C.ptr(K),
                                                   C.ptr(inv_K), C.ptr(Hm), C.ptr(Rn), C.ptr(shift), C.ptr(mask),
                                                   C.stream_handle(dev)), "pd_homography_matrices_fwd")
        ctx.save_for_backward(distance, norm, T, K, inv_K)
        ctx.mode, ctx.rows = mode, rows
        ctx.set_materialize_grads(False)   # (else autograd zero-fills gradients for the non-differentiable Rn / mask: two launches)
        ctx.mark_non_differentiable(Rn)
        if mode == C.PD_HMAT_STEREO_ROWS:
            ctx.mark_non_differentiable(mask)
            return shift, mask, Rn
        return Hm, Rn

    @staticmethod
    def backward(ctx, g_first, *_):
        lib = C.load()
        distance, norm, T, K, inv_K = ctx.saved_tensors
        B, N = distance.shape
        dev = distance.device
        need_d, need_n, need_T = ctx.needs_input_grad[:3]
        stereo = ctx.mode == C.PD_HMAT_STEREO_ROWS
        if stereo and (need_n or need_T):
            raise RuntimeError("PD_HMAT_STEREO_ROWS carries the gradient of `distance` only (h00 is not part of the "
                               "per-row shift); use PD_HMAT_PLANES when the pose or the normals need gradients")
        if g_first is None:   # the matrices took no part in the loss
            return None, None, None, None, None, None, None
        g_first = _contig(g_first.float())
        gd = torch.empty(B, N, device=dev) if need_d else None
        gn = torch.empty(B, N, 3, device=dev) if need_n else None
        gT = torch.empty(B, 4, 4, device=dev) if need_T else None
        with C.on_device(dev):
            C.check(lib.pd_homography_matrices_bwd(B, N, ctx.mode, ctx.rows, C.ptr(distance), C.ptr(norm), C.ptr(T),
                                                   C.ptr(K), C.ptr(inv_K), C.ptr(None if stereo else g_first),
                                                   C.ptr(g_first if stereo else None), C.ptr(gd), C.ptr(gn), C.ptr(gT),
                                                   C.stream_handle(dev)), "pd_homography_matrices_bwd")
        return gd, gn, gT, None, None, None, None


def homography_matrices_fused(distance, norm, T, K, inv_K, mode=C.PD_HMAT_PLANES, rows=0):
    """layers.py:206-219, 223-225 in one launch (fp64 inside, rounded once): see include/planedepth_hip.h,
    ``pd_homography_matrices_fwd``.  distance [B,N], norm [B,N,3], T / K / inv_K [B,4,4] (NOT expanded over planes).
    Returns (H_t2s, Rn) — [B,N,3,3] or, PD_HMAT_UNIFORM, [B,4,3,3] — or (shift, mask, Rn) for PD_HMAT_STEREO_ROWS."""
    B, N = distance.shape
    if tuple(norm.shape) != (B, N, 3):
        norm = norm.expand(B, N, 3)
    return _HomographyMatrices.apply(distance, norm, T, K, inv_K, int(mode), int(rows))




def plane_sweep_homography(src, tgt, logits, sigma, distance, norm, T, K, inv_K, *, use_mixture_loss=True,
                           automask=False, render_probability=False, dists=None, return_mean=False, plane_uniform=False,
                           stereo_rows=False, defer=False):
    """``homography_warp`` sweep (reference trainer.py:556-560 + layers.py:206-234 + trainer.py:567-603, 728-742).

    distance [B,N], norm [B,N,3]; T, K, inv_K are the per-image [B,4,4] matrices (expanded over planes here).

    ``plane_uniform=True`` is the caller's promise that T has ZERO translation (what Trainer.predict_poses produces for
    the novel frames without COLMAP, trainer.py:386-400): K (R + t n^T/d) K^-1 is then the same matrix for every plane,
    so ONE homography per image is formed (from plane 0's d, n — they drop out) and the plane-uniform kernels run
    (geometry once per pixel, atomic-free backward).  The facing test keeps its per-plane normals.

    ``stereo_rows=True`` is the caller's promise that T is the dataset's stereo extrinsic (identity rotation, translation
    along x only: datasets/mono_dataset.py:203-211) and that no plane normal has an x component (xy and xz planes,
    networks/depth_decoder.py:153-207).  K (I + t n^T/d) K^-1 then differs from the identity in h01 and h02 only: the
    warp is a horizontal shift ``h01*y + h02`` per (plane, row) and the facing test is constant along x, i.e. exactly
    the ``disp_warp`` sweep with per-row disparities and a per-row mask, which runs on the row-shift kernels (no
    atomics).  H_t2s is still formed by the reference's chain (torch.inverse and all) and autograd carries the
    gradient of the shifts back into ``distance``; it is NOT taken when T or norm require gradients (their
    derivatives need h00 as well).
    """
    B, N, H, W = logits.shape
    if plane_uniform and N * H * W >= (1 << 29):
        plane_uniform = False   # the plane-uniform kernels address one image's [N,H,W] block with 32-bit byte offsets; beyond
        # that the per-plane route below (one matrix per plane, 64-bit addressing) serves the same poses
    # (PD_TORCH_HOMOGRAPHY: the row form's premise h00 = 1, z = 1 holds to 2e-7 for the fp64-formed matrices only; an fp32
    # torch.inverse at cond ~1e3 leaves h00 - 1 ~ 1e-5, i.e. up to 6e-3 pixels across a 640-pixel row, which the reference's own
    # chain carries into the result (measured on the reference-captured matrices: rgb_rec 1.9e-4 off) -> per-plane kernels)
    if stereo_rows and not S.TORCH_HOMOGRAPHY and not T.requires_grad and not norm.requires_grad:
        return _stereo_rows_sweep(src, tgt, logits, sigma, distance, norm, T, K, inv_K, use_mixture_loss, automask,
                                  return_mean, defer, render_probability, dists)
    ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
    inv_K3 = inv_K[:, :3, :3]
    flags = _flags(use_mixture_loss, automask, render=render_probability)
    tw = None
    if plane_uniform:
        # One matrix per image (slice 0, layers.py:216-218 for plane 0 with the — zero — translation detached) plus the
        # homographies of three virtual planes n/d = e_j that carry the translation's gradient (include/planedepth_hip.h,
        # PD_HOMO_UNIFORM): dL/dt = sum_j <sum_n G_n n_n[j]/d_n, d f(R + t e_j^T)/dt> is the per-plane formulation's.
        if S.TORCH_HOMOGRAPHY:
            Rm, t = T[:, :3, :3], T[:, :3, 3:4]
            K3 = K[:, :3, :3]
            n0 = norm[:, 0].reshape(B, 1, 3)
            eye = torch.eye(3, device=T.device, dtype=T.dtype)
            # [B,4,3,3] in one batch: slice 0 = R + t_detached n0^T / d0, slices 1..3 = R_detached + t e_j^T
            Rtnd = torch.cat([(Rm + torch.matmul(t.detach(), n0) / distance[:, 0].reshape(B, 1, 1))[:, None],
                              Rm.detach()[:, None] + t[:, None] * eye.reshape(1, 3, 1, 3)], 1)
            H_t2s = torch.inverse(torch.matmul(K3[:, None], torch.matmul(Rtnd, inv_K3[:, None])))       # [B,4,3,3]
            with torch.no_grad():
                Rn = torch.matmul(Rm[:, None], norm.reshape(B, N, 3, 1))[..., 0].reshape(B * N, 3)
        else:
            H_t2s, Rn = homography_matrices_fused(distance.detach(), norm.detach(), T, K, inv_K, C.PD_HMAT_UNIFORM)
            Rn = Rn.reshape(B * N, 3)
        with torch.no_grad():
            tw = (norm / distance[..., None]).contiguous()                                # [B,N,3]
        flags |= C.PD_HOMO_UNIFORM
    elif S.TORCH_HOMOGRAPHY:
        H_t2s, Rn = homography_matrices(distance, norm, ex(T), ex(K), ex(inv_K))
    else:
        H_t2s, Rn = homography_matrices_fused(distance, norm, T, K, inv_K)
        H_t2s, Rn = H_t2s.reshape(B * N, 3, 3), Rn.reshape(B * N, 3)
    call = (src, tgt, logits, sigma if use_mixture_loss else None, H_t2s, Rn.detach().contiguous(), inv_K3.detach(), tw,
            dists if render_probability else None, C.PD_WARP_HOMOGRAPHY, flags, 0.0)
    if defer:
        return call
    out = _PlaneSweep.apply(*call)
    return out if return_mean else out[:2]


def _stereo_rows_sweep(src, tgt, logits, sigma, distance, norm, T, K, inv_K, mix, automask, return_mean, defer=False,
                       render=False, dists=None):
    B, N, H, W = logits.shape
    shift, mask, _ = homography_matrices_fused(distance, norm, T, K, inv_K, C.PD_HMAT_STEREO_ROWS, rows=H)
    return plane_sweep_disp(src, tgt, logits, sigma, None, None, target_side="r", use_mixture_loss=mix,
                            automask=automask, row_uniform=True, return_mean=return_mean, defer=defer,
                            render_probability=render, dists=dists, _rows=(shift, mask))


def plane_sweep_layers(src, logits, sigma, *, disp_layered=None, padding_mask=None, target_side="r",
                       homography=None, use_mixture_loss=True, render_probability=False, dists=None,
                       want=("rgb_rec_layered", "logit_rec", "probability_rec", "sigma_rec", "pi_rec")):
    """Materialise the per-plane tensors the reference keeps in ``outputs`` (trainer.py:582-602).  No gradients."""
    lib = C.load()
    B, N, H, W = logits.shape
    with torch.no_grad():
        if homography is None:
            if tuple(disp_layered.shape) != (B, N, H, W):
                disp_layered = disp_layered.expand(B, N, H, W)
            per_plane = disp_layered.stride(2) == 0 and disp_layered.stride(3) == 0
            plane = (disp_layered[:, :, 0, 0] if per_plane else disp_layered).contiguous()  # layers: general kernels
            aux = k3 = None
            mode, sign = C.PD_WARP_DISP, _SIGN.get(target_side, 0.0)
            flags = _flags(use_mixture_loss, False, dense=not per_plane, render=render_probability)
            if padding_mask is not None:
                padding_mask = padding_mask.float().expand(B, N, H, W).contiguous()
        else:
            plane, aux, k3 = (t.contiguous() for t in homography)
            mode, sign, padding_mask = C.PD_WARP_HOMOGRAPHY, 0.0, None
            flags = _flags(use_mixture_loss, False, render=render_probability)
        dev = logits.device
        out = {}
        shapes = dict(rgb_rec_layered=(B, N, 3, H, W), logit_rec=(B, N, H, W), probability_rec=(B, N, H, W),
                      sigma_rec=(B, N, H, W), pi_rec=(B, N, H, W))
        for k in want:
            if k in ("sigma_rec", "pi_rec") and not use_mixture_loss:
                continue
            out[k] = torch.empty(shapes[k], device=dev, dtype=torch.float32)
        d = _desc(B, N, H, W, mode, flags, sign)
        with C.on_device(dev):
            rc = lib.pd_plane_sweep_layers(ctypes.byref(d), C.ptr(src.contiguous()), C.ptr(logits.contiguous()),
                                           C.ptr(_contig(sigma) if use_mixture_loss else None), C.ptr(plane),
                                           C.ptr(aux), C.ptr(k3), C.ptr(padding_mask),
                                           C.ptr(dists.contiguous() if render_probability else None),
                                           C.ptr(out.get("rgb_rec_layered")), C.ptr(out.get("logit_rec")),
                                           C.ptr(out.get("probability_rec")), C.ptr(out.get("sigma_rec")),
                                           C.ptr(out.get("pi_rec")), C.stream_handle(dev))
        C.check(rc, "pd_plane_sweep_layers")
    return out


