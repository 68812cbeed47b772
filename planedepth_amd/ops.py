"""Autograd-aware operators over the C ABI (``include/planedepth_hip.h``).

Every function here launches hand-written HIP kernels through ctypes on torch's current stream.  PyTorch is used for
device memory, streams and autograd plumbing only; there is no eager / CPU implementation behind these ops.
"""
import ctypes
import os

import torch

from . import _capi as C


# Kernel selection for the sweep (C.PD_IMPL_AUTO | C.PD_IMPL_GENERAL).  Tests flip it to cross-check the specialised
# row-shift kernels against the general ones; leave it alone otherwise.
def _env_int(name):
    """A numeric environment switch as the library parses it (atoi; unset, empty, non-numeric or <= 0: off)."""
    try:
        return max(int(os.environ.get(name, "0") or 0), 0)
    except ValueError:
        return 0


SWEEP_IMPL = int(os.environ.get("PD_SWEEP_IMPL", C.PD_IMPL_AUTO))  # 0 auto, 1 general kernels, 2 fast rows (A/B runs)
LAST_SWEEP_FLAGS = None  # flags of the most recent sweep forward (introspection for tests)
DEBUG_STASH = None       # diagnostics (scripts/diag_w70.py): set to a list to collect the forward's per-pixel stash
PAIR_GATHER = os.environ.get("PD_PAIR_GATHER", "1") != "0"   # two plane-uniform views of a step: their second passes in one kernel
PAIR_FORWARD = os.environ.get("PD_PAIR_FORWARD", "1") != "0"   # ... and their forwards / first passes in one launch each
DEBUG_WORKSPACE = None   # diagnostics (tests): set to a list to collect (descriptor, workspace) of every sweep backward
if int(os.environ.get("PD_DEBUG_POISON_MEM", "0")):
    # diagnostics: every buffer this module allocates uninitialised (outputs, stash, workspaces) starts as NaNs, so a
    # kernel that reads global memory nobody wrote produces NaNs instead of depending on the allocator's leftovers
    class _PoisonedTorch:
        def __getattr__(self, name):
            return getattr(_real_torch, name)

        @staticmethod
        def empty(*a, **k):
            t = _real_torch.empty(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() and t.device.type == "cuda" else t

        @staticmethod
        def empty_like(x, **k):
            t = _real_torch.empty_like(x, **k)
            return t.fill_(float("nan")) if t.is_floating_point() and t.device.type == "cuda" else t

    _real_torch = torch
    torch = _PoisonedTorch()
KERNEL_EVENTS = None     # measurement (bench.py): set to a dict {"fwd": [], "bwd": []} to collect (start, end) CUDA events
                         # recorded on the launch stream around the sweep's C-ABI calls INSIDE a training step


class _timed:
    """Record a pair of events around a launch when ops.KERNEL_EVENTS is set (no cost otherwise)."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        if KERNEL_EVENTS is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if KERNEL_EVENTS is not None:
            self.b.record()
            KERNEL_EVENTS.setdefault(self.kind, []).append((self.a, self.b))
        return False


def _desc(B, N, H, W, mode, flags, sign):
    return C.SweepDesc(B, N, H, W, mode, flags, float(sign), SWEEP_IMPL)


def _contig(t):
    return None if t is None else t.contiguous()


ZERO_POOL = os.environ.get("PD_ZERO_POOL", "1") != "0"   # A/B switch: 0 = a memset launch per forward call instead
PLANE_ADDS = os.environ.get("PD_PLANE_ADDS", "1") != "0"   # A/B switch: 0 = per-row partial sums + a reduction launch per backward
_ZERO_POOL = {}   # (device, stream) -> [pool tensor, next free slot]


def _zero_scalar(device, slots=4096):
    """A fresh [1] float32 tensor that holds 0.0: slot i of a pool zeroed ONCE per `slots` calls (one fill launch for 4096
    forward calls instead of one memset launch each).  Every call gets its own slot, so a result the caller keeps (the
    loss value of an earlier step) is never written again; an exhausted pool is simply replaced (its slots live on through
    the tensors that view them).  Under stream capture (HIP graphs) the slot is zeroed in the captured work itself —
    a replay must start from zero every time."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(1, device=device, dtype=torch.float32)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)   # zeroed on the stream its slots are used on
    st = _ZERO_POOL.get(key)
    if st is None or st[1] >= slots:
        st = _ZERO_POOL[key] = [torch.zeros(slots, device=device, dtype=torch.float32), 0]
    i = st[1]
    st[1] = i + 1
    return st[0][i:i + 1]


_ZERO_BLOCKS = {}   # (device, stream) -> [pool tensor, next free float]
_ZERO_BLOCK_FLOATS = 1 << 18


def _zero_block(device, shape):
    """A fresh float32 tensor of ``shape`` that holds zeros, cut from a 1 MB pool zeroed once (same contract as
    ``_zero_scalar``: every call gets floats of its own, nothing handed out is ever written by the pool again).  Serves the
    per-plane disparity gradient under PD_BWD_PLANE_ZEROED — [B, N], 1.5 KB a call at the benchmark's shape."""
    n = 1
    for k in shape:
        n *= int(k)
    if torch.cuda.is_current_stream_capturing() or not ZERO_POOL or n > _ZERO_BLOCK_FLOATS // 8:
        return torch.zeros(shape, device=device, dtype=torch.float32)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    st = _ZERO_BLOCKS.get(key)
    if st is None or st[1] + n > _ZERO_BLOCK_FLOATS:
        st = _ZERO_BLOCKS[key] = [torch.zeros(_ZERO_BLOCK_FLOATS, device=device, dtype=torch.float32), 0]
    i = st[1]
    st[1] = i + ((n + 3) & ~3)   # 16-byte steps
    return st[0][i:i + n].view(shape)


def _plane_grad_buffer(plane, mode, flags):
    """(g_plane buffer, extra descriptor flags) for a backward call that wants the plane-parameter gradient: one disparity
    per plane gets a pre-zeroed [B, N] block and PD_BWD_PLANE_ZEROED (the row-stream backward then adds its rows' shares
    there and launches no reduction kernel; the other kernels overwrite it as ever)."""
    # (float atomics: the sum's last bits depend on the order of the adds — under torch.use_deterministic_algorithms(True) the
    # deterministic partial sums + reduction launch are used instead)
    if (PLANE_ADDS and not torch.are_deterministic_algorithms_enabled() and mode == C.PD_WARP_DISP
            and not flags & (C.PD_DISP_DENSE | C.PD_DISP_ROWS)):
        return _zero_block(plane.device, tuple(plane.shape)), C.PD_BWD_PLANE_ZEROED
    return torch.empty_like(plane), 0


# ---------------------------------------------------------------------------------------------------------------------
# Fused plane sweep + photometric loss
# ---------------------------------------------------------------------------------------------------------------------
def _sweep_forward(src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, mode, flags, sign):
    """One target view through pd_plane_sweep_fwd -> ((rgb_rec, ph_map, ph_mean[1]), tensors the backward needs)."""
    global LAST_SWEEP_FLAGS
    LAST_SWEEP_FLAGS = flags
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
    if ZERO_POOL:
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
    if DEBUG_STASH is not None:
        DEBUG_STASH.append(stash)
    return (rgb_rec, ph_map, ph_mean), (src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, rgb_rec, stash)


def _sweep_forward_pair(src, logits, sigma, side_a, side_b):
    """pd_uniform_fwd_pair: two plane-uniform target views (``side_*`` = (tgt, plane, plane_aux, inv_K3, padding_mask, dists,
    mode, flags, sign) with equal mode / flags / sign) of the same src / logits / sigma in one launch.  Returns what two
    ``_sweep_forward`` calls return."""
    global LAST_SWEEP_FLAGS
    lib = C.load()
    mode, flags, sign = side_a[6:9]
    LAST_SWEEP_FLAGS = flags
    B, N, H, W = logits.shape
    C.require_gpu_tensor("logits", logits)
    C.require_gpu_tensor("src", src, (B, 3, H, W))
    if flags & C.PD_MIXTURE:
        C.require_gpu_tensor("sigma", sigma, (B, N, H, W))
    src, logits, sigma = _contig(src), _contig(logits), _contig(sigma)
    d = _desc(B, N, H, W, mode, flags, sign)
    k = lib.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
    if ZERO_POOL:
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
        ph_mean = _zero_scalar(logits.device) if ZERO_POOL else torch.empty(1, device=logits.device, dtype=torch.float32)
        stash = torch.empty(B, k, H, W, device=logits.device, dtype=torch.float32)
        views.append(C.sweep_view(tgt=tgt, plane=plane, plane_aux=plane_aux, inv_K3=inv_K3, dists=dists, rgb_rec=rgb_rec,
                                  ph_map=ph_map, ph_mean=ph_mean, stash=stash))
        results.append(((rgb_rec, ph_map, ph_mean),
                        (src, tgt, logits, sigma, plane, plane_aux, inv_K3, padding_mask, dists, rgb_rec, stash)))
        if DEBUG_STASH is not None:
            DEBUG_STASH.append(stash)
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
        if DEBUG_WORKSPACE is not None:
            DEBUG_WORKSPACE.append((d, ws))
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
    if DEBUG_WORKSPACE is not None:
        DEBUG_WORKSPACE.append((d, ws))
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
        if PAIR_FORWARD:
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
            return (PAIR_GATHER and not _env_int("PD_UNI_CHUNK") and mode == C.PD_WARP_HOMOGRAPHY and
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
                if PAIR_FORWARD:   # both first passes in one launch, the pair gather, the reductions: one call
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
        probe = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, C.PD_DISP_ROWS | C.PD_MASK_ROWS, 1.0, SWEEP_IMPL)
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
        probe = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, C.PD_DISP_ROWS, 1.0, SWEEP_IMPL)
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
        probe = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, flags, 1.0, SWEEP_IMPL)
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


# PD_TORCH_HOMOGRAPHY=1: form the matrices with the stock torch chain (homography_matrices above: torch.inverse and its
# rounding, ~12 launches + rocSOLVER, not graph-capturable) instead of pd_homography_matrices_fwd/bwd
TORCH_HOMOGRAPHY = bool(int(os.environ.get("PD_TORCH_HOMOGRAPHY", "0")))


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
    if stereo_rows and not TORCH_HOMOGRAPHY and not T.requires_grad and not norm.requires_grad:
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
        if TORCH_HOMOGRAPHY:
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
    elif TORCH_HOMOGRAPHY:
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
# Post-process warps — forward only, as in the reference (no_grad networks, detached result)
# ---------------------------------------------------------------------------------------------------------------------
def _pp_disp(disp_layered, B, N, H, W):
    """(tensor, flags): per-plane [B,N] when the map is an H/W-expanded view, else the dense [B,N,H,W] map."""
    if tuple(disp_layered.shape) != (B, N, H, W):
        disp_layered = disp_layered.expand(B, N, H, W)
    if disp_layered.stride(2) == 0 and disp_layered.stride(3) == 0:
        return disp_layered[:, :, 0, 0].contiguous(), 0
    return disp_layered.contiguous(), C.PD_PP_DISP_DENSE


def warp_softmax(planes, disp_layered, sign, flip_src=False):
    """softmax over the planes of ``planes`` sampled at x + sign * disp (trainer.py:443-446 / 451-453)."""
    lib = C.load()
    C.require_gpu_tensor("planes", planes)
    B, N, H, W = planes.shape
    with torch.no_grad():
        planes = planes.detach().contiguous()
        disp, flags = _pp_disp(disp_layered.detach(), B, N, H, W)
        out = torch.empty_like(planes)
        with C.on_device(planes.device):
            C.check(lib.pd_warp_softmax(B, N, H, W, float(sign), flags | (C.PD_PP_FLIP_SRC if flip_src else 0),
                                        C.ptr(planes), C.ptr(disp), C.ptr(out), C.stream_handle(planes.device)),
                    "pd_warp_softmax")
    return out


def warp_sum(planes, disp_layered, sign, cap=1.0, flip_src=False):
    """min(cap, sum over the planes of ``planes`` sampled at x + sign * disp) (trainer.py:447-449, 454-456, 463-465)."""
    lib = C.load()
    C.require_gpu_tensor("planes", planes)
    B, N, H, W = planes.shape
    with torch.no_grad():
        planes = planes.detach().contiguous()
        disp, flags = _pp_disp(disp_layered.detach(), B, N, H, W)
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


def post_process_disp(logits, probability, disp, disp_layered):
    """trainer.py:421-466 given the fixed model's outputs for cat([image, mirrored image]) -> (disp_pp, mask_novel)."""
    B = probability.shape[0] // 2
    with torch.no_grad():
        dl_r, dl_l = disp_layered[:B], disp_layered[B:]
        plr = warp_softmax(logits[:B], dl_r, +1.0)                       # :443-446
        o_l = warp_sum(plr, dl_l, -1.0)                                  # :447-449
        pfrl = warp_softmax(logits[B:], dl_l, -1.0, flip_src=True)       # :451-453 (the flip is folded into the read)
        o_fr = warp_sum(pfrl, dl_r, +1.0)                                # :454-456
        disp_pp = pp_combine(disp, o_fr, o_l)                            # :458-461
        prob = probability.tensor() if hasattr(probability, "tensor") else probability
        mask_novel = warp_sum(prob[:B], dl_r, +1.0)                      # :463-465
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
