"""Plumbing shared by the operator modules: descriptors, event timing around launches, pre-zeroed pools, the (optionally
poisoned) allocator front."""
import os

import torch

from . import _capi as C
from . import _state as S

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
class _timed:
    """Record a pair of events around a launch when ops.KERNEL_EVENTS is set (no cost otherwise)."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        if S.KERNEL_EVENTS is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if S.KERNEL_EVENTS is not None:
            self.b.record()
            S.KERNEL_EVENTS.setdefault(self.kind, []).append((self.a, self.b))
        return False


def _desc(B, N, H, W, mode, flags, sign):
    return C.SweepDesc(B, N, H, W, mode, flags, float(sign), S.SWEEP_IMPL)


def _contig(t):
    return None if t is None else t.contiguous()


_ZERO_POOL = {}   # (device, stream) -> [pool tensor, next free slot]


def _zero_scalar(device, slots=4096):
    """A fresh [1] float32 tensor that holds 0.0: slot i of a pool zeroed ONCE per `slots` calls (one fill launch for 4096
    forward calls instead of one memset launch each).  Every call gets its own slot, so a result the caller keeps (the
    loss value of an earlier step) is never written again; an exhausted pool is simply replaced (its slots live on through
    the tensors that view them).  Under stream capture (HIP graphs) the slot is zeroed in the captured work itself —
    a replay must start from zero every time."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(1, device=device, dtype=torch.float32)
    key = (device.index, C.raw_stream(device))   # zeroed on the stream its slots are used on
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
    if torch.cuda.is_current_stream_capturing() or not S.ZERO_POOL or n > _ZERO_BLOCK_FLOATS // 8:
        return torch.zeros(shape, device=device, dtype=torch.float32)
    key = (device.index, C.raw_stream(device))
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
    if (S.PLANE_ADDS and not torch.are_deterministic_algorithms_enabled() and mode == C.PD_WARP_DISP
            and not flags & (C.PD_DISP_DENSE | C.PD_DISP_ROWS)):
        return _zero_block(plane.device, tuple(plane.shape)), C.PD_BWD_PLANE_ZEROED
    return torch.empty_like(plane), 0

