"""ctypes binding of ``libplanedepth_hip.so`` (C ABI declared in ``include/planedepth_hip.h``).

There is deliberately NO fallback: if the shared library is missing or a symbol is absent, importing the product
ops raises.  The library is built in-tree by ``__graft_entry__.build()`` (``hipcc --offload-arch=gfx950``).
"""
import ctypes
import os

import torch  # noqa: F401  (imported first so the process already holds torch's libamdhip64 — one HIP runtime only)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PD_LIB") or os.path.join(_HERE, "lib", "libplanedepth_hip.so")  # PD_LIB: diagnostics builds

PD_WARP_DISP, PD_WARP_HOMOGRAPHY = 0, 1
PD_MIXTURE, PD_AUTOMASK, PD_RENDER_PROB, PD_DISP_DENSE, PD_DISP_ROWS, PD_MASK_ROWS, PD_HOMO_UNIFORM = 1, 2, 4, 8, 16, 32, 64
PD_BWD_ACCUMULATE = 128
PD_BWD_DEFER_GATHER = 256
PD_PH_MEAN_ZEROED = 512
PD_BWD_PLANE_ZEROED = 1024
PD_PAD_ZEROS, PD_PAD_BORDER = 0, 1
PD_TAIL_MIXTURE, PD_TAIL_DISP_DENSE = 1, 2
PD_PP_DISP_DENSE, PD_PP_FLIP_SRC, PD_PP_DISP_ROWS = 1, 2, 4
PD_HMAT_PLANES, PD_HMAT_UNIFORM, PD_HMAT_STEREO_ROWS = 0, 1, 2
PD_IMPL_AUTO, PD_IMPL_GENERAL, PD_IMPL_FAST_ROWS, PD_IMPL_TILE, PD_IMPL_ROWS1, PD_IMPL_UNIFORM_DIRECT = 0, 1, 2, 3, 4, 5
PD_IMPL_EXACT_ROWS = 6


class SweepDesc(ctypes.Structure):
    """Mirror of ``pd_sweep_desc``."""
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("mode", ctypes.c_int32), ("flags", ctypes.c_int32), ("sign", ctypes.c_float),
                ("impl", ctypes.c_int32)]


class SweepView(ctypes.Structure):
    """Mirror of ``pd_sweep_view``: what differs between two target views of one source image (pd_uniform_*_pair)."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "tgt", "plane", "plane_aux", "inv_K3", "padding_mask", "dists", "rgb_rec", "ph_map", "ph_mean", "stash",
        "g_rgb_rec", "g_ph_map", "g_ph_mean", "g_plane", "g_dists", "workspace")]


def sweep_view(**tensors):
    """SweepView from keyword tensors (missing / None -> NULL)."""
    v = SweepView()
    for name, t in tensors.items():
        setattr(v, name, None if t is None else t.data_ptr())
    return v


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_int64
_D = ctypes.POINTER(SweepDesc)

# name -> (restype, argtypes); must list every symbol include/planedepth_hip.h declares
SIGNATURES = {
    "pd_version": (_I, []),
    "pd_last_error": (ctypes.c_char_p, []),
    "pd_source_hash": (ctypes.c_char_p, []),
    "pd_sweep_uses_rowshift": (_I, [_D]),
    "pd_sweep_auto_row_eps": (_F, []),
    "pd_sweep_bwd_accumulates": (_I, [_D]),
    "pd_sweep_bwd_plane_adds": (_I, [_D]),
    "pd_sweep_stash_floats": (ctypes.c_size_t, [_D]),
    "pd_sweep_bwd_workspace_floats": (ctypes.c_size_t, [_D]),
    "pd_plane_sweep_fwd": (_I, [_D] + [_P] * 14),
    "pd_plane_sweep_bwd": (_I, [_D] + [_P] * 20),
    "pd_plane_sweep_layers": (_I, [_D] + [_P] * 14),
    "pd_sweep_bwd_tail_fuses": (_I, [_D]),
    "pd_plane_sweep_bwd_tail": (_I, [_D] + [_P] * 20),
    "pd_uniform_gather_pair": (_I, [_D] + [_P] * 9),
    "pd_uniform_fwd_pair": (_I, [_D] + [_P] * 3 + [ctypes.POINTER(SweepView)] * 2 + [_P]),
    "pd_uniform_bwd_pair": (_I, [_D] + [_P] * 3 + [ctypes.POINTER(SweepView)] * 2 + [_P] * 3),
    "pd_ssim_fwd": (_I, [_I] * 4 + [_P] * 4),
    "pd_ssim_bwd": (_I, [_I] * 4 + [_P] * 6),
    "pd_reproj_loss_fwd": (_I, [_I] * 4 + [_P] * 4),
    "pd_reproj_loss_bwd": (_I, [_I] * 4 + [_P] * 6),
    "pd_mixture_nll_fwd": (_I, [_I] * 5 + [_P] * 5),
    "pd_mixture_nll_bwd": (_I, [_I] * 5 + [_P] * 8),
    "pd_decoder_tail_bwd_workspace_floats": (ctypes.c_size_t, [_I] * 4),
    "pd_decoder_tail_fwd": (_I, [_I] * 5 + [_P] * 10),
    "pd_decoder_tail_layers": (_I, [_I] * 5 + [_P] * 7),
    "pd_decoder_tail_bwd": (_I, [_I] * 5 + [_P] * 15),
    "pd_plade_tail_fwd": (_I, [_I] * 5 + [_P] * 11),
    "pd_plade_tail_layers": (_I, [_I] * 5 + [_P] * 8),
    "pd_plade_tail_bwd": (_I, [_I] * 5 + [_P] * 16),
    "pd_smooth_loss_fwd": (_I, [_I] * 4 + [_P, _L, _L, _P, _L, _L, _L, _F, _P, _P]),
    "pd_smooth_loss_bwd": (_I, [_I] * 4 + [_P, _L, _L, _P, _L, _L, _L, _F, _P, _P, _P]),
    "pd_smooth_loss_bwd_padded": (_I, [_I] * 5 + [_P, _L, _L, _P, _L, _L, _L, _F, _P, _P, _P]),
    "pd_warp_softmax": (_I, [_I] * 4 + [_F, _I, _P, _P, _P, _P]),
    "pd_warp_sum": (_I, [_I] * 4 + [_F, _I, _P, _P, _F, _P, _P]),
    "pd_post_process_workspace_floats": (ctypes.c_size_t, [_I] * 4),
    "pd_post_process": (_I, [_I] * 5 + [_P] * 8),
    "pd_pp_combine": (_I, [_I] * 3 + [_P] * 5),
    "pd_cat_flip": (_I, [_I] * 4 + [_P, _P, _I, _P, _P]),
    "pd_plane_levels_fwd": (_I, [_I, _I, _F, _F, _F, _P, _P, _P, _P]),
    "pd_plane_levels_bwd": (_I, [_I, _I, _F, _F, _F, _P, _P, _P, _P, _P]),
    "pd_crop_grid": (_I, [_I] * 3 + [_P, _P, _P]),
    "pd_selftest_division": (_I, [_F, _I, _F, _F, _P, _P]),
    "pd_experiments": (_I, []),
    "pd_build_flags": (_I, []),
    "pd_debug_gather_flags": (_I, [_D, _P, _P, _P]),
    "pd_debug_poison_lds": (_I, [_P]),
    "pd_debug_count_lds_nans": (_I, [_P, _P]),
    "pd_masked_photometric_fwd": (_I, [_I] * 4 + [_P] * 9),
    "pd_masked_photometric_bwd": (_I, [_I] * 4 + [_P] * 9),
    "pd_homography_matrices_fwd": (_I, [_I] * 4 + [_P] * 10),
    "pd_homography_matrices_bwd": (_I, [_I] * 4 + [_P] * 11),
    "pd_backproject": (_I, [_I] * 3 + [_P] * 4),
    "pd_backproject_bwd": (_I, [_I] * 3 + [_P] * 4),
    "pd_project3d": (_I, [_I] * 3 + [_F] + [_P] * 4),
    "pd_project3d_bwd": (_I, [_I] * 3 + [_F] + [_P] * 7),
    "pd_homography_grid": (_I, [_I] * 3 + [_P] * 6),
    "pd_homography_grid_bwd": (_I, [_I] * 3 + [_P] * 5),
    "pd_grid_sample_fwd": (_I, [_I] * 7 + [_P] * 4),
    "pd_grid_sample_bwd": (_I, [_I] * 7 + [_P] * 6),
}

_lib = None


class PlaneDepthHipError(RuntimeError):
    pass


def load():
    """Load the library once and attach prototypes.  Raises if it is missing — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PlaneDepthHipError(
            "planedepth_amd: %s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pd_last_error().decode(errors="replace")
        raise PlaneDepthHipError("%s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


POISON_LDS = bool(int(os.environ.get("PD_DEBUG_POISON_LDS", "0")))   # diagnostics: NaNs into the CUs' LDS before every launch


class _NoSwitch:
    """`with` target that does nothing (the tensor's device already is the current one)."""
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(device):
    """`with on_device(t.device):` — torch.cuda.device(device) only when a switch is needed.  The context manager costs
    several microseconds per entry (two set_device calls and their bookkeeping), which is most of what the HOST spends on
    an operator whose kernels take 30 us; on a one-GPU-per-process layout the tensor's device always is the current one."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


_RAW_STREAM = None if os.environ.get("PD_RAW_STREAM") == "0" else getattr(torch._C, "_cuda_getCurrentRawStream", None)   # (PD_RAW_STREAM=0: A/B)


def raw_stream(device=None):
    """torch's current stream on ``device`` as an integer (the hipStream_t).  torch.cuda.current_stream() builds a Stream object
    per call — ~10 us, four times per training step here (two launches, two pool look-ups: 40 of the step's 214 us of host time);
    the raw getter is a plain C call."""
    if _RAW_STREAM is not None:
        if device is None or isinstance(device, int):
            idx = device
        else:
            idx = (device if isinstance(device, torch.device) else torch.device(device)).index
        return _RAW_STREAM(torch.cuda.current_device() if idx is None else int(idx))
    return torch.cuda.current_stream(device).cuda_stream


def stream_handle(device=None):
    """The raw hipStream_t torch is currently enqueueing on (so our launches order with torch's ops)."""
    h = ctypes.c_void_p(raw_stream(device))
    if POISON_LDS:   # a kernel that reads shared memory it never wrote then produces NaNs instead of luck
        load().pd_debug_poison_lds(h)
    return h


def require_gpu_tensor(name, t, shape=None, dtype=torch.float32):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise PlaneDepthHipError("%s is on %s: planedepth_amd runs on the GPU only (no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    return t
