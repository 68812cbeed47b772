"""Autograd-aware operators over the C ABI (``include/planedepth_hip.h``) — the one namespace the rest of the package, the tests
and ``bench.py`` use.  The operators live in ``sweep`` (the fused plane sweep: autograd nodes, routing, homography algebra),
``tails`` (decoder / PladeNet tails), ``losses`` (SSIM, mixture NLL, masked photometric, smoothness), ``postprocess``
(self-distillation warps, batch doubling, crop grid), ``geometry`` (backproject / project / homography grids, grid_sample) and
``_buffers`` (descriptors, pre-zeroed pools); the switches tests flip between calls (``ops.SWEEP_IMPL = ...``) are attributes
of ``_state`` that this module forwards both ways.

Every function launches hand-written HIP kernels through ctypes on torch's current stream.  PyTorch is used for device
memory, streams and autograd plumbing only; there is no eager / CPU implementation behind these ops.
"""
import sys
import types

from . import _capi as C  # noqa: F401
from . import _state
from ._buffers import (  # noqa: F401
    _timed, _desc, _contig, _ZERO_POOL, _zero_scalar, _ZERO_BLOCKS,
    _ZERO_BLOCK_FLOATS, _zero_block, _plane_grad_buffer)
from .sweep import (  # noqa: F401
    _sweep_forward, _sweep_forward_pair, _sweep_backward_pair, _sweep_backward, _sweep_backward_tail, _gather_pair,
    TailLink, _GradTap, tail_taps, _PlaneSweep, _PER_SIDE, _MultiPlaneSweep,
    plane_sweep_multi, _flags, _SIGN, _per_plane_view, _FirstColumn, plane_sweep_disp,
    homography_matrices, _HomographyMatrices, homography_matrices_fused, plane_sweep_homography, _stereo_rows_sweep, plane_sweep_layers)
from .tails import (  # noqa: F401
    _PlaneLevels, plane_disparities,
    _DecoderTail, decoder_tail, _PladeTail, _RAY_NORM, camera_ray_norm, _camera_ray_norm,
    plade_tail)
from .losses import (  # noqa: F401
    _SSIM, ssim, _ReprojLoss, reprojection_loss, _MixtureNLL, multimodal_loss,
    _MaskedPhotometric, masked_photometric, _row_strided, _SmoothLoss, smooth_loss_disp)
from .postprocess import (  # noqa: F401
    _pp_disp, warp_softmax, warp_sum, pp_combine, post_process_disp, post_process_disp_stepwise, cat_flip,
    crop_grid)
from .geometry import (  # noqa: F401
    _Backproject, backproject_depth, _Project3D, project_3d, _HomographyGrid, homography_grid,
    _GridSample, grid_sample)
from ._state import _env_int  # noqa: F401


class _OpsModule(types.ModuleType):
    """``ops.<SWITCH>`` is ``_state.<SWITCH>``: read at call time by the operator modules, so an assignment here (tests,
    bench.py, monkeypatch) takes effect everywhere."""

    def __getattr__(self, name):
        if name in _state.SWITCHES:
            return getattr(_state, name)
        raise AttributeError("module %r has no attribute %r" % (self.__name__, name))

    def __setattr__(self, name, value):
        if name in _state.SWITCHES:
            setattr(_state, name, value)
        else:
            super().__setattr__(name, value)

    def __delattr__(self, name):
        if name in _state.SWITCHES:
            raise AttributeError("%s is a switch of planedepth_amd._state" % name)
        super().__delattr__(name)


sys.modules[__name__].__class__ = _OpsModule
