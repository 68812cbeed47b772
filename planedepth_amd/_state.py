"""Process-wide switches of the operators (``planedepth_amd.ops.<NAME>`` reads and writes these: tests and ``bench.py`` flip
them between calls; the modules that act on them read ``_state.<NAME>`` at call time)."""
import os

from . import _capi as C


def _env_int(name):
    """A numeric environment switch as the library parses it (atoi; unset, empty, non-numeric or <= 0: off)."""
    try:
        return max(int(os.environ.get(name, "0") or 0), 0)
    except ValueError:
        return 0


# Kernel selection for the sweep (C.PD_IMPL_AUTO | C.PD_IMPL_GENERAL | ...).  Tests flip it to cross-check the specialised
# row kernels against the general ones; leave it alone otherwise.
SWEEP_IMPL = int(os.environ.get("PD_SWEEP_IMPL", C.PD_IMPL_AUTO))  # 0 auto, 1 general kernels, 2 fast rows (A/B runs)
LAST_SWEEP_FLAGS = None  # flags of the most recent sweep forward (introspection for tests)
DEBUG_STASH = None       # diagnostics: set to a list to collect the forward's per-pixel stash
PAIR_GATHER = os.environ.get("PD_PAIR_GATHER", "1") != "0"   # two plane-uniform views of a step: their second passes in one kernel
PAIR_FORWARD = os.environ.get("PD_PAIR_FORWARD", "1") != "0"   # ... and their forwards / first passes in one launch each
DEBUG_WORKSPACE = None   # diagnostics (tests): set to a list to collect (descriptor, workspace) of every sweep backward
KERNEL_EVENTS = None     # measurement (bench.py): set to a dict {"fwd": [], "bwd": []} to collect (start, end) CUDA events
                         # recorded on the launch stream around the sweep's C-ABI calls INSIDE a training step
ZERO_POOL = os.environ.get("PD_ZERO_POOL", "1") != "0"   # A/B switch: 0 = a memset launch per forward call instead
PLANE_ADDS = os.environ.get("PD_PLANE_ADDS", "1") != "0"   # A/B switch: 0 = per-row partial sums + a reduction launch per backward
# PD_TORCH_HOMOGRAPHY=1: form the matrices with the stock torch chain (sweep.homography_matrices: torch.inverse and its
# rounding, ~12 launches + rocSOLVER, not graph-capturable) instead of pd_homography_matrices_fwd/bwd
TORCH_HOMOGRAPHY = bool(int(os.environ.get("PD_TORCH_HOMOGRAPHY", "0")))

SWITCHES = ("SWEEP_IMPL", "LAST_SWEEP_FLAGS", "DEBUG_STASH", "PAIR_GATHER", "PAIR_FORWARD", "DEBUG_WORKSPACE", "KERNEL_EVENTS",
            "ZERO_POOL", "PLANE_ADDS", "TORCH_HOMOGRAPHY")
