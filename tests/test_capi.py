"""CPU-side checks of the boundary: the C-ABI library loads, exports every declared symbol, validates arguments
without touching a GPU, and the Python operators refuse CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from planedepth_amd import _capi as C
from planedepth_amd import ops


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "planedepth_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.load()
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n
        assert n in C.SIGNATURES, "ctypes prototype missing for " + n
    assert sorted(C.SIGNATURES) == names
    assert lib.pd_version() >= 100
    import planedepth_amd
    major, minor, patch = (int(v) for v in planedepth_amd.__version__.split("."))
    assert lib.pd_version() == major * 1000 + minor * 100 + patch * 10, (lib.pd_version(), planedepth_amd.__version__)
    assert lib.pd_experiments() in (0, 1)
    # the library the suite runs against carries no timing-ablation / trace code (built with -DPD_DIAGNOSTICS: parts of the
    # kernels compiled out, results wrong by design) and, unless PD_LIB points at the experiments library, no experiments
    flags = lib.pd_build_flags()
    assert not flags & 2, "this library was built with -DPD_DIAGNOSTICS (timing ablations / traces): not a product build"
    assert (flags & 1) == lib.pd_experiments()
    if not os.environ.get("PD_LIB"):
        assert flags == 0, flags


def test_desc_layout_and_host_queries():
    assert ctypes.sizeof(C.SweepDesc) == 32
    lib = C.load()
    d = C.SweepDesc(2, 49, 192, 640, C.PD_WARP_DISP, C.PD_MIXTURE, 1.0, 0)
    assert lib.pd_sweep_stash_floats(ctypes.byref(d)) == (4 + 2) * 192 * 640
    # row-shift backward: [B][H][N] partial sums + [B][H][nseg*N][4] boundary spill (nseg = 10 at W = 640)
    assert lib.pd_sweep_bwd_workspace_floats(ctypes.byref(d)) == 2 * 192 * 49 * (1 + 4 * 10)
    d.mode = C.PD_WARP_HOMOGRAPHY
    assert lib.pd_sweep_stash_floats(ctypes.byref(d)) == 4 * 192 * 640
    # gather backward (one homography per plane): [B][480 workgroups][N*9] partial sums, 12 floats per (image, plane),
    # 4 flag words, the (g_l, g_s) scratch [B][N][H*W][2], 8 floats of alignment slack
    assert lib.pd_sweep_bwd_workspace_floats(ctypes.byref(d)) == 2 * 480 * 49 * 9 + 2 * 49 * 12 + 4 + 2 * 2 * 49 * 192 * 640 + 8
    d.impl = C.PD_IMPL_GENERAL
    # general backward: workgroups of four 63-pixel waves (ceil(192*640 / 252) = 488), nine sums per plane
    assert lib.pd_sweep_bwd_workspace_floats(ctypes.byref(d)) == 2 * 488 * 49 * 9


def test_argument_validation_needs_no_gpu():
    lib = C.load()
    d = C.SweepDesc(1, 4, 8, 8, C.PD_WARP_DISP, C.PD_MIXTURE, 1.0, 0)
    null = [None] * 14
    assert lib.pd_plane_sweep_fwd(ctypes.byref(d), *null) == 1  # PD_ERR_ARG: NULL tensors
    assert b"NULL" in lib.pd_last_error()
    d.mode = 7
    assert lib.pd_plane_sweep_fwd(ctypes.byref(d), *null) == 1
    assert b"warp mode" in lib.pd_last_error()
    assert lib.pd_ssim_fwd(0, 3, 8, 8, None, None, None, None) == 1
    assert lib.pd_grid_sample_fwd(1, 1, 4, 4, 4, 4, 9, None, None, None, None) == 1
    # the next-row entry points validate before they launch, too
    assert lib.pd_decoder_tail_fwd(1, 4, 8, 8, C.PD_TAIL_MIXTURE, *([None] * 10)) == 1
    assert b"NULL" in lib.pd_last_error()
    assert lib.pd_decoder_tail_fwd(1, 4, 8, 8, 64, *([None] * 10)) == 1
    assert b"flags" in lib.pd_last_error()
    assert lib.pd_smooth_loss_fwd(1, 3, 1, 8, None, 0, 0, None, 0, 0, 0, 1.0, None, None) == 1
    assert lib.pd_warp_sum(1, 4, 8, 8, 1.0, 0, None, None, 1.0, None, None) == 1
    assert lib.pd_pp_combine(1, 8, 8, None, None, None, None, None) == 1
    assert lib.pd_post_process(1, 4, 8, 8, 0, None, None, None, None, None, None, None, None) == 1
    assert lib.pd_post_process_workspace_floats(2, 4, 8, 8) == 2 * 4 * 64 + 6 * 2 * 64   # [B,N,H,W] + the chains' 2 x 3 S maps
    assert lib.pd_plane_levels_fwd(4, 4, 2.0, 300.0, 1.0, None, None, None, None) == 1
    assert lib.pd_plane_levels_bwd(4, 4, 2.0, 300.0, 1.0, None, None, None, None, None) == 1
    assert lib.pd_cat_flip(0, 3, 8, 8, None, None, 0, None, None) == 1
    assert lib.pd_mixture_nll_fwd(1, 4, 8, 8, 1, None, None, None, None, None) == 1
    assert lib.pd_decoder_tail_bwd_workspace_floats(2, 49, 192, 640) == 2 * 480 * 49
    # round 3: the pair gather takes the descriptor of two deferred plane-uniform backward calls, nothing else
    d = C.SweepDesc(1, 4, 8, 8, C.PD_WARP_HOMOGRAPHY, C.PD_MIXTURE | C.PD_HOMO_UNIFORM, 0.0, 0)
    assert lib.pd_uniform_gather_pair(ctypes.byref(d), *([None] * 9)) == 1
    assert b"PD_BWD_DEFER_GATHER" in lib.pd_last_error()
    d.flags |= C.PD_BWD_DEFER_GATHER
    assert lib.pd_uniform_gather_pair(ctypes.byref(d), *([None] * 9)) == 1
    assert b"NULL" in lib.pd_last_error()
    # round 4: the pair entry points take two pd_sweep_view structs over one src / logits / sigma, plane-uniform views only
    d = C.SweepDesc(1, 4, 8, 8, C.PD_WARP_HOMOGRAPHY, C.PD_MIXTURE | C.PD_HOMO_UNIFORM, 0.0, 0)
    va, vb = C.sweep_view(), C.sweep_view()
    assert ctypes.sizeof(C.SweepView) == 16 * ctypes.sizeof(ctypes.c_void_p)
    assert lib.pd_uniform_fwd_pair(ctypes.byref(d), None, None, None, None, None, None) == 1
    assert b"NULL" in lib.pd_last_error()
    assert lib.pd_uniform_fwd_pair(ctypes.byref(d), None, None, None, ctypes.byref(va), ctypes.byref(vb), None) == 1
    assert b"NULL" in lib.pd_last_error()
    assert lib.pd_uniform_bwd_pair(ctypes.byref(d), None, None, None, ctypes.byref(va), ctypes.byref(vb), None, None, None) == 1
    d.flags = C.PD_MIXTURE
    assert lib.pd_uniform_fwd_pair(ctypes.byref(d), None, None, None, ctypes.byref(va), ctypes.byref(vb), None) == 1
    assert b"PD_HOMO_UNIFORM" in lib.pd_last_error()
    d = C.SweepDesc(1, 4, 8, 8, C.PD_WARP_HOMOGRAPHY, C.PD_MIXTURE | C.PD_BWD_DEFER_GATHER, 0.0, 0)   # without PD_HOMO_UNIFORM
    assert lib.pd_plane_sweep_bwd(ctypes.byref(d), *([None] * 20)) == 1
    assert lib.pd_debug_gather_flags(ctypes.byref(d), None, None, None) == 1


def test_ops_refuse_cpu_tensors():
    B, N, H, W = 1, 3, 4, 6
    src = torch.rand(B, 3, H, W)
    lg = torch.randn(B, N, H, W)
    with pytest.raises(C.PlaneDepthHipError):
        ops.plane_sweep_disp(src, src, lg, lg.sigmoid(), torch.ones(B, N, 1, 1).expand(B, N, H, W))
    with pytest.raises(C.PlaneDepthHipError):
        ops.ssim(src, src)
    with pytest.raises(C.PlaneDepthHipError):
        ops.grid_sample(src, torch.zeros(B, H, W, 2))
    with pytest.raises(NotImplementedError):
        ops.grid_sample(src, torch.zeros(B, H, W, 2), padding_mode="reflection")
    with pytest.raises(C.PlaneDepthHipError):
        ops.decoder_tail(lg, lg, None, torch.ones(B, N, 1, 1).expand(B, N, H, W))
    with pytest.raises((C.PlaneDepthHipError, TypeError)):
        ops.smooth_loss_disp(src[:, :1], src, 2.0)
    with pytest.raises(C.PlaneDepthHipError):
        ops.warp_sum(lg, torch.ones(B, N, 1, 1).expand(B, N, H, W), 1.0)
    with pytest.raises(C.PlaneDepthHipError):
        ops.cat_flip(src, src)
    with pytest.raises(C.PlaneDepthHipError):
        ops.multimodal_loss(lg, lg.sigmoid(), lg.softmax(1), "lap")


def test_product_never_imports_the_oracle():
    import planedepth_amd
    pkg = os.path.dirname(planedepth_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle container", ""), f


def test_lazy_layers_is_safe_to_copy_and_reports_metadata_without_materialising():
    """LazyLayers must not recurse on copy / pickle (private names are not forwarded) and must answer
    shape / device / dtype / dim() without building the [B,N,H,W] tensor."""
    import copy
    import torch
    from planedepth_amd.decoder_tail import LazyLayers
    calls = []
    lz = LazyLayers((2, 3, 4, 5), lambda: calls.append(1) or torch.ones(2, 3, 4, 5), torch.device("cpu"), torch.float32)
    assert lz.shape == (2, 3, 4, 5) and lz.dim() == 4 and lz.size(1) == 3 and lz.dtype == torch.float32
    assert str(lz.device) == "cpu" and not calls
    dup = copy.copy(lz)
    assert dup.shape == lz.shape and not calls
    assert float(lz.sum()) == 120.0 and calls == [1]      # first real use materialises, once
    assert float(lz[0, 0, 0, 0]) == 1.0 and calls == [1]
    try:
        lz._no_such_private
    except AttributeError:
        pass
    else:
        raise AssertionError("private names must not be forwarded")


def test_first_column_hands_the_decoder_the_row_gradient_without_a_dense_zero_fill():
    """ops._FirstColumn (the [B,N,H] rows of a map that is constant along x by the caller's promise): same values as
    ``dense[..., 0]``, and the same gradient for whatever built the map from x-independent quantities (the decoder's
    expand of per-plane scalars, its y-grid formula for the xz planes), but carried by a stride-0 view instead of a
    zero-filled [B,N,H,W] tensor."""
    import torch
    from planedepth_amd import ops
    g = torch.Generator().manual_seed(5)
    B, N, H, W = 2, 5, 7, 12
    gain = 1.0 + torch.rand(1, N, H, 1, generator=g)
    weight = torch.randn(B, N, H, generator=g)
    grads = {}
    for name in ("select", "first_column"):
        p = torch.rand(B, N, 1, 1, generator=torch.Generator().manual_seed(6)).requires_grad_(True)   # per-plane scalars
        q = torch.rand(B, N, H, 1, generator=torch.Generator().manual_seed(7)).requires_grad_(True)   # a y-dependent term
        dense = p.expand(B, N, H, W) * gain + q.expand(B, N, H, W)
        rows = dense[..., 0] if name == "select" else ops._FirstColumn.apply(dense)
        assert tuple(rows.shape) == (B, N, H) and (name == "select" or rows.is_contiguous())
        (rows * weight).sum().backward()
        grads[name] = (p.grad.clone(), q.grad.clone(), rows.detach().clone())
    for a, b in zip(grads["select"], grads["first_column"]):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
    probe = torch.rand(B, N, H, W, generator=g).requires_grad_(True)
    out = ops._FirstColumn.apply(probe)
    gin, = torch.autograd.grad(out.sum(), probe)
    assert gin.stride(-1) == 0     # nothing [B,N,H,W]-sized was written


@pytest.mark.parametrize("src,flag", [("pd_plane_sweep_rowstream.hip", "-DPD_STREAM_ABL=8"), ("pd_plane_sweep_fwdstream.hip", "-DPD_FS_ABL=64"),
                                      ("pd_plane_sweep_fwdstream.hip", "-DPD_FS_TRACE=1"), ("pd_plane_sweep_rowshift.hip", "-DPD_ABLATE=1")])
def test_timing_ablation_switches_refuse_to_compile_without_the_diagnostics_flag(src, flag):
    """the timing ablations of the headline kernels (parts of the arithmetic or traffic compiled out: wrong results
    by design) can only be built into a library that reports it (pd_build_flags() & 2, which this suite and bench.py reject)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    cmd = [hipcc, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "planedepth_amd", "csrc"),
           "--cuda-host-only", "-fsyntax-only", flag, os.path.join(ROOT, "planedepth_amd", "csrc", src)]
    bad = subprocess.run(cmd, capture_output=True, text=True)
    assert bad.returncode != 0 and "need -DPD_DIAGNOSTICS" in bad.stderr, bad.stderr[-500:]
    good = subprocess.run(cmd + ["-DPD_DIAGNOSTICS"], capture_output=True, text=True)
    assert good.returncode == 0, good.stderr[-500:]


def test_per_source_compiler_flags_name_real_sources_and_enter_the_source_hash(monkeypatch):
    """__graft_entry__.FILE_FLAGS (the instruction scheduler per source, NOTEBOOK 11.5): every key is a source that exists, the
    flags are scheduler switches only (nothing that could change the arithmetic), and they are part of the hash that decides
    whether the library on disk is rebuilt — a library built with other flags must not pass for the tree's."""
    import __graft_entry__ as entry
    for name, flags in entry.FILE_FLAGS.items():
        assert os.path.isfile(os.path.join(entry.CSRC, name)), name
        assert flags[0::2] == ["-mllvm"] * (len(flags) // 2) and all(f.startswith("-amdgpu-sched-strategy=") for f in flags[1::2]), flags
    assert not any(f.startswith(("-ffast", "-ffp", "-O", "-funsafe")) for f in entry.BASE_FLAGS if f != "-O3")
    before = entry.source_hash()
    monkeypatch.setitem(entry.FILE_FLAGS, "pd_smooth.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"])
    assert entry.source_hash() != before
    monkeypatch.undo()
    assert entry.source_hash() == before
