"""Parity tests of the kernels that LOST their A/B runs and are therefore not in the product library: the
four-pixels-per-lane row kernels (scripts/experiments/pd_plane_sweep_rowquad.hip), the owned-tile backward
(scripts/experiments/pd_plane_sweep_tile.hip) and the one-kernel plane-uniform backward
(scripts/experiments/pd_plane_sweep_uniform_fused.inc).  Not collected by the default runs (tests/conftest.py); to run them:

    bash scripts/build_experiments.sh
    PD_TEST_EXPERIMENTS=1 PD_LIB=$PWD/planedepth_amd/lib/libpd_experiments.so python -m pytest tests/experiments -q
"""
import pytest
import torch

from cases import rel_err
from test_gpu_parity import _wild_homographies, test_plane_uniform_homography_kernels_equal_the_general_ones as _uniform_body

pytestmark = [pytest.mark.experiments, pytest.mark.gpu]


def _need_experiments():
    """The kernels that lost their A/B runs (row-quad, owned-tile, one-kernel plane-uniform backward) are compiled with
    -DPD_EXPERIMENTS only (scripts/build_variants.sh); the product library does not carry them."""
    from planedepth_amd import _capi as C
    if not C.load().pd_experiments():
        pytest.skip("built without -DPD_EXPERIMENTS: this kernel is not part of the product library")



@pytest.mark.parametrize("B,N,H,W,mix", [(2, 5, 40, 150, True), (1, 9, 33, 70, True), (1, 3, 50, 200, False),
                                         (1, 4, 5, 7, True), (2, 6, 64, 64, True)])
def test_tile_backward_equals_atomic_backward(B, N, H, W, mix):
    """The owned-tile backward (pd_plane_sweep_tile.hip, PD_IMPL_TILE: no atomics, every gradient element stored once)
    against the atomic scatter on homographies far from the identity (rotation, zoom, shear, perspective, large shifts,
    planes facing away), ragged sizes (W not a multiple of 4 or 64, images smaller than one tile).  Two independent
    adjoints of the same gather: they must agree to summation order."""
    _need_experiments()
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics
    g = torch.Generator().manual_seed(100 + W)
    dev = "cuda"
    src, tgt = torch.rand(B, 3, H, W, generator=g).to(dev), torch.rand(B, 3, H, W, generator=g).to(dev)
    logits = torch.randn(B, N, H, W, generator=g).to(dev)
    sigma = (0.011 + 0.978 * torch.rand(B, N, H, W, generator=g)).to(dev)
    gw = torch.randn(B, 3, H, W, generator=g).to(dev)
    Hm, Rn = _wild_homographies(B, N, H, W, 7 + H)
    _, inv_K = intrinsics(B, H, W)
    flags = (C.PD_MIXTURE if mix else 0) | C.PD_AUTOMASK
    res = {}
    for impl in (C.PD_IMPL_TILE, C.PD_IMPL_GENERAL):
        ops.SWEEP_IMPL = impl
        try:
            lg, sg, Hd = logits.clone().requires_grad_(True), sigma.clone().requires_grad_(True), Hm.to(dev).requires_grad_(True)
            rgb, ph, ph_mean = ops._PlaneSweep.apply(src, tgt, lg, sg if mix else None, Hd, Rn.to(dev),
                                                     inv_K[:, :3, :3].contiguous().to(dev), None, None,
                                                     C.PD_WARP_HOMOGRAPHY, flags, 0.0)
            (ph_mean * 3.0 + (rgb * gw).sum()).backward()
            res[impl] = (lg.grad.cpu(), sg.grad.cpu() if mix else None, Hd.grad.cpu())
        finally:
            ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    new, old = res[C.PD_IMPL_TILE], res[C.PD_IMPL_GENERAL]
    assert float(old[0].abs().max()) > 0
    assert rel_err(new[0], old[0]) < 2e-6, rel_err(new[0], old[0])
    if mix:
        assert rel_err(new[1], old[1]) < 2e-6, rel_err(new[1], old[1])
    assert rel_err(new[2], old[2]) < 5e-5, rel_err(new[2], old[2])   # sums over the image in a different order



@pytest.mark.parametrize("W,H,N,side,kw", [
    (640, 12, 9, "r", dict(disp_min=2.0, disp_max=300.0)),            # three full/partial 256-pixel segments
    (258, 9, 7, "r", dict(disp_min=0.5, disp_max=120.0)),             # a segment of two pixels
    (257, 5, 5, "l", dict(disp_min=0.5, disp_max=80.0)),              # sign < 0: runs that start left of the image
    (70, 11, 10, "r", dict(special_disp=[0.0, 1.0, 2.0, 1.9999999, 3.0000002, 7.5, 68.9999, 69.0, 75.0, 1e6])),
    (130, 7, 10, "l", dict(special_disp=[0.0, 0.25, 1.0, 63.0, 64.0, 64.00001, 65.5, 127.99999, 129.0, 200.0])),
    (300, 8, 8, "r", dict(special_disp=[299.99997, 2.0000002, 1.9999998, 0.99999994, 100.0, 33.333332, 255.0, 256.00003])),
    (9, 4, 3, "r", dict(disp_min=0.3, disp_max=4.0)),
    (1280, 6, 4, "r", dict(disp_min=2.0, disp_max=300.0)),
    (200, 33, 12, "r", dict(disp_min=0.5, disp_max=60.0, n_xz=4)),    # per-row disparities + row masks
])
@pytest.mark.parametrize("mix,automask", [(True, False), (True, True), (False, True)])
@pytest.mark.parametrize("quad_bwd", [False, True])
def test_rowquad_kernels_equal_rowshift_kernels(W, H, N, side, kw, mix, automask, quad_bwd, monkeypatch):
    """The four-pixels-per-lane kernels (pd_plane_sweep_rowquad.hip: 16-byte loads and stores) against the
    one-pixel-per-lane row-shift kernels (PD_IMPL_ROWS1) on the same inputs: whole and ragged segments, both signs,
    integer and almost-integer shifts (the general routing path), shifts beyond the row, xz planes."""
    _need_experiments()
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    kw = dict(kw)
    kw.setdefault("disp_min", 0.5)
    kw.setdefault("disp_max", 9.0)
    monkeypatch.setenv("PD_QUAD_FWD", "1")     # the wide-access kernels are opt-in (NOTEBOOK.md 3.5)
    if quad_bwd:
        monkeypatch.setenv("PD_QUAD_BWD", "1")
    else:
        monkeypatch.delenv("PD_QUAD_BWD", raising=False)
    case = build_case(B=2, N=N, H=H, W=W, seed=4000 + W, sigma_interior=True, **kw)
    if "special_disp" in kw and 0.0 in kw["special_disp"]:
        automask = False   # knife edge (d) of DESIGN.md section 5
    run = dict(target_side=side, use_mixture_loss=mix, automask=automask)
    extra = dict(yz_levels=0, xz_levels=kw.get("n_xz", 0))
    quad = run_product(case, run, opt_extra=extra)
    monkeypatch.delenv("PD_QUAD_FWD", raising=False)
    monkeypatch.delenv("PD_QUAD_BWD", raising=False)
    ops.SWEEP_IMPL = C.PD_IMPL_ROWS1
    try:
        one = run_product(case, run, opt_extra=extra)
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    for k in ("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp"):
        if float(one[k].abs().max()) == 0.0:
            assert float(quad[k].abs().max()) < 1e-6, k
        else:
            assert rel_err(quad[k], one[k]) < (2e-5 if k == "g_disp_pp" else 3e-6), (k, rel_err(quad[k], one[k]))



@pytest.mark.parametrize("B,N,H,W,mix,automask,rot,zoom", [
    (2, 7, 24, 80, True, True, 0.02, 1.0), (1, 9, 33, 70, True, False, 0.15, 1.0), (2, 5, 40, 150, False, True, 0.05, 1.0),
    (1, 3, 5, 7, True, False, 0.3, 1.0), (1, 63, 192, 640, True, True, 0.01, 1.0),
    (1, 4, 30, 90, True, False, 0.05, 2.6), (1, 4, 30, 90, True, False, 0.05, 0.45)])
def test_one_kernel_plane_uniform_backward_equals_the_general_kernels(B, N, H, W, mix, automask, rot, zoom, monkeypatch):
    """The body of test_plane_uniform_homography_kernels_equal_the_general_ones with PD_UNI_FUSED set."""
    _need_experiments()
    monkeypatch.setenv("PD_UNI_FUSED", "1")
    _uniform_body(B, N, H, W, mix, automask, rot, zoom, "fused", monkeypatch)


@pytest.mark.parametrize("shape", [(2, 9, 48, 200), (8, 49, 192, 640), (3, 5, 97, 130), (1, 7, 384, 256)])
def test_row_pair_backward_equals_the_row_per_workgroup_backward(shape):
    """Round 6 (NOTEBOOK.md 11.1): the row-stream backward with row pairs (scripts/experiments/pd_rowstream_pairs.inc,
    PD_BWD_PAIRS=1 on the experiments library) against the product's one-row-per-workgroup kernel, both in one process
    (scripts/diag_kernel_ab.py --check): the same expressions on the same operands — gradients equal up to the hand-over
    atomics' order (the waves per workgroup differ), the per-plane disparity gradient up to its summation order."""
    import os
    import re
    import subprocess
    import sys
    _need_experiments()
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    exp = os.environ.get("PD_LIB") or os.path.join(root, "planedepth_amd", "lib", "libpd_experiments.so")
    B, N, H, W = shape
    env = dict(os.environ, PD_BWD_PAIRS="1")
    env.pop("PD_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "diag_kernel_ab.py"), "--check", "--rounds", "1", "--iters", "4",
                          "--batch", str(B), "--planes", str(N), "--height", str(H), "--width", str(W), "product", "pairs=" + exp],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = next(l for l in out.stdout.splitlines() if l.startswith("check pairs"))
    got = {k: (float(d), float(r)) for k, d, r in re.findall(r"(\w+) ([0-9.e+-]+) \(of ([0-9.e+-]+)\)", line)}
    assert got["rgb_rec"][0] == 0 and got["ph_map"][0] == 0, line
    for k, tol in (("g_logits", 1e-6), ("g_sigma", 1e-6), ("g_plane", 1e-5)):
        assert got[k][1] > 0 and got[k][0] <= tol * got[k][1], line
