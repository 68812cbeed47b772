"""Parity of the HIP path (through the C ABI) against the golden vectors captured from the reference and against
the oracle.  Tolerance: BASELINE.json north_star — 1e-4 relative, fp32 — measured as normalised max error
max|a-b| / max|b| per tensor (cases.rel_err)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from cases import SMALL, load_fixture, rel_err, run_oracle
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-4
NOT_YET = set()
# Ill-conditioned quantities (the oracle itself disagrees fp32 vs fp64 by O(1) on them — see DESIGN.md "knife edges"):
#  * stereo homography: every sample row sits exactly on an integer y, where d(bilinear)/dy is discontinuous, so the
#    y-rows of g_Rt depend on the last ulp of iy.
ILL_CONDITIONED = {"homo_mix_stereo": {"g_Rt"}}


@pytest.fixture(params=["exact", "fast"])
def row_mode(request):
    """Both treatments of a second source row whose bilinear weight is fp32 noise of the reference's y round trip (<= 6e-6 at
    H = 192; pd_rowshift_common.h: two_row_form): PD_IMPL_EXACT_ROWS serves it, PD_IMPL_FAST_ROWS drops it below 2^-16.
    VERDICT r4 #1a: every oracle / golden parity test of the disp path runs under both, at the same 1e-4 — which of the
    two PD_IMPL_AUTO means (pd_sweep_auto_row_eps) follows from that, not from taste.  Only the row kernels look at the
    mode; the other kernels run as under PD_IMPL_AUTO."""
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    prev = ops.SWEEP_IMPL
    ops.SWEEP_IMPL = C.PD_IMPL_FAST_ROWS if request.param == "fast" else C.PD_IMPL_EXACT_ROWS
    yield request.param
    ops.SWEEP_IMPL = prev


# Where PD_IMPL_FAST_ROWS is beyond BASELINE's 1e-4 (measured: profiles/r05_parity_rows.md — white-noise inputs at full size;
# the small fixtures, the KATs and the band-limited set hold 1e-4 in both modes).  This is the test-made decision VERDICT r4 #1a
# asked for: the mode is red at the contract's tolerance on configs[1]/[2]/[3]/[4]-sized inputs (worst: g_sigma 3.1e-4 at batch 12;
# g_logits 1.5e-2 of a 7.8e-6 range on the horizon row of the 49 + 14 plane set), so PD_IMPL_AUTO keeps every second source
# row (pd_sweep_auto_row_eps() == 0) and FAST_ROWS stays an opt-in approximation, held here to 2 x what was measured.
FAST_ROWS_TOL = {"fullsize": 2e-4, "n63_xz": 3e-2, "hr": 3e-4, "batch12": 6e-4}


def _row_tol(row_mode, label):
    return TOL if row_mode == "exact" else FAST_ROWS_TOL[label]


def _compare3(got, case, run, keys=None, tag="", skip=(), factor=2.0, caps=True):
    """Three-way bound (cases.three_way) of a product result against the oracle in fp32 (the reference's arithmetic) and
    in fp64 (the same formulas, exact to ~1e-13), plus the absolute caps against fp64 (cases.FWD_CAP / GRAD_CAP)."""
    from cases import cap_for, three_way
    ref32, exact = run_oracle(case, run), run_oracle(case, run, dtype=torch.float64)
    for k, w in exact.items():
        if k not in got or (keys and k not in keys) or k in skip:
            continue
        if float(w.abs().max()) == 0.0:
            assert float(got[k].abs().max()) < 1e-6, (tag, k)
            continue
        ok, e_got, e_ref = three_way(got[k], ref32[k], w, cap=cap_for(k) if caps else None)
        assert ok, (tag, k, e_got, e_ref)


def _compare(got, want, keys=None, tol=TOL, tag="", skip=(), zero_floor=0.0):
    """zero_floor: a reference tensor whose largest entry is below it is rounding noise around an exact zero (e.g. one
    plane: pi = 1, g_logits = 0; target == source under automask: the identity loss wins everywhere) and is compared as
    zero, not relatively."""
    for k, w in want.items():
        if k not in got or (keys and k not in keys) or k in skip:
            continue
        assert got[k].shape == w.shape, (tag, k, got[k].shape, w.shape)
        if float(w.abs().max()) <= zero_floor:
            assert float(got[k].abs().max()) < 1e-6, (tag, k)   # e.g. one plane: pi = 1, g_logits = 0 + rounding
        else:
            e = rel_err(got[k], w)
            assert e < tol, (tag, k, e)


# xy-plane fixtures whose decoder mask is all ones: with opt.xz_levels = opt.yz_levels = 0 the trainer mirror knows that by
# construction and does not hand the mask over, which puts them on the HEADLINE kernels (segment-stream forward, row-stream
# backward) instead of the per-pixel-mask row-shift kernels
XY_ONLY = ["disp_mix_r", "disp_mix_l", "disp_mix_automask", "disp_l1", "disp_l1_automask", "disp_mix_oob",
           "disp_mix_integer_d", "disp_mix_masknovel", "disp_l1_masknovel"]


@pytest.mark.parametrize("name,opt_extra", [(n, None) for n in SMALL if n not in NOT_YET] +
                         [(n, dict(xz_levels=0, yz_levels=0)) for n in XY_ONLY])
def test_fixture_vs_reference_golden(name, opt_extra, row_mode):
    """The reference-captured vectors (tests/golden/*.npz, written by make_golden.py from the imported reference) through
    the product API — once as the trainer hands them over in general (dense mask: row-shift kernels) and, for the xy-plane
    fixtures, with the options that route the same inputs to the headline kernels."""
    from gpu_cases import run_product
    case, want, run = load_fixture(name)
    got = run_product(case, run, opt_extra=opt_extra)
    _compare(got, want, tag=name, skip=ILL_CONDITIONED.get(name, ()))


@pytest.mark.parametrize("name", ["disp_mix_r", "disp_mix_l", "disp_mix_automask", "disp_l1", "disp_mix_integer_d",
                                  "disp_mix_oob"])
def test_dense_disparity_path_matches_per_plane_path(name, row_mode):
    """The same xy-plane case fed as a dense [B,N,H,W] disparity map must give the same answer."""
    from gpu_cases import run_product
    case, want, run = load_fixture(name)
    got = run_product(case, run, force_dense=True)
    _compare(got, want, tag=name + "/dense")


@pytest.mark.parametrize("seed,kw,run", [
    (201, dict(B=2, N=9, H=24, W=80, disp_min=0.5, disp_max=40.0), dict()),
    (202, dict(B=1, N=12, H=33, W=70, disp_min=0.5, disp_max=30.0, n_xz=4), dict(automask=True)),
    (203, dict(B=2, N=7, H=24, W=80, disp_min=0.5, disp_max=20.0, stereo_T=False), dict(warp_type="homography_warp")),
    (204, dict(B=1, N=7, H=24, W=80, disp_min=0.5, disp_max=20.0, stereo_T=False),
     dict(warp_type="homography_warp", use_mixture_loss=False, automask=True)),
    (205, dict(B=3, N=5, H=17, W=66, disp_min=0.5, disp_max=20.0), dict(use_mixture_loss=False, target_side="l")),
    (206, dict(B=1, N=70, H=16, W=64, disp_min=0.5, disp_max=30.0, n_xz=20), dict()),  # > 64 planes: 3 mask words
    (207, dict(B=2, N=8, H=17, W=66, disp_min=0.5, disp_max=20.0, render_probability=True),
     dict(render_probability=True, automask=True)),
    (208, dict(B=1, N=6, H=17, W=66, disp_min=0.5, disp_max=20.0, render_probability=True, n_xz=2),
     dict(render_probability=True, use_mixture_loss=False)),
    (209, dict(B=1, N=6, H=24, W=80, disp_min=0.5, disp_max=20.0, render_probability=True, stereo_T=False),
     dict(render_probability=True, warp_type="homography_warp")),
])
def test_random_cases_vs_oracle(seed, kw, run, row_mode):
    """Ragged sizes (W not a multiple of 64, odd H), many planes, against the oracle (fp32, the reference's arithmetic).

    homography_warp end to end uses the three-way bound: H_t2s = inverse(K (R + t n^T/d) K^-1) is
    an fp32 torch.inverse in the reference (LAPACK / cuSOLVER / rocSOLVER: three roundings) and cond(H) ~ 1e3-1e4 turns
    its last-ulp differences into ~1e-4-relative coordinate differences, so the product has to be as
    close to the fp64 evaluation as the fp32 oracle is.  The kernel itself is held to 1e-4 with H_t2s pinned in the
    next test.  (g_Rt of the stereo pose: ILL_CONDITIONED, every sample on an integer row.)"""
    from gpu_cases import run_product
    from planedepth_amd.synthetic import build_case
    case = build_case(seed=seed, sigma_interior=True, **kw)
    got = run_product(case, run)
    if run.get("warp_type") == "homography_warp":
        # g_Rt only where every sample sits on an integer row (the stereo pose: d bilinear / dy is discontinuous there,
        # ILL_CONDITIONED above); a pose with a rotation has a well-defined pose gradient and is held to the same bar
        _compare3(got, case, run, tag="seed%d" % seed, skip=("g_Rt",) if kw.get("stereo_T", True) else ())
    else:
        _compare(got, run_oracle(case, run), tag="seed%d" % seed)


@pytest.mark.parametrize("mix,automask", [(True, False), (True, True), (False, True)])
def test_homography_kernel_with_pinned_matrices(mix, automask):
    """Per-pixel homography path at 1e-4: the same fp32 H_t2s is handed to the HIP kernel and to the oracle."""
    from oracle import planedepth_oracle as orc
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(B=2, N=7, H=24, W=80, seed=210, disp_min=0.5, disp_max=20.0, stereo_T=False, sigma_interior=True)
    B, N, H, W = case["logits"].shape
    dist = 0.1 * 0.58 * W / case["disp_pp"][:, :, 0, 0]
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].expand(B, N, -1)
    ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
    H64, Rn64 = orc.homography_matrices(dist.double(), norm.double(), ex(case["Rt"].double()), ex(case["K"].double()),
                                        ex(case["inv_K"].double()))
    Hm = H64.float()
    # oracle (fp64 arithmetic on the fp32-rounded matrices)
    lg, sg, Hl = (case["logits"].double().requires_grad_(True), case["sigma"].double().requires_grad_(True),
                  Hm.double().requires_grad_(True))
    r = orc.warp_and_loss(case["color_l"].double(), case["color_r"].double(), lg, sg if mix else None,
                          warp_type="homography_warp", distance=dist.double(), norm=norm.double(),
                          T=case["Rt"].double(), K=case["K"].double(), inv_K=case["inv_K"].double(),
                          use_mixture_loss=mix, automask=automask, H_t2s=Hl)
    (r["ph_loss"] + (r["rgb_rec"] * case["g_rgb_rec"].double()).sum()).backward()
    # product kernel on the same matrices
    dev = "cuda"
    lgd, sgd, Hd = (case["logits"].to(dev).requires_grad_(True), case["sigma"].to(dev).requires_grad_(True),
                    Hm.to(dev).requires_grad_(True))
    flags = (C.PD_MIXTURE if mix else 0) | (C.PD_AUTOMASK if automask else 0)
    rgb, ph, _ = ops._PlaneSweep.apply(case["color_l"].to(dev), case["color_r"].to(dev), lgd, sgd if mix else None, Hd,
                                    Rn64.float().reshape(B * N, 3).to(dev), case["inv_K"][:, :3, :3].to(dev), None, None, C.PD_WARP_HOMOGRAPHY,
                                    flags, 0.0)
    (ph.mean() + (rgb * case["g_rgb_rec"].to(dev)).sum()).backward()
    assert rel_err(rgb.detach().cpu(), r["rgb_rec"].detach().float()) < TOL
    assert rel_err(ph.detach().cpu(), r["ph_map"].detach().float()) < TOL
    assert rel_err(lgd.grad.cpu(), lg.grad.float()) < TOL
    if mix:
        assert rel_err(sgd.grad.cpu(), sg.grad.float()) < TOL
    assert rel_err(Hd.grad.cpu(), Hl.grad.float()) < 2e-4


@pytest.mark.parametrize("W,side,disps", [
    (70, "r", [0.0, 1.0, 2.0, 1.9999999, 3.0000002, 7.5, 68.9999, 69.0, 75.0, 1e6]),
    (130, "l", [0.0, 0.25, 1.0, 63.0, 64.0, 64.00001, 65.5, 127.99999, 129.0, 200.0]),
    (192, "r", [299.99997, 2.0000002, 1.9999998, 0.99999994, 100.0, 33.333332, 191.0, 190.99998]),
    (64, "l", [0.5, 1.0, 31.999998, 63.0]),
    (5, "r", [0.3, 1.0, 2.7, 4.0]),
])
def test_rowshift_kernels_vs_general_kernels_and_oracle(W, side, disps):
    """The specialised row-shift kernels against the general (atomic) kernels and the oracle on disparities chosen to
    hit the rare paths: integer and almost-integer shifts (delta = -1/+1 lanes, rule 26 of the kernel guide), shifts
    >= W (nothing in view), ragged widths (partial last segment, W < 64), both signs."""
    from gpu_cases import run_product
    from planedepth_amd import ops
    from planedepth_amd import _capi as C
    from planedepth_amd.synthetic import build_case
    N = len(disps)
    case = build_case(B=2, N=N, H=11, W=W, seed=300 + W, disp_min=0.5, disp_max=9.0, special_disp=disps,
                      sigma_interior=True)
    # automask only without a zero-disparity plane: that plane reconstructs the source pixel itself, so where every
    # other plane is out of view the warped and the identity likelihoods tie to the last ulp and the min() of
    # trainer.py:734 becomes a coin toss between builds (knife edge (d) in DESIGN.md section 5)
    run = dict(target_side=side, automask=(0.0 not in disps))
    fast = run_product(case, run)
    ops.SWEEP_IMPL = C.PD_IMPL_GENERAL
    try:
        slow = run_product(case, run)
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    # fp32 oracle: with (almost-)integer shifts floor(ix) is decided by the last ulp, so only an evaluation that
    # follows the reference's fp32 op order lands on the same side (the fp64 oracle legitimately differs there).
    want = run_oracle(case, run)
    keys = ("rgb_rec", "ph_map", "g_logits", "g_sigma", "g_disp_pp")
    # g_disp_pp at 2e-4: with (almost-)integer shifts the bilinear derivative is a one-sided difference that switches
    # sides with the last ulp of the coordinate — the disparities of this test are chosen to sit exactly there
    _compare(slow, {k: want[k] for k in keys if k != "g_disp_pp"}, tag="general/W%d" % W)
    _compare(fast, {k: want[k] for k in keys if k != "g_disp_pp"}, tag="rowshift/W%d" % W)
    _compare(slow, {"g_disp_pp": want["g_disp_pp"]}, tag="general/W%d" % W, tol=2e-4)
    _compare(fast, {"g_disp_pp": want["g_disp_pp"]}, tag="rowshift/W%d" % W, tol=2e-4)
    for k in ("g_logits", "g_sigma"):  # the only deliberate difference: the eps-weighted cross-row adjoint term
        assert rel_err(fast[k], slow[k]) < 3e-5, (k, rel_err(fast[k], slow[k]))


def _stream_tol(W):
    """pd_plane_sweep_rowstream.hip irregular_tol(W): closer than this to an integer, a shift takes the general path."""
    return 2.5e-4 + 1.25e-6 * W


@pytest.mark.parametrize("W,H,N,side,mix,kw", [
    # shifts on both sides of the regular / irregular threshold around several integers, and exact integers
    (640, 6, 12, "r", True, "threshold"),
    (130, 7, 12, "l", True, "threshold"),
    (1280, 4, 12, "r", True, "threshold"),
    # ragged widths: one partial segment, W < one segment, the smallest even width, a width that is not even (falls
    # back to the row-shift backward)
    (258, 9, 7, "r", True, dict(disp_min=0.5, disp_max=120.0)),
    (70, 5, 5, "l", True, dict(disp_min=0.3, disp_max=40.0)),
    (2, 3, 2, "r", True, dict(disp_min=0.2, disp_max=1.5)),
    (257, 5, 5, "l", True, dict(disp_min=0.5, disp_max=80.0)),
    # the shapes the row-quad experiments were checked on: whole and ragged segments, both signs, integer and
    # almost-integer shifts, shifts beyond the row
    (640, 12, 9, "r", True, dict(disp_min=2.0, disp_max=300.0)),
    (70, 11, 10, "r", True, dict(special_disp=[0.0, 1.0, 2.0, 1.9999999, 3.0000002, 7.5, 68.9999, 69.0, 75.0, 1e6], disp_min=0.5, disp_max=9.0)),
    (130, 7, 9, "l", True, dict(special_disp=[0.25, 1.0, 63.0, 64.0, 64.00001, 65.5, 127.99999, 129.0, 200.0], disp_min=0.5, disp_max=9.0)),
    (300, 8, 8, "r", False, dict(special_disp=[299.99997, 2.0000002, 1.9999998, 0.99999994, 100.0, 33.333332, 255.0, 256.00003], disp_min=0.5, disp_max=9.0)),
    (1280, 6, 4, "r", True, dict(disp_min=2.0, disp_max=300.0)),      # 83 KB plain = one workgroup per CU; packed 72 KB = two
    (1024, 4, 9, "l", True, dict(disp_min=2.0, disp_max=400.0)),      # 66 KB of context per row: two 12-wave workgroups per CU
    (2048, 3, 3, "l", True, dict(disp_min=2.0, disp_max=900.0)),      # 131 KB: one 16-wave workgroup per CU
    (2600, 2, 2, "r", True, dict(disp_min=2.0, disp_max=900.0)),      # 167 KB plain: only the packed context (146 KB) fits
    (3000, 2, 2, "r", True, dict(disp_min=2.0, disp_max=900.0)),      # beyond the LDS either way: the row-shift backward takes over
    # more items than waves / fewer items than waves, L1 loss, per-row disparities with a horizon mask
    (640, 3, 1, "r", True, dict(disp_min=5.0, disp_max=5.0)),
    (130, 3, 2, "r", True, dict(special_disp=[5.0, 9.3], disp_min=0.5, disp_max=9.0)),
    (384, 8, 63, "r", False, dict(disp_min=0.5, disp_max=200.0)),
    (200, 33, 12, "r", True, dict(disp_min=0.5, disp_max=60.0, n_xz=4)),
    (200, 33, 12, "l", False, dict(disp_min=0.5, disp_max=60.0, n_xz=4)),
])
def test_rowstream_backward_equals_rowshift_backward_and_oracle(W, H, N, side, mix, kw):
    """The row-stream backward (pd_plane_sweep_rowstream.hip: lanes own aligned source slots, default) against the
    target-ordered row-shift backward (PD_IMPL_ROWS1) on the same forward, and against the fp32 oracle: shifts just inside
    and just outside the threshold that sends a plane down the general path, exact integers, both signs (negative shifts:
    the targets whose left tap is column -1 come from the epilogue), ragged and tiny widths, range boundaries of the
    waves inside a row (N * segments not a multiple of the wave count)."""
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    if kw == "threshold":
        t = _stream_tol(W)
        ks = [1.0, 7.0, 64.0, float(W // 2)]
        disps = [ks[0], ks[0] + 0.9 * t, ks[0] + 1.1 * t, ks[1] - 0.9 * t, ks[1] - 1.1 * t, ks[1] + 0.5,
                 ks[2], ks[2] + 1.1 * t, ks[2] - 1.1 * t, ks[3] + 0.9 * t, ks[3] - 0.9 * t, float(W + 3)]
        kw = dict(special_disp=disps[:N], disp_min=0.5, disp_max=9.0)
    kw = dict(kw)
    case = build_case(B=2, N=N, H=H, W=W, seed=5000 + W + H, sigma_interior=True, **kw)
    run = dict(target_side=side, use_mixture_loss=mix, automask=0.0 not in kw.get("special_disp", ()))   # knife edge (d)
    extra = dict(yz_levels=0, xz_levels=kw.get("n_xz", 0))
    new = run_product(case, run, opt_extra=extra)
    ops.SWEEP_IMPL = C.PD_IMPL_ROWS1
    try:
        old = run_product(case, run, opt_extra=extra)
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    want = run_oracle(case, run)
    # same closed form, same coordinates: what differs is the summation order of the four taps / of g_disp.  zero_floor:
    # with one plane pi = 1 and g_logits is rounding noise around an exact zero
    zf = 1e-6 if N == 1 else 0.0
    if N > 1:   # (one plane: pi = 1 exactly, every gradient through the softmax is cancellation noise — each kernel family
        # then sits 5e-5 from the oracle in its own way; the oracle comparison below is the check)
        _compare(new, {k: old[k] for k in ("g_logits", "g_sigma")}, tag="stream-vs-shift/W%d" % W, tol=3e-6)
        _compare(new, {"g_disp_pp": old["g_disp_pp"]}, tag="stream-vs-shift/W%d" % W, tol=5e-5)
    _compare(new, {k: want[k] for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma")}, tag="rowstream/W%d" % W, zero_floor=zf)
    _compare(new, {"g_disp_pp": want["g_disp_pp"]}, tag="rowstream/W%d" % W, tol=2e-4)


_FS_CASES = [
    (8, 49, 24, 640, 1.0, True, False, False),    # the headline row shape: five segments
    (3, 49, 192, 640, 1.0, True, True, False),    # every row of H = 192 (48 with two live source rows per image), B % 8 != 0
    (2, 12, 9, 640, -1.0, True, False, False),    # target "l": negative shifts (one segment per plane straddles column 0)
    (1, 5, 3, 64, 1.0, True, False, False),       # half a segment
    (2, 7, 17, 132, 1.0, False, False, False),    # ragged width (W % 128 != 0), L1 loss
    (2, 7, 5, 130, -1.0, True, True, False),      # ragged, negative shifts, automask
    (2, 63, 21, 200, 1.0, True, True, True),      # per-row disparities (xz planes), PD_DISP_ROWS
    (2, 49, 48, 1280, 1.0, True, False, False),   # wide rows: two column blocks of five segments, three rows per workgroup
    (1, 9, 7, 2048, -1.0, True, False, False),    # four column blocks of four segments (100 KB of LDS), a ragged row group
    (1, 9, 4, 1416, 1.0, True, True, False),      # twelve segments in three blocks, the last one ragged
    # regrouped rows (fwdstream_rows: linked rows share a workgroup; tests/test_row_groups.py checks the table on the CPU)
    (1, 5, 213, 64, 1.0, True, False, False),     # H % 3 != 0: the last group has one row, 71 groups
    (1, 9, 384, 640, -1.0, True, True, False),    # BASELINE configs[4]'s height: 94 linked rows
    (1, 3, 640, 64, 1.0, True, False, False),     # the largest height the table holds
    (1, 3, 643, 64, 1.0, False, False, False),    # beyond it: consecutive rows
]


# (alpha compositing, PD_RENDER_PROB: the shapes up to 24 x 640 cover it)
@pytest.mark.parametrize("B,N,H,W,sign,mix,automask,rows,render", [c + (False,) for c in _FS_CASES] +
                         [c + (True,) for c in _FS_CASES if c[2] * c[3] <= 24 * 640])
def test_segment_stream_forward_equals_the_plane_group_forward(B, N, H, W, sign, mix, automask, rows, render):
    """The segment-stream forward (pd_plane_sweep_fwdstream.hip: a wave per 128-pixel segment, two pixels per lane, 12-byte
    tap loads, one plane per iteration) against the plane-group row-shift forward (PD_IMPL_ROWS1) through the C ABI: rgb_rec,
    ph_map and the backward's stash agree to a few ulp (same expressions; the planes of a pixel are summed by one wave
    in order instead of merged plane ranges, and rows with two live source rows blend the colour row before the
    horizontal taps), mean(ph_map) to summation order — and each kernel reproduces itself bit for bit from call to call."""
    from planedepth_amd import _capi as C
    lib = C.load()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 1000 + W + N)
    mk = lambda *sh: torch.rand(*sh, generator=g).to(dev)  # noqa: E731
    src, tgt = mk(B, 3, H, W), mk(B, 3, H, W)
    logits = (torch.randn(B, N, H, W, generator=g) * 2.0).to(dev)
    sigma = (torch.rand(B, N, H, W, generator=g) * 1.2).to(dev)
    disp = (300.0 * (2.0 / 300.0) ** ((torch.arange(N, dtype=torch.float32)[None] + torch.rand(B, N, generator=g) - 0.5) / max(N - 1, 1)))
    disp = disp * (W / 640.0)
    flags = (C.PD_MIXTURE if mix else 0) | (C.PD_AUTOMASK if automask else 0) | (C.PD_RENDER_PROB if render else 0)
    dists = (torch.rand(B, N - 1, H, W, generator=g) * 2.0).to(dev) if render else None   # (trainer.py:587; a decoder output)
    if rows:
        gain = torch.linspace(0.2, 1.6, H)[None, None, :]
        plane = (disp[:, :, None] * gain).contiguous().to(dev)   # [B,N,H]
        flags |= C.PD_DISP_ROWS
    else:
        plane = disp.contiguous().to(dev)
    st = C.stream_handle(dev)
    out = {}
    for impl in (C.PD_IMPL_AUTO, C.PD_IMPL_ROWS1):
        d = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, flags, sign, impl)
        k = lib.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
        res = []
        for rep in range(2):
            rgb = torch.full((B, 3, H, W), float("nan"), device=dev)
            ph = torch.full((B, 1, H, W), float("nan"), device=dev)
            stash = torch.full((B, k, H, W), float("nan"), device=dev)
            phm = torch.full((1,), float("nan"), device=dev)
            C.check(lib.pd_plane_sweep_fwd(ctypes.byref(d), C.ptr(src), C.ptr(tgt), C.ptr(logits), C.ptr(sigma if mix else None),
                                           C.ptr(plane), None, None, None, C.ptr(dists), C.ptr(rgb), C.ptr(ph), C.ptr(phm), C.ptr(stash), st), "fwd")
            torch.cuda.synchronize()
            res.append((rgb.cpu(), ph.cpu(), stash[:, :4].cpu(), phm.cpu()))
        for a_, b_ in zip(res[0][:3], res[1][:3]):
            assert torch.equal(a_, b_), "not reproducible from call to call"
        out[impl] = res[0]
    new, old = out[C.PD_IMPL_AUTO], out[C.PD_IMPL_ROWS1]
    for name, a_, b_ in zip(("rgb_rec", "ph_map", "stash"), new[:3], old[:3]):
        assert not torch.isnan(a_).any(), name
        if name == "stash":   # (lse2, sum pi/sigma, sum pi*lap, automask flag): the flag may flip on exact ties only
            assert float((a_[:, 3] != b_[:, 3]).float().mean()) < 1e-5
            a_, b_ = a_[:, :3], b_[:, :3]
        for ch in range(a_.shape[1]):
            assert rel_err(a_[:, ch], b_[:, ch]) < 3e-6, (name, ch, rel_err(a_[:, ch], b_[:, ch]))
    assert abs(float(new[3]) - float(old[3])) <= 2e-6 * abs(float(old[3])), (float(new[3]), float(old[3]))
    assert abs(float(old[3]) - float(old[1].mean())) <= 1e-5 * abs(float(old[3]))


@pytest.mark.parametrize("kind", ["first_plane_far_below", "one_plane_far_above", "nan_logit"])
def test_fixed_reference_softmax_falls_back_on_extreme_logits(kind):
    """The segment-stream forward's softmax runs against a FIXED per-pixel reference (the first plane's scaled logit) and
    re-does a wave's planes with the rescaling accumulator when a term went beyond 2^90 (pd_plane_sweep_fwdstream.hip:
    PD_FS_FIXREF).  Logit sets that trip it — the first plane 300 below the rest; one plane 300 above — and a NaN logit
    (which must propagate like the reference's softmax, not trip anything) against the oracle and the plane-group forward."""
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(B=1, N=7, H=9, W=256, seed=77, disp_min=0.5, disp_max=30.0, sigma_interior=True)
    if kind == "first_plane_far_below":
        case["logits"][:, 0] -= 300.0
        case["logits"][:, 0, :, :40] -= 1000.0          # (and some of it below exp's range altogether)
    elif kind == "one_plane_far_above":
        case["logits"][:, 3, 2:6] += 300.0
    else:
        case["logits"][0, 2, 4, 100] = float("nan")
    extra = dict(xz_levels=0, yz_levels=0)
    got = run_product(case, {}, opt_extra=extra)
    ops.SWEEP_IMPL = C.PD_IMPL_ROWS1
    try:
        old = run_product(case, {}, opt_extra=extra)
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    if kind == "nan_logit":
        for k in ("rgb_rec", "ph_map"):
            assert torch.equal(torch.isnan(got[k]), torch.isnan(old[k])), k
            m = ~torch.isnan(old[k])
            assert rel_err(got[k][m], old[k][m]) < 3e-6, k
        return
    _compare(got, {k: old[k] for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma")}, tag=kind + " vs plane-group", tol=3e-6)
    _compare(got, run_oracle(case, {}), keys=("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp"), tag=kind)


def test_rowstream_backward_at_the_high_resolution_configuration():
    """BASELINE configs[4] (384x1280, 49 planes; one image here): the packed LDS context and 12-wave workgroups of the wide
    rows against the target-ordered row-shift backward on the same forward."""
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import survey_fullsize_case
    case = survey_fullsize_case(B=1, H=384, W=1280, sigma_interior=True)
    run = dict(automask=True)
    new = run_product(case, run)
    ops.SWEEP_IMPL = C.PD_IMPL_ROWS1
    try:
        old = run_product(case, run)
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    assert float(old["g_logits"].abs().max()) > 0
    _compare(new, {k: old[k] for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma")}, tag="hr stream-vs-shift", tol=3e-6)
    _compare(new, {"g_disp_pp": old["g_disp_pp"]}, tag="hr stream-vs-shift", tol=5e-5)


@pytest.mark.parametrize("name", ["disp_mix_xz", "disp_mix_r", "disp_mix_automask", "disp_mix_integer_d"])
def test_per_row_disparities_use_the_rowshift_kernels(name):
    """opt.yz_levels == 0 promises row-uniform disparity maps: dense maps (xz planes, horizon mask) then go through the
    row-shift kernels as [B,N,H] per-row disparities; same golden vectors, incl. the gradient w.r.t. the plane levels."""
    from gpu_cases import run_product
    case, want, run = load_fixture(name)
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    got = run_product(case, run, force_dense=True, opt_extra=dict(yz_levels=0, xz_levels=3))
    assert ops.LAST_SWEEP_FLAGS & C.PD_DISP_ROWS
    # the horizon mask of the xz planes is row-uniform too: handed over as [B,N,H] (no per-pixel mask traffic)
    assert bool(ops.LAST_SWEEP_FLAGS & C.PD_MASK_ROWS) == bool((case["padding_mask"] != 1).any() or True)
    _compare(got, want, tag=name + "/rows")


def test_fast_division_is_exact():
    """The row-shift kernels divide by (W-1) with a refined reciprocal + one correction step; the sampling position
    must not move by a single ulp relative to the reference's IEEE division, so compare bit patterns exhaustively over
    the coordinate range for the widths in BASELINE.json's configs."""
    import ctypes
    from planedepth_amd import _capi as C
    lib = C.load()
    fn = lib.pd_selftest_division
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    mism = torch.zeros(1, dtype=torch.int32, device="cuda")
    for W in (5, 16, 64, 70, 130, 192, 640, 1280, 2048):
        for lo, step in ((-2.0 * W, 1.0 / 64), (0.0, 0.013), (-300.5, 0.0071)):
            count = min(int(4 * W / step), 1 << 22)
            C.check(fn(float(W - 1), count, lo, step, C.ptr(mism), C.stream_handle()), "pd_selftest_division")
    assert int(mism.item()) == 0


@pytest.mark.parametrize("opt_extra", [None, dict(xz_levels=0, yz_levels=0)], ids=["dense-mask", "headline-kernels"])
def test_fullsize_known_answers(opt_extra, row_mode):
    """192x640, 49 planes: scalars captured from the reference (tests/golden/kat_fullsize.json) — with the
    decoder's dense all-ones mask handed over (row-shift kernels) and with the options that tell the trainer mirror the mask
    is all ones by construction (the headline kernels: segment-stream forward, row-stream backward)."""
    from gpu_cases import run_product
    from planedepth_amd.synthetic import survey_fullsize_case
    with open(os.path.join(GOLDEN, "kat_fullsize.json")) as f:
        kat = json.load(f)
    case = survey_fullsize_case()
    sums = lambda r: dict(ph_loss=float(r["ph_loss"]), sum_rgb_rec=float(r["rgb_rec"].double().sum()),  # noqa: E731
                          l1_g_logits=float(r["g_logits"].double().abs().sum()),
                          l1_g_sigma=float(r["g_sigma"].double().abs().sum()),
                          l1_g_disp_pp=float(r["g_disp_pp"].double().abs().sum()))
    for name, k in kat.items():
        got = sums(run_product(case, k["run"], opt_extra=opt_extra))
        # The reference's captured scalars at 1e-4 where its fp32 arithmetic is the thing to match (disp_warp: ph_loss,
        # sum rgb_rec, |g_logits|).  Three-way against the fp64 oracle where two fp32 evaluations legitimately differ:
        # homography_warp (the reference's own fp32 inverse moves its loss by 1.6e-4 relative to its disp_warp twin,
        # SURVEY.md H2 / BASELINE.md section 4: 0.78829277 vs 0.78841859) and the two sums that see the 1 % of sigma
        # the prescribed inputs put exactly ON the clamp bound 0.01, where the clamp's gradient gate flips with the
        # last ulp of the interpolated value.
        homo = k["run"].get("warp_type") == "homography_warp"
        exact = None
        for key, v in got.items():
            ref = k[key]
            if not ref:
                continue
            if not homo and key in ("ph_loss", "sum_rgb_rec", "l1_g_logits"):
                assert abs(v - ref) <= TOL * abs(ref), (name, key, v, ref)
                continue
            if exact is None:
                exact = sums(run_oracle(case, k["run"], dtype=torch.float64))
            assert abs(v - exact[key]) <= 2.0 * abs(ref - exact[key]) + TOL * abs(exact[key]), (name, key, v, ref, exact[key])


def test_fullsize_vs_oracle_tensors(row_mode):
    """Every output tensor and gradient at the BASELINE size (B=1) against the oracle run on the host."""
    from gpu_cases import run_product
    from planedepth_amd.synthetic import survey_fullsize_case
    case = survey_fullsize_case(sigma_interior=True)
    for run in (dict(), dict(automask=True, use_mixture_loss=True), dict(use_mixture_loss=False)):
        got = run_product(case, run)
        want = run_oracle(case, run)  # fp32: at x ~ 600 the fp32 coordinate rounding of the reference itself moves
        # results by 1e-4..3e-2 relative to exact arithmetic (scripts/diag_errors.py), so "the reference's fp32
        # arithmetic" is the thing to match, as BASELINE.json's north_star asks.
        # (g_sigma carries 1/sigma^3-type amplification — the fp32 oracle is 2.6e-4 away from the fp64 value — and was
        # held to 2e-4 in round 1; measured 3.4e-6 against the fp32 oracle now: profiles/r02_parity.md)
        _compare(got, want, keys=("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp"), tag=str(run),
                 tol=_row_tol(row_mode, "fullsize"))


def _band_limited(shape, gen, cutoff=0.12, lo=0.0, hi=1.0):
    """White noise low-passed in the Fourier domain (keeps |f| <= cutoff of Nyquist per axis) and rescaled to [lo, hi]:
    neighbouring rows / columns are strongly correlated, as network outputs and photographs are and white noise is not
    (SURVEY.md H2's second parity set)."""
    x = torch.randn(shape, generator=gen)
    H, W = shape[-2:]
    fy = torch.fft.fftfreq(H).abs()[:, None] * 2.0
    fx = torch.fft.rfftfreq(W).abs()[None, :] * 2.0
    keep = ((fy <= cutoff) & (fx <= cutoff)).to(x.dtype)
    x = torch.fft.irfft2(torch.fft.rfft2(x) * keep, s=(H, W))
    flat = x.flatten(-2)
    mn, mx = flat.min(-1)[0][..., None, None], flat.max(-1)[0][..., None, None]
    return lo + (hi - lo) * (x - mn) / (mx - mn)


@pytest.mark.parametrize("automask", [False, True])
def test_band_limited_inputs_fullsize_vs_oracle(automask, row_mode):
    """SURVEY.md H2's second parity set at the BASELINE size: band-limited images, logits and sigma (smooth like network
    outputs: the difference between neighbouring source rows is not white noise), both row modes, every tensor at 1e-4
    against the fp32 oracle."""
    from gpu_cases import run_product
    from planedepth_amd.synthetic import survey_fullsize_case
    case = survey_fullsize_case(sigma_interior=True)
    g = torch.Generator().manual_seed(99)
    B, N, H, W = case["logits"].shape
    case["color_l"] = _band_limited((B, 3, H, W), g)
    case["color_r"] = _band_limited((B, 3, H, W), g)
    case["logits"] = _band_limited((B, N, H, W), g, lo=-3.0, hi=3.0)
    case["sigma"] = _band_limited((B, N, H, W), g, lo=0.02, hi=0.9)
    run = dict(automask=automask)
    got = run_product(case, run, opt_extra=dict(xz_levels=0, yz_levels=0))
    want = run_oracle(case, run)
    _compare(got, want, keys=("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp"), tag="band/%s" % row_mode)


def test_fast_rows_mode_is_opt_in_and_bounded():
    """PD_IMPL_FAST_ROWS drops a second source row whose bilinear weight is below 2^-16 (fp32 noise of the reference's y
    round trip, a quarter of the rows at H=192); PD_IMPL_EXACT_ROWS keeps it (rounding-level agreement with the fp32
    oracle).  Both modes are distinct code paths, both inside the parity budget on random data; PD_IMPL_AUTO is one of the
    two, or a threshold in between, and the library says which (pd_sweep_auto_row_eps)."""
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import survey_fullsize_case
    case = survey_fullsize_case(sigma_interior=True)
    want = run_oracle(case, dict())
    keys = ("rgb_rec", "ph_map", "g_logits", "g_disp_pp")
    res = {}
    for impl in (C.PD_IMPL_EXACT_ROWS, C.PD_IMPL_FAST_ROWS, C.PD_IMPL_AUTO):
        ops.SWEEP_IMPL = impl
        try:
            res[impl] = run_product(case, dict())
        finally:
            ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    exact, fast, auto = res[C.PD_IMPL_EXACT_ROWS], res[C.PD_IMPL_FAST_ROWS], res[C.PD_IMPL_AUTO]
    _compare(exact, want, keys=keys, tag="exact_rows", tol=1.5e-5)
    _compare(fast, want, keys=keys, tag="fast_rows", tol=1e-4)
    _compare(fast, want, keys=("g_sigma",), tag="fast_rows", tol=FAST_ROWS_TOL["fullsize"])
    assert not torch.equal(fast["rgb_rec"], exact["rgb_rec"])  # the two modes really are different code paths
    eps = C.load().pd_sweep_auto_row_eps()
    if eps == 0.0 or eps >= 2.0 ** -16:   # AUTO is one of the two pure modes
        same_as = fast if eps > 0.0 else exact
        for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma"):
            assert torch.equal(auto[k], same_as[k]), k
        assert rel_err(auto["g_disp_pp"], same_as["g_disp_pp"]) < 1e-5   # (summed with float atomics, LDS and global: order-dependent rounding)
    else:
        _compare(auto, want, keys=keys + ("g_sigma",), tag="auto", tol=1e-4)


def test_size_independent_properties_at_benchmark_size():
    """B=8, 192x640, N=49 (BASELINE.json configs[1]) — properties that need no oracle run."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import survey_fullsize_case
    c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in survey_fullsize_case(B=8).items()}
    B, N, H, W = c["logits"].shape
    disp = c["disp_pp"].expand(B, N, H, W)

    def run(src, tgt, logits, sigma, side="r", g_scale=1.0, perm=None):
        lg, sg, dp = logits.clone().requires_grad_(True), sigma.clone().requires_grad_(True), c["disp_pp"].clone().requires_grad_(True)
        if perm is not None:
            dpx = dp[perm]
        else:
            dpx = dp
        rgb, ph = ops.plane_sweep_disp(src, tgt, lg, sg, dpx.expand(B, N, H, W), c["padding_mask"], target_side=side)
        ((ph.mean() + (rgb * c["g_rgb_rec"]).sum()) * g_scale).backward()
        return rgb.detach(), ph.detach(), lg.grad, sg.grad, dp.grad

    rgb, ph, gl, gs, gd = run(c["color_l"], c["color_r"], c["logits"], c["sigma"])
    # (1) forward is bit-deterministic; composite is a convex combination of (zero-padded) source colours
    rgb2, ph2, *_ = run(c["color_l"], c["color_r"], c["logits"], c["sigma"])
    assert torch.equal(rgb, rgb2) and torch.equal(ph, ph2)
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= float(c["color_l"].max()) * (1 + 1e-6)
    assert torch.isfinite(ph).all() and torch.isfinite(gl).all() and torch.isfinite(gs).all()
    # (2) batch elements are independent: permuting the batch permutes the results
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device="cuda")
    rgbp, php, glp, gsp, gdp = run(c["color_l"][perm], c["color_r"][perm], c["logits"][perm], c["sigma"][perm], perm=perm)
    assert torch.equal(rgbp, rgb[perm]) and torch.equal(php, ph[perm])
    # (3) backward is linear in the upstream gradient
    _, _, gl3, gs3, gd3 = run(c["color_l"], c["color_r"], c["logits"], c["sigma"], g_scale=3.0)
    assert rel_err(gl3, 3 * gl) < 1e-5 and rel_err(gs3, 3 * gs) < 1e-5 and rel_err(gd3, 3 * gd) < 1e-4
    # (4) mirror symmetry: target "l" on horizontally flipped images == flipped target "r"
    f = lambda t: t.flip(-1)  # noqa: E731
    rgbf, phf, glf, gsf, _ = run(f(c["color_l"]), f(c["color_r"]), f(c["logits"]), f(c["sigma"]), side="l")
    # g_rgb_rec is not flipped, so compare forward tensors only (rounding of the coordinates differs slightly)
    assert rel_err(f(rgbf), rgb) < 2e-3 and rel_err(f(phf), ph) < 2e-3  # fp32 coordinate rounding at x~600: ulp 6e-5 px


def test_rowstream_backward_properties_at_benchmark_size():
    """B=8, 192x640, N=49 WITHOUT a padding mask — the benchmark path itself (row-shift forward, row-stream backward):
    properties that need no oracle run, and the target-ordered row-shift backward on the same forward as a second
    implementation of the same adjoint."""
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import survey_fullsize_case
    c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in survey_fullsize_case(B=8, sigma_interior=True).items()}
    B, N, H, W = c["logits"].shape

    def run(perm=None, g_scale=1.0):
        sel = (lambda t: t[perm]) if perm is not None else (lambda t: t)
        lg, sg = sel(c["logits"]).clone().requires_grad_(True), sel(c["sigma"]).clone().requires_grad_(True)
        dp = sel(c["disp_pp"]).clone().requires_grad_(True)
        rgb, ph, pm = ops.plane_sweep_disp(sel(c["color_l"]), sel(c["color_r"]), lg, sg, dp.expand(B, N, H, W), None,
                                           return_mean=True)
        ((pm + (rgb * sel(c["g_rgb_rec"])).sum()) * g_scale).backward()
        return lg.grad, sg.grad, dp.grad

    gl, gs, gd = run()
    assert torch.isfinite(gl).all() and torch.isfinite(gs).all() and torch.isfinite(gd).all()
    # (1) every element of g_logits / g_sigma is stored exactly once (the few range hand-overs add one value onto a stored
    #     slot): bit-identical between runs
    gl2, gs2, _ = run()
    assert torch.equal(gl, gl2) and torch.equal(gs, gs2)
    # (2) images are independent: permuting the batch permutes the gradients, bit for bit
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device="cuda")
    glp, gsp, gdp = run(perm=perm)
    assert torch.equal(glp, gl[perm]) and torch.equal(gsp, gs[perm])
    assert rel_err(gdp, gd[perm]) < 1e-5     # (sums over the image: the order of the waves' partial sums is not fixed)
    # (3) linear in the upstream gradient
    gl3, gs3, gd3 = run(g_scale=3.0)
    assert rel_err(gl3, 3 * gl) < 1e-6 and rel_err(gs3, 3 * gs) < 1e-6 and rel_err(gd3, 3 * gd) < 1e-5
    # (4) the target-ordered backward (one pixel per lane, shifted stores, in-wave routing) computes the same adjoint
    ops.SWEEP_IMPL = C.PD_IMPL_ROWS1
    try:
        glo, gso, gdo = run()
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    assert rel_err(gl, glo) < 3e-6 and rel_err(gs, gso) < 3e-6 and rel_err(gd, gdo) < 5e-5, (
        rel_err(gl, glo), rel_err(gs, gso), rel_err(gd, gdo))


def test_modules_vs_reference_golden():
    import planedepth_amd as pa
    from planedepth_amd import ops
    z = {k: torch.from_numpy(v).cuda() for k, v in np.load(os.path.join(GOLDEN, "modules.npz")).items()}
    B, _, H, W = z["bp_depth"].shape
    cam = pa.BackprojectDepth(H, W)(z["bp_depth"], z["bp_inv_K"])
    assert rel_err(cam.cpu(), z["bp_cam"].cpu()) < TOL
    grid = pa.Project3D(H, W)(cam, z["bp_K"], z["bp_T"])
    assert rel_err(grid.cpu(), z["pj_grid"].cpu()) < TOL
    N = z["hw_d"].shape[1]
    ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
    hgrid, hmask = pa.HomographyWarp(H, W)(z["hw_d"], z["hw_n"], ex(z["bp_T"]), ex(z["bp_K"]), ex(z["bp_inv_K"]))
    assert rel_err(hgrid.cpu(), z["hw_grid"].cpu()) < TOL
    assert hmask.dtype == torch.bool and torch.equal(hmask.float().cpu(), z["hw_mask"].cpu())
    assert rel_err(pa.SSIM()(z["ssim_x"], z["ssim_y"]).cpu(), z["ssim_out"].cpu()) < TOL
    x = z["ssim_x"].clone().requires_grad_(True)
    ns = type("S", (), {"opt": type("O", (), {"use_ssim": True})()})()
    rl = pa.compute_reprojection_loss(ns, x, z["ssim_y"])
    assert rel_err(rl.cpu(), z["reproj_ssim"].cpu()) < TOL
    (rl * z["reproj_gw"]).sum().backward()
    assert rel_err(x.grad.cpu(), z["reproj_g_pred"].cpu()) < TOL
    ns.opt.use_ssim = False
    assert rel_err(pa.compute_reprojection_loss(ns, z["ssim_x"], z["ssim_y"]).cpu(), z["reproj_l1"].cpu()) < TOL
    assert rel_err(pa.multimodal_loss(z["mm_err"], z["mm_sigma"], z["mm_pi"], dist="lap").cpu(), z["mm_lap"].cpu()) < TOL
    assert rel_err(pa.multimodal_loss(z["mm_err"], z["mm_sigma"], z["mm_pi"]).cpu(), z["mm_gauss"].cpu()) < TOL
    from oracle import planedepth_oracle as orc
    for dist in ("lap", "gaussian"):  # backward of the standalone mixture NLL against autograd through the oracle
        e64, s64, p64 = (z[k].cpu().double().requires_grad_(True) for k in ("mm_err", "mm_sigma", "mm_pi"))
        gw = torch.rand(z["mm_err"].shape[0], 1, *z["mm_err"].shape[2:], dtype=torch.float64)
        (orc.multimodal_loss(e64, s64, p64, dist) * gw).sum().backward()
        eg, sg, pg = (z[k].clone().requires_grad_(True) for k in ("mm_err", "mm_sigma", "mm_pi"))
        (pa.multimodal_loss(eg, sg, pg, dist=dist) * gw.float().cuda()).sum().backward()
        assert rel_err(eg.grad.cpu(), e64.grad.float()) < TOL and rel_err(sg.grad.cpu(), s64.grad.float()) < TOL
        assert rel_err(pg.grad.cpu(), p64.grad.float()) < TOL
    assert rel_err(ops.grid_sample(z["ssim_x"], z["pj_grid"], padding_mode="border").cpu(), z["gs_border"].cpu()) < TOL
    assert rel_err(ops.grid_sample(z["ssim_x"], z["pj_grid"], padding_mode="zeros").cpu(), z["gs_zeros"].cpu()) < TOL


def test_geometry_and_sampling_gradients_vs_oracle():
    """Backward of the standalone modules against autograd through the oracle (fp64)."""
    import planedepth_amd as pa
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics, small_pose
    g = torch.Generator().manual_seed(5)
    B, H, W, C = 2, 13, 22, 3
    K, inv_K = intrinsics(B, H, W)
    T = small_pose(g, B, rot=0.05, trans=0.2)
    depth = torch.rand(B, 1, H, W, generator=g) * 5 + 0.5
    img = torch.rand(B, C, H, W, generator=g)
    gw = torch.randn(B, C, H, W, generator=g)
    from cases import three_way

    def chain(dt, pad):   # the reference's operators (oracle restatement) in the given precision
        dd, TT, im = (v.detach().clone().to(dt).requires_grad_(True) for v in (depth, T, img))
        cam = orc.backproject_depth(dd, inv_K.to(dt))
        grid = orc.project_3d(cam, K.to(dt), TT, H, W)
        (orc.bilinear_sample(im, grid, pad) * gw.to(dt)).sum().backward()
        return dd, TT, im, grid

    for pad in ("zeros", "border"):
        d64, T64, im64, grid = chain(torch.float64, pad)
        d32, T32, _, _ = chain(torch.float32, pad)
        dg, Tg, ig = depth.cuda().requires_grad_(True), T.cuda().requires_grad_(True), img.cuda().requires_grad_(True)
        camg = pa.BackprojectDepth(H, W)(dg, inv_K.cuda())
        gridg = pa.Project3D(H, W)(camg, K.cuda(), Tg)
        out = ops.grid_sample(ig, gridg, padding_mode=pad)
        (out * gw.cuda()).sum().backward()
        assert rel_err(gridg.detach().cpu(), grid.detach().float()) < TOL
        # gradients THROUGH the fp32 sampling coordinates (bilinear derivative = difference of neighbours / rounding of
        # the position): three-way against the same chain in torch fp32
        assert three_way(dg.grad.cpu(), d32.grad, d64.grad)[0], (pad, three_way(dg.grad.cpu(), d32.grad, d64.grad))
        assert three_way(Tg.grad.cpu(), T32.grad, T64.grad)[0], (pad, three_way(Tg.grad.cpu(), T32.grad, T64.grad))
        assert rel_err(ig.grad.cpu(), im64.grad.float()) < TOL, pad
    # HomographyWarp backward
    N = 3
    d = torch.rand(B, N, generator=g) * 4 + 0.3
    n = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    gg = torch.randn(B * N, H, W, 2, generator=g)
    ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
    def hchain(dt):
        dd, TT = d.detach().clone().to(dt).requires_grad_(True), T.detach().clone().to(dt).requires_grad_(True)
        gr, _ = orc.homography_grid(dd, n.to(dt), ex(TT), ex(K.to(dt)), ex(inv_K.to(dt)), H, W)
        (gr * gg.to(dt)).sum().backward()
        return dd, TT

    (d64, T64), (d32, T32) = hchain(torch.float64), hchain(torch.float32)
    dg, Tg = d.cuda().requires_grad_(True), T.cuda().requires_grad_(True)
    gridg, _ = pa.HomographyWarp(H, W)(dg, n.cuda(), ex(Tg), ex(K.cuda()), ex(inv_K.cuda()))
    (gridg * gg.cuda()).sum().backward()
    assert three_way(dg.grad.cpu(), d32.grad, d64.grad)[0], three_way(dg.grad.cpu(), d32.grad, d64.grad)
    assert three_way(Tg.grad.cpu(), T32.grad, T64.grad)[0], three_way(Tg.grad.cpu(), T32.grad, T64.grad)
    # SSIM backward w.r.t. both inputs
    x, y = torch.rand(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g)
    x64, y64 = x.double().requires_grad_(True), y.double().requires_grad_(True)
    (orc.ssim(x64, y64) * gw.double()).sum().backward()
    xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    (pa.SSIM()(xg, yg) * gw.cuda()).sum().backward()
    assert rel_err(xg.grad.cpu(), x64.grad.float()) < TOL
    assert rel_err(yg.grad.cpu(), y64.grad.float()) < TOL


def test_only_one_hip_runtime_and_native_library_loaded():
    """The in-tree .so must be the thing that ran, on the HIP runtime torch already holds."""
    from planedepth_amd import _capi
    _capi.load()
    maps = open("/proc/self/maps").read()
    assert "libplanedepth_hip.so" in maps
    hips = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(hips) == 1, hips


# ---------------------------------------------------------------------------------------------------------------------
# fused decoder tail
# ---------------------------------------------------------------------------------------------------------------------
def _tail_group(tag):
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "decoder_tail.npz"))
    return {k.split("/", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/")}


@pytest.mark.parametrize("tag", ["mix_xz", "mix_xy", "l1_xy"])
def test_decoder_tail_against_reference_vectors(tag):
    """HIP decoder tail (forward, lazily materialised pi / probability, backward) against what the reference
    DepthDecoder produced from the same conv outputs (tests/golden/decoder_tail.npz)."""
    from planedepth_amd.decoder_tail import fused_decoder_tail
    z = _tail_group(tag)
    mix = bool(int(z["mixture"]))
    dev = "cuda"
    rl = z["raw_logits"].to(dev).requires_grad_(True)
    rs = z["raw_sigma"].to(dev).requires_grad_(True)
    dl = z["disp_layered"].to(dev).requires_grad_(True)
    has_mask = bool((z["padding_mask"] != 1).any())
    outputs = {"disp_layered": dl, "padding_mask": z["padding_mask"].to(dev)}
    fused_decoder_tail(outputs, rl, rs if mix else None, use_mixture_loss=mix, all_ones_mask=not has_mask)
    assert tuple(outputs["probability"].shape) == tuple(rl.shape)
    for k in ("logits", "disp", "depth") + (("sigma",) if mix else ()):
        assert rel_err(outputs[k].detach().cpu(), z[k]) < TOL, (tag, k)
    assert rel_err(outputs["probability"].tensor().cpu(), z["probability"]) < TOL
    if mix:
        assert rel_err(outputs["pi"].tensor().cpu(), z["pi"]) < TOL
    obj = (outputs["logits"] * z["gw_logits"].to(dev)).sum() + (outputs["disp"] * z["gw_disp"].to(dev)).sum() + \
          (outputs["depth"] * z["gw_depth"].to(dev)).sum()
    if mix:
        obj = obj + (outputs["sigma"] * z["gw_sigma"].to(dev)).sum()
    obj.backward()
    assert rel_err(rl.grad.cpu(), z["g_raw_logits"]) < TOL, rel_err(rl.grad.cpu(), z["g_raw_logits"])
    if mix:
        assert rel_err(rs.grad.cpu(), z["g_raw_sigma"]) < TOL, rel_err(rs.grad.cpu(), z["g_raw_sigma"])
    if "g_disp_layered" in z:
        assert rel_err(dl.grad.cpu(), z["g_disp_layered"]) < TOL, rel_err(dl.grad.cpu(), z["g_disp_layered"])


@pytest.mark.parametrize("tag", ["mix_xy", "mix_xz", "l1_xy"])
def test_plade_tail_against_reference_vectors(tag):
    """HIP PladeNet tail with --render_probability (forward, lazily materialised pi / probability, backward) against what
    the reference PladeNet produced from the same conv outputs (tests/golden/plade_tail.npz, make_golden.plade_tail_vectors:
    networks/plade_net.py:309-341 run by the reference itself).  mix_xz: ground planes behind the frontal ones — the depth
    layers are not sorted there, distances go negative and the reference's compositing weights leave [0, 1] by orders of
    magnitude; the kernels follow the same arithmetic, so the vectors still have to match."""
    from planedepth_amd.decoder_tail import fused_plade_tail
    raw = np.load(os.path.join(GOLDEN, "plade_tail.npz"))
    z = {k.split("/", 1)[1]: torch.from_numpy(raw[k]) for k in raw.files if k.startswith(tag + "/")}
    mix = bool(int(z["mixture"]))
    dev = "cuda"
    rl = z["raw_logits"].to(dev).requires_grad_(True)
    rs = z["raw_sigma"].to(dev).requires_grad_(True)
    dl = z["disp_layered"].to(dev).requires_grad_(True)
    outputs = {"disp_layered": dl}
    fused_plade_tail(outputs, rl, rs if mix else None, use_mixture_loss=mix)
    B, N, H, W = z["disp_layered"].shape
    assert tuple(outputs["probability"].shape) == (B, N, H, W) and tuple(outputs["logits"].shape) == (B, N, H, W)
    for k in ("logits", "dists", "disp", "depth") + (("sigma",) if mix else ()):
        assert rel_err(outputs[k].detach().cpu(), z[k]) < TOL, (tag, k, rel_err(outputs[k].detach().cpu(), z[k]))
    assert rel_err(outputs["probability"].tensor().cpu(), z["probability"]) < TOL
    if mix:
        assert rel_err(outputs["pi"].tensor().cpu(), z["pi"]) < TOL
    obj = (outputs["logits"] * z["gw_logits"].to(dev)).sum() + (outputs["dists"] * z["gw_dists"].to(dev)).sum() + \
          (outputs["disp"] * z["gw_disp"].to(dev)).sum() + (outputs["depth"] * z["gw_depth"].to(dev)).sum()
    if mix:
        obj = obj + (outputs["sigma"] * z["gw_sigma"].to(dev)).sum()
    obj.backward()
    assert rel_err(rl.grad.cpu(), z["g_raw_logits"]) < TOL, rel_err(rl.grad.cpu(), z["g_raw_logits"])
    if mix:
        assert rel_err(rs.grad.cpu(), z["g_raw_sigma"]) < TOL, rel_err(rs.grad.cpu(), z["g_raw_sigma"])
    if "g_disp_layered" in z:
        assert rel_err(dl.grad.cpu(), z["g_disp_layered"]) < TOL, rel_err(dl.grad.cpu(), z["g_disp_layered"])


@pytest.mark.parametrize("mix,shape", [(True, (2, 49, 24, 80)), (False, (3, 9, 17, 33)), (True, (1, 2, 5, 7))])
def test_plade_tail_vs_oracle_per_plane_disparities(mix, shape):
    """The PladeNet tail at other shapes (49 planes; H*W not a multiple of 4: one pixel per thread; the two-plane minimum)
    with the network's per-plane levels as an expanded [B,N,1,1] view (the [B,N] disparity gradient goes through the
    workspace reduction), against the oracle in fp32 and — gradients — three-way against its fp64 evaluation."""
    from planedepth_amd.decoder_tail import fused_plade_tail
    from oracle import planedepth_oracle as orc
    B, N, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    rl0 = torch.randn(B, N - 1, H, W, generator=g) * 1.5
    rs0 = torch.randn(B, N, H, W, generator=g) * 2.0
    lv = torch.arange(N, dtype=torch.float32)[None, :, None, None] + torch.rand(B, N, 1, 1, generator=g) - 0.5
    dl0 = 300.0 * (W / 640.0) * (2.0 / 300.0) ** (lv / max(N - 1, 1))      # plade_net.py:285, descending disparity = ascending depth
    gws = [torch.randn(B, N, H, W, generator=g), torch.randn(B, N - 1, H, W, generator=g) * 0.05, torch.randn(B, N, H, W, generator=g),
           torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g) * 0.1]

    def run(device, dtype, fused):
        rl, rs, d0 = (t.detach().clone().to(device=device, dtype=dtype).requires_grad_(True) for t in (rl0, rs0, dl0))
        dl = d0.expand(-1, -1, H, W)
        if fused:
            o = {"disp_layered": dl}
            fused_plade_tail(o, rl, rs if mix else None, use_mixture_loss=mix)
        else:
            o = orc.plade_tail(rl, rs if mix else None, dl, W, orc.camera_ray_norm(H, W, dtype), mix)
        w = [t.to(device=device, dtype=dtype) for t in gws]
        obj = (o["logits"] * w[0]).sum() + (o["dists"] * w[1]).sum() + (o["disp"] * w[3]).sum() + (o["depth"] * w[4]).sum()
        if mix:
            obj = obj + (o["sigma"] * w[2]).sum()
        obj.backward()
        res = {k: o[k].detach().cpu().float() for k in ("logits", "dists", "disp", "depth") + (("sigma",) if mix else ())}
        res.update(g_raw_logits=rl.grad.cpu().float(), g_disp_pp=d0.grad.cpu().float())
        if mix:
            res["g_raw_sigma"] = rs.grad.cpu().float()
        return res

    got, want, exact = run("cuda", torch.float32, True), run("cpu", torch.float32, False), run("cpu", torch.float64, False)
    for k in want:
        if k.startswith("g_"):   # three-way: the fp32 oracle's own distance from fp64 is the yardstick for the gradients
            e_got, e_ref = rel_err(got[k], exact[k]), rel_err(want[k], exact[k])
            assert e_got <= 2.0 * e_ref + TOL, (k, e_got, e_ref)
        else:
            assert rel_err(got[k], want[k]) < TOL, (k, rel_err(got[k], want[k]))


def test_plade_tail_feeds_the_compositing_sweep():
    """PladeNet's fused tail as the producer of what the sweep's --render_probability branch consumes (trainer.py:584-591):
    conv outputs -> fused_plade_tail -> pred_novel_images + compute_losses (alpha compositing over the warped logits with the
    tail's `dists`) -> backward to the conv outputs and the plane levels, against the same chain through the oracle
    (plade_tail -> warp_and_loss) in fp32, gradients three-way against its fp64 evaluation."""
    import types
    import planedepth_amd
    from planedepth_amd.decoder_tail import fused_plade_tail
    from oracle import planedepth_oracle as orc
    B, N, H, W = 2, 9, 24, 80
    g = torch.Generator().manual_seed(31)
    rl0, rs0 = torch.randn(B, N - 1, H, W, generator=g) * 1.5, torch.randn(B, N, H, W, generator=g) * 0.8
    lv = torch.arange(N, dtype=torch.float32)[None, :, None, None] + torch.rand(B, N, 1, 1, generator=g) - 0.5
    d00 = 40.0 * (0.5 / 40.0) ** (lv / (N - 1))
    cl, cr = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    gw = torch.randn(B, 3, H, W, generator=g) * 0.05

    def run(device, dtype, fused):
        rl, rs, d0 = (t.detach().clone().to(device=device, dtype=dtype).requires_grad_(True) for t in (rl0, rs0, d00))
        dl = d0.expand(-1, -1, H, W)
        a, b, w = (t.to(device=device, dtype=dtype) for t in (cl, cr, gw))
        if fused:
            outputs = {"disp_layered": dl, "padding_mask": None}
            fused_plade_tail(outputs, rl, rs)
            opt = types.SimpleNamespace(warp_type="disp_warp", match_aug=False, use_mixture_loss=True, automask=False,
                                        render_probability=True, alpha_pc=0.0, alpha_self=0.0, self_distillation=0.0,
                                        gamma_smooth=2.0, alpha_smooth=0.0, use_ssim=False, xz_levels=0, yz_levels=0)
            ns = types.SimpleNamespace(opt=opt, target_sides=["r"], perceptual_loss=lambda *x, **k: torch.zeros((), device=device))
            planedepth_amd.pred_novel_images(ns, {("color", "l"): a, ("color", "r"): b}, outputs)
            rgb, ph = outputs[("rgb_rec", "r")], outputs[("ph_mean", "r")]
        else:
            o = orc.plade_tail(rl, rs, dl, W, orc.camera_ray_norm(H, W, dtype), True)
            r = orc.warp_and_loss(a, b, o["logits"], o["sigma"], warp_type="disp_warp", target_side="r", disp_layered=dl,
                                  padding_mask=torch.ones_like(dl), distance=None, norm=None, T=None, K=None, inv_K=None,
                                  use_mixture_loss=True, automask=False, render_probability=True, dists=o["dists"])
            rgb, ph = r["rgb_rec"], r["ph_loss"]
        (ph + (rgb * w).sum()).backward()
        return dict(rgb_rec=rgb.detach().cpu().float(), ph_loss=ph.detach().cpu().float(), g_raw_logits=rl.grad.cpu().float(),
                    g_raw_sigma=rs.grad.cpu().float(), g_levels=d0.grad.cpu().float())

    got, want, exact = run("cuda", torch.float32, True), run("cpu", torch.float32, False), run("cpu", torch.float64, False)
    for k in want:
        if k.startswith("g_"):
            e_got, e_ref = rel_err(got[k], exact[k]), rel_err(want[k], exact[k])
            assert e_got <= 2.0 * e_ref + TOL, (k, e_got, e_ref)
        else:
            assert rel_err(got[k], want[k]) < TOL, (k, rel_err(got[k], want[k]))


@pytest.mark.parametrize("mix,mask,shape", [(True, False, (2, 49, 24, 80)), (True, True, (2, 63, 20, 72)),
                                            (False, False, (3, 9, 17, 33))])
def test_decoder_tail_vs_oracle_per_plane_disparities(mix, mask, shape):
    """The decoder's usual case — disp_layered an expanded view of [B,N,1,1] levels that need a gradient
    (--plane_residual) — at the plane counts of BASELINE.json, against autograd through the oracle."""
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    B, N, H, W = shape
    g = torch.Generator().manual_seed(5 + N)
    rl = torch.randn(B, N, H, W, generator=g) * 2.5
    rs = torch.randn(B, N, H, W, generator=g) * 3 - 1
    lv = torch.arange(N, dtype=torch.float32)[None, :, None, None] + torch.rand(B, N, 1, 1, generator=g) - 0.5
    pm = (torch.rand(B, N, H, W, generator=g) > 0.2).float() if mask else None
    if mask:
        pm[:, :5] = 1.0  # never mask every plane of a pixel
    gw = [torch.randn(B, N, H, W, generator=g), torch.randn(B, N, H, W, generator=g),
          torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g) * 0.1]

    def run(device, fused):
        a, s, l = (t.to(device).clone().requires_grad_(True) for t in (rl, rs, lv))
        dl = (300.0 * (2.0 / 300.0) ** (l / (N - 1))).expand(-1, -1, H, W)
        m = None if pm is None else pm.to(device)
        if fused:
            logits, sigma, disp, depth, layers = ops.decoder_tail(a, s if mix else None, m, dl, use_mixture_loss=mix)
            prob = layers()[1]
        else:
            o = orc.decoder_tail(a, s, m if m is not None else torch.ones_like(a), dl, W, use_mixture_loss=mix)
            logits, sigma, disp, depth, prob = o["logits"], o.get("sigma"), o["disp"], o["depth"], o["probability"]
        w = [t.to(device) for t in gw]
        obj = (logits * w[0]).sum() + (disp * w[2]).sum() + (depth * w[3]).sum()
        if mix:
            obj = obj + (sigma * w[1]).sum()
        obj.backward()
        res = dict(logits=logits, disp=disp, depth=depth, prob=prob, g_rl=a.grad, g_lv=l.grad)
        if mix:
            res.update(sigma=sigma, g_rs=s.grad)
        return {k: v.detach().cpu() for k, v in res.items()}

    got, want = run("cuda", True), run("cpu", False)
    for k in want:
        assert rel_err(got[k], want[k]) < TOL, (k, rel_err(got[k], want[k]))


@pytest.mark.parametrize("case", ["decoder_tail_npz", "many_planes_left_view", "clamp_bounds_and_integer_shift"])
def test_sweep_backward_applies_the_fused_decoder_tail(case):
    """SURVEY 8f rank 1, second half ("removes the round-trips"): with ``fused_decoder_tail(..., fuse_sweep_backward=True)`` the
    row-stream backward applies the decoder tail's backward on the values it holds (pd_plane_sweep_bwd_tail) and hands autograd
    the gradients of the decoder's CONV outputs; the tail's own backward kernel does not run.  The objective is the trainer's
    shape — photometric mean + a gradient on rgb_rec (the sweep) + functions of disp and depth (the smoothness term's place) —
    and the gradients w.r.t. the conv outputs and the plane levels must agree (a) fused against unfused on the GPU, tightly,
    and (b) both against autograd through the oracle's decoder tail + warp_and_loss on the CPU.  Inputs: the conv outputs of
    tests/golden/decoder_tail.npz (captured next to the reference DepthDecoder); 49 planes with target "l" (negative shifts,
    every segment count); sigmas ON both clamp bounds (sigmoid saturated at 1, pushed below 0.01) plus a plane whose shift is an
    integer (the irregular path: atomics on rows that start from the tail's own term)."""
    import types
    from gpu_cases import make_stub_trainer
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    from planedepth_amd.decoder_tail import fused_decoder_tail
    from planedepth_amd.synthetic import intrinsics
    g = torch.Generator().manual_seed(321)
    side = "r"
    if case == "decoder_tail_npz":
        z = _tail_group("mix_xy")
        rl, rs = z["raw_logits"], z["raw_sigma"]
        lv_disp = z["disp_layered"][:, :, :1, :1].clone()          # the decoder's per-plane disparities
    else:
        B, N, H, W = (1, 49, 6, 640) if case == "many_planes_left_view" else (2, 7, 9, 256)
        rl = torch.randn(B, N, H, W, generator=g) * 2.5
        rs = torch.randn(B, N, H, W, generator=g) * 3 - 1
        lv_disp = 0.3 * W * (2.0 / (0.3 * W)) ** ((torch.arange(N, dtype=torch.float32)[None, :, None, None] +
                                                    torch.rand(B, N, 1, 1, generator=g) - 0.5) / (N - 1))
        if case == "many_planes_left_view":
            side = "l"
        else:
            rs[:, :, :, :40] = -9.0      # sigmoid = 1.2e-4: clamped to 0.01, gate closed
            rs[:, :, :, 40:80] = 30.0    # sigmoid = 1.0 exactly: on the upper bound, sigmoid' = 0
            lv_disp[:, 3] = 17.0         # an integer shift: the irregular path
    B, N, H, W = rl.shape
    col_l, col_t = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    gw = [torch.randn(B, 3, H, W, generator=g) * 1e-3, torch.randn(B, 1, H, W, generator=g) * 1e-2,
          torch.randn(B, 1, H, W, generator=g) * 1e-3]
    K, inv_K = intrinsics(B, H, W)

    def objective(ph, rgb, disp, depth, dev):
        return ph + (rgb * gw[0].to(dev)).sum() + (disp * gw[1].to(dev)).sum() + (depth * gw[2].to(dev)).sum()

    def run_gpu(fuse):
        dev = "cuda"
        a, s, d = (t.to(dev).clone().requires_grad_(True) for t in (rl, rs, lv_disp))
        outputs = {"disp_layered": d.expand(-1, -1, H, W), "padding_mask": None}
        fused_decoder_tail(outputs, a, s, use_mixture_loss=True, all_ones_mask=True, fuse_sweep_backward=fuse)
        link = getattr(outputs["logits"], "_pd_tail_link", None)
        inputs = {("color", "l"): col_l.to(dev), "K": K.to(dev), "inv_K": inv_K.to(dev)}
        if side != "l":
            inputs[("color", side)] = col_t.to(dev)
        opt = types.SimpleNamespace(warp_type="disp_warp", match_aug=False, use_mixture_loss=True, automask=False,
                                    render_probability=False, alpha_pc=0.0, alpha_self=0.0, self_distillation=0.0,
                                    gamma_smooth=2.0, alpha_smooth=0.0, use_ssim=False, xz_levels=0, yz_levels=0)
        trainer = make_stub_trainer(opt, [side])
        ops.KERNEL_EVENTS = {"fwd": [], "bwd": []}
        try:
            trainer.pred_novel_images(inputs, outputs)
            objective(outputs[("ph_mean", side)], outputs[("rgb_rec", side)], outputs["disp"], outputs["depth"], dev).backward()
            tail_launches = len(ops.KERNEL_EVENTS.get("tail_bwd", []))
        finally:
            ops.KERNEL_EVENTS = None
        assert (link is not None) == fuse
        if fuse:
            assert link.fused_passes == 1 and link.applied is None and not link.seen   # applied by the sweep, consumed by the tail's node
            assert tail_launches == 0, "the tail's own backward kernel ran although the sweep applied it"
        else:
            assert tail_launches == 1
        return {k: v.grad.detach().cpu() for k, v in (("g_raw_logits", a), ("g_raw_sigma", s), ("g_disp", d))}

    def run_cpu():
        a, s, d = (t.clone().requires_grad_(True) for t in (rl, rs, lv_disp))
        dl = d.expand(-1, -1, H, W)
        o = orc.decoder_tail(a, s, torch.ones_like(a), dl, W, use_mixture_loss=True)
        Rt = torch.eye(4)[None].repeat(B, 1, 1)
        r = orc.warp_and_loss(col_l, col_l if side == "l" else col_t, o["logits"], o["sigma"], warp_type="disp_warp",
                              target_side=side, disp_layered=dl, padding_mask=torch.ones_like(a),
                              distance=0.1 * 0.58 * W / d[:, :, 0, 0], norm=torch.tensor([0.0, 0.0, 1.0])[None, None].expand(B, N, -1),
                              T=Rt, K=K, inv_K=inv_K, use_mixture_loss=True, automask=False)
        objective(r["ph_loss"], r["rgb_rec"], o["disp"], o["depth"], "cpu").backward()
        return {"g_raw_logits": a.grad, "g_raw_sigma": s.grad, "g_disp": d.grad}

    plain, fused, want = run_gpu(False), run_gpu(True), run_cpu()
    for k in want:
        assert rel_err(fused[k], plain[k]) < 5e-6, (k, "fused vs unfused", rel_err(fused[k], plain[k]))
        assert rel_err(plain[k], want[k]) < TOL, (k, "unfused vs oracle", rel_err(plain[k], want[k]))
        assert rel_err(fused[k], want[k]) < TOL, (k, "fused vs oracle", rel_err(fused[k], want[k]))


def test_fused_decoder_tail_stays_correct_when_disp_is_consumed_before_the_taps():
    """The fused form relies on gradient taps that pred_novel_images puts on outputs["disp"] / ["depth"] AFTER the sweep's node
    exists.  A consumer that took ``disp`` before that (here: the objective holds the untapped tensor) delivers its gradient to
    the tail's node only, after the sweep's backward has run: the tail's backward must then add that remainder itself."""
    import types
    from gpu_cases import make_stub_trainer
    from planedepth_amd.decoder_tail import fused_decoder_tail
    from planedepth_amd.synthetic import intrinsics
    g = torch.Generator().manual_seed(11)
    B, N, H, W = 1, 6, 8, 128
    rl, rs = torch.randn(B, N, H, W, generator=g) * 2, torch.randn(B, N, H, W, generator=g) * 2
    lv = 40.0 * (2.0 / 40.0) ** (torch.arange(N, dtype=torch.float32)[None, :, None, None] / (N - 1))
    cl, ct, w = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    K, inv_K = intrinsics(B, H, W)
    res = []
    for fuse in (False, True):
        a, s, d = (t.cuda().clone().requires_grad_(True) for t in (rl, rs, lv))
        outputs = {"disp_layered": d.expand(-1, -1, H, W), "padding_mask": None}
        fused_decoder_tail(outputs, a, s, use_mixture_loss=True, all_ones_mask=True, fuse_sweep_backward=fuse)
        early_disp = outputs["disp"]                       # taken BEFORE pred_novel_images installs the taps
        opt = types.SimpleNamespace(warp_type="disp_warp", match_aug=False, use_mixture_loss=True, automask=False,
                                    render_probability=False, alpha_pc=0.0, alpha_self=0.0, self_distillation=0.0,
                                    gamma_smooth=2.0, alpha_smooth=0.0, use_ssim=False, xz_levels=0, yz_levels=0)
        trainer = make_stub_trainer(opt, ["r"])
        trainer.pred_novel_images({("color", "l"): cl.cuda(), ("color", "r"): ct.cuda(), "K": K.cuda(), "inv_K": inv_K.cuda()}, outputs)
        (outputs[("ph_mean", "r")] + (early_disp * w.cuda()).sum() + (outputs["disp"] * 0.5 * w.cuda()).sum()).backward()
        res.append([t.grad.cpu() for t in (a, s, d)])
    for x, y in zip(*res):
        assert rel_err(y, x) < 5e-6, rel_err(y, x)


def test_fused_decoder_tail_survives_a_second_backward_over_the_same_graph():
    """what the sweep's backward leaves with the TailLink (the taps' gradients, "the tail's terms are applied") is
    state of ONE backward pass.  With retain_graph=True the graph is walked again — both passes must give the unfused graph's
    gradients (accumulated: twice the single pass), and a torch.autograd.grad over the retained graph likewise."""
    import types
    from gpu_cases import make_stub_trainer
    from planedepth_amd.decoder_tail import fused_decoder_tail
    from planedepth_amd.synthetic import intrinsics
    g = torch.Generator().manual_seed(12)
    B, N, H, W = 2, 5, 8, 128
    rl, rs = torch.randn(B, N, H, W, generator=g) * 2, torch.randn(B, N, H, W, generator=g) * 2
    lv = 40.0 * (2.0 / 40.0) ** (torch.arange(N, dtype=torch.float32)[None, :, None, None] / (N - 1)).repeat(B, 1, 1, 1)
    cl, ct, w = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    K, inv_K = intrinsics(B, H, W)
    res = {}
    for fuse in (False, True):
        a, s, d = (t.cuda().clone().requires_grad_(True) for t in (rl, rs, lv))
        outputs = {"disp_layered": d.expand(-1, -1, H, W), "padding_mask": None}
        fused_decoder_tail(outputs, a, s, use_mixture_loss=True, all_ones_mask=True, fuse_sweep_backward=fuse)
        opt = types.SimpleNamespace(warp_type="disp_warp", match_aug=False, use_mixture_loss=True, automask=False,
                                    render_probability=False, alpha_pc=0.0, alpha_self=0.0, self_distillation=0.0,
                                    gamma_smooth=2.0, alpha_smooth=0.0, use_ssim=False, xz_levels=0, yz_levels=0)
        trainer = make_stub_trainer(opt, ["r"])
        trainer.pred_novel_images({("color", "l"): cl.cuda(), ("color", "r"): ct.cuda(), "K": K.cuda(), "inv_K": inv_K.cuda()}, outputs)
        obj = outputs[("ph_mean", "r")] + (outputs["disp"] * w.cuda()).sum() + (outputs["depth"] * 1e-2 * w.cuda()).sum()
        obj.backward(retain_graph=True)
        first = [t.grad.clone() for t in (a, s, d)]
        again = torch.autograd.grad(obj, (a, s, d), retain_graph=True)
        obj.backward()
        res[fuse] = dict(first=[t.cpu() for t in first], again=[t.cpu() for t in again], accumulated=[t.grad.cpu() for t in (a, s, d)])
        link = getattr(outputs["logits"], "_pd_tail_link", None)
        if fuse:
            assert link.fused_passes == 3 and link.applied is None and not link.seen
    for k in ("first", "again", "accumulated"):
        for x, y in zip(res[False][k], res[True][k]):
            assert rel_err(y, x) < 5e-6, (k, rel_err(y, x))
    for x, y in zip(res[True]["first"], res[True]["again"]):
        assert rel_err(y, x) < 2e-6, rel_err(y, x)       # (the per-plane disparity gradient is a float-atomic sum)
    for x, y in zip(res[True]["first"], res[True]["accumulated"]):
        assert rel_err(y, 2 * x) < 2e-6, rel_err(y, 2 * x)


def test_smooth_loss_against_reference_vector_and_oracle():
    """get_smooth_loss_disp as a HIP kernel — the reference's own value (modules.npz), then the
    0.2W crop of trainer.py:768 read in place through its strides, forward and backward, against the oracle."""
    import numpy as np
    import planedepth_amd as pa
    from oracle import planedepth_oracle as orc
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "modules.npz")).items()}
    got = pa.get_smooth_loss_disp(z["sm_disp"].cuda(), z["sm_img"].cuda(), gamma=2)
    assert rel_err(got.cpu(), z["sm_loss"]) < TOL
    g = torch.Generator().manual_seed(12)
    B, H, W = 3, 37, 90
    disp = torch.rand(B, 1, H, W, generator=g) * 20
    img = torch.rand(B, 3, H, W, generator=g)
    x0 = int(0.2 * W)
    d64 = disp.double().requires_grad_(True)
    want = orc.smooth_loss_disp(d64[..., x0:], img.double()[..., x0:], 2.0)
    (want * 3.0).backward()
    dg = disp.cuda().requires_grad_(True)
    got = pa.get_smooth_loss_disp(dg[..., x0:], img.cuda()[..., x0:], gamma=2.0)
    (got * 3.0).backward()
    assert rel_err(got.detach().cpu(), want.detach().float()) < TOL
    assert rel_err(dg.grad.cpu(), d64.grad.float()) < TOL
    # the form compute_losses uses: the crop as an operator argument (pointer offset forward, the uncropped gradient with
    # its zeros written by the backward kernel — no slice node in the graph): the same gradient bit for bit, the same
    # loss up to the order in which the forward's block sums reach its one atomic word
    from planedepth_amd import ops
    dx = disp.cuda().requires_grad_(True)
    gotx = ops.smooth_loss_disp(dx, img.cuda(), 2.0, x0=x0)
    (gotx * 3.0).backward()
    assert abs(float(gotx) - float(got)) <= 4e-7 * abs(float(got)) and torch.equal(dx.grad, dg.grad)
    assert float(dx.grad[..., :x0].abs().max()) == 0.0


@pytest.mark.parametrize("tag,promise", [("xy", False), ("rows", False), ("rows", True)])
def test_post_process_against_reference_vectors(tag, promise):
    """the fused post-process warps against Trainer.generate_post_process_disp's own output
    (tests/golden/post_process.npz), through the trainer-level entry point with a stub for the fixed networks."""
    import types
    import numpy as np
    import planedepth_amd as pa
    z = np.load(os.path.join(GOLDEN, "post_process.npz"))
    z = {k.split("/", 1)[1]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith(tag + "/")}
    B2, N, H, W = z["logits"].shape
    dl = z["disp_layered"]
    if tag == "xy":  # hand it over the way the decoder does for xy planes: an expanded view of [2B,N,1,1]
        dl = dl[:, :, :1, :1].contiguous().expand(-1, -1, H, W)
    fixed = dict(logits=z["logits"], probability=z["probability"], disp=z["disp"], disp_layered=dl)
    # promise: opt.yz_levels == 0 — the dense map of the xz-plane fixture is then read as one disparity per (plane, row) and served
    # by the row kernels / row chains (PD_PP_DISP_ROWS) instead of the per-pixel gather form
    ns = types.SimpleNamespace(opt=types.SimpleNamespace(num_ep=1, net_type="ResNet", **(dict(yz_levels=0) if promise else {})),
                               fixed_models={"encoder": lambda x: None, "depth": lambda f, g: fixed})
    inputs = {("color_aug", "l"): torch.zeros(B2 // 2, 3, H, W, device="cuda"),
              "grid": torch.zeros(B2 // 2, 2, H, W, device="cuda")}
    disp_pp, mask_novel = pa.generate_post_process_disp(ns, inputs)
    assert rel_err(disp_pp.cpu(), z["disp_pp"].cpu()) < TOL
    assert rel_err(mask_novel.cpu(), z["mask_novel"].cpu()) < TOL


def test_post_process_fullsize_vs_oracle():
    """192x640, 49 planes: the fused warps against the oracle's restatement (fp32, same op order for coordinates)."""
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    B, N, H, W = 1, 49, 192, 640
    g = torch.Generator().manual_seed(31)
    logits = torch.randn(2 * B, N, H, W, generator=g) * 2
    sigma = torch.rand(2 * B, N, H, W, generator=g) * 0.9 + 0.05
    w = torch.softmax(logits, 1) / sigma
    prob = w / w.sum(1, True)
    lv = torch.arange(N, dtype=torch.float32)[None, :, None, None] + torch.rand(2 * B, N, 1, 1, generator=g) - 0.5
    dl = (300.0 * (2.0 / 300.0) ** (lv / (N - 1))).expand(-1, -1, H, W)
    disp = (prob * dl).sum(1, True)
    want = orc.post_process_disp(logits, prob, disp, dl)
    got = ops.post_process_disp(logits.cuda(), prob.cuda(), disp.cuda(), dl.cuda())
    assert rel_err(got[0].cpu(), want[0]) < TOL
    assert rel_err(got[1].cpu(), want[1]) < TOL


@pytest.mark.parametrize("H,W,N,dmax", [(2, 2, 1, 0.9), (3, 5, 2, 9.0), (7, 63, 5, 40.0), (9, 65, 4, 70.0), (5, 130, 9, 300.0)])
def test_post_process_row_kernels_on_ragged_shapes(H, W, N, dmax):
    """The row-shift form of the post-process warps (per-plane scalar disparities) on widths below / just above one
    64-lane segment, one plane, disparities beyond the row (every tap out of view), integer disparities (taps exactly on
    columns, the x0 = -1 / mirrored-edge fix-ups) — against the oracle; the mirrored (flip) read is half of every call."""
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    B = 2
    g = torch.Generator().manual_seed(1000 + W)
    logits = torch.randn(2 * B, N, H, W, generator=g) * 2
    sigma = torch.rand(2 * B, N, H, W, generator=g) * 0.9 + 0.05
    w = torch.softmax(logits, 1) / sigma
    prob = w / w.sum(1, True)
    lv = torch.rand(2 * B, N, 1, 1, generator=g) * dmax
    lv[:, 0] = torch.round(lv[:, 0])                       # an integer disparity per image
    dl = lv.expand(-1, -1, H, W)
    disp = (prob * dl).sum(1, True)
    want = orc.post_process_disp(logits, prob, disp, dl)
    got = ops.post_process_disp(logits.cuda(), prob.cuda(), disp.cuda(), dl.cuda())
    for a_, b_ in zip(got, want):
        if float(b_.abs().max()) == 0.0:
            assert float(a_.abs().max()) < 1e-6
        else:
            assert rel_err(a_.cpu(), b_) < TOL


@pytest.mark.parametrize("H,W,N,dmax", [(4, 2, 3, 1.5), (6, 128, 20, 60.0), (5, 130, 33, 140.0), (9, 256, 63, 300.0), (3, 258, 64, 400.0),
                                        (4, 384, 70, 300.0)])
def test_post_process_segment_kernels_vs_oracle_and_row_kernels(H, W, N, dmax):
    """Round 6: the segment form of the post-process warps (two pixels per lane, 12-byte taps, all planes' samples in registers:
    N <= 32 and N <= 64 instantiations; N = 70 keeps the row kernels' softmax with the segment sum) on even widths below, at and
    beyond a 128-pixel segment, with integer disparities (the irregular planes' per-pixel path), disparities beyond the row, the
    mirrored read and both signs — against the oracle at 1e-4, and the two warps alone against the row kernels (a child process
    under PD_PP_SEG=0: the switches are read once per process) within fp32 reassociation."""
    import subprocess
    import sys
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    B = 2
    g = torch.Generator().manual_seed(2000 + W + N)
    logits = torch.randn(2 * B, N, H, W, generator=g) * 2
    sigma = torch.rand(2 * B, N, H, W, generator=g) * 0.9 + 0.05
    w = torch.softmax(logits, 1) / sigma
    prob = w / w.sum(1, True)
    lv = torch.rand(2 * B, N, 1, 1, generator=g) * dmax
    lv[:, 0] = torch.round(lv[:, 0])
    if N > 2:
        lv[:, 1] = float(W + 5)                            # out of view
    dl = lv.expand(-1, -1, H, W)
    disp = (prob * dl).sum(1, True)
    want = orc.post_process_disp(logits, prob, disp, dl)
    got = ops.post_process_disp(logits.cuda(), prob.cuda(), disp.cuda(), dl.cuda())
    for a_, b_ in zip(got, want):
        assert rel_err(a_.cpu(), b_) < TOL
    # pd_post_process against the six operator calls: equal bit for bit where it IS those calls (N > 64), within fp32
    # reassociation where the row chains take the vertical weights out of the plane sum (N <= 64, W <= 1024)
    for a_, b_ in zip(got, ops.post_process_disp_stepwise(logits.cuda(), prob.cuda(), disp.cuda(), dl.cuda())):
        assert torch.equal(a_, b_) if N > 64 else float((a_ - b_).abs().max()) <= 3e-6 * max(float(b_.abs().max()), 1.0), float((a_ - b_).abs().max())
    here = {}
    for sign, flip in ((1.0, False), (-1.0, False), (-1.0, True), (1.0, True)):
        here[(sign, flip)] = (ops.warp_softmax(logits[:B].cuda(), dl[:B].cuda(), sign, flip_src=flip).cpu(),
                              ops.warp_sum(prob[:B].cuda(), dl[:B].cuda(), sign, flip_src=flip).cpu())
    path = "/tmp/pd_pp_seg_%d_%d.pt" % (W, N)
    torch.save({"logits": logits[:B], "prob": prob[:B], "dl": lv[:B]}, path)
    code = ("import torch, sys; sys.path.insert(0, %r); from planedepth_amd import ops; z = torch.load(%r); "
            "dl = z['dl'].expand(-1, -1, %d, %d).cuda(); out = {}\n"
            "for sign, flip in ((1.0, False), (-1.0, False), (-1.0, True), (1.0, True)):\n"
            "    out[(sign, flip)] = (ops.warp_softmax(z['logits'].cuda(), dl, sign, flip_src=flip).cpu(), "
            "ops.warp_sum(z['prob'].cuda(), dl, sign, flip_src=flip).cpu())\n"
            "torch.save(out, %r)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, H, W, path + ".out"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PD_PP_SEG="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = torch.load(path + ".out")
    for key, (sm, su) in here.items():
        assert float((sm - rows[key][0]).abs().max()) < 2e-6, key
        assert float((su - rows[key][1]).abs().max()) < 2e-6, key


@pytest.mark.parametrize("H,W,N,n_xz", [(12, 128, 20, 6), (9, 256, 49, 14), (7, 130, 63, 14), (6, 770, 24, 5), (5, 384, 70, 9)])
def test_post_process_with_per_row_disparities(H, W, N, n_xz):
    """PD_PP_DISP_ROWS: the last ``n_xz`` planes get a disparity that changes with the row (the decoder's ground planes: a dense
    [2B,N,H,W] map that is constant along x).  Under the ``row_uniform`` promise the row kernels and, through pd_post_process, the
    row chains serve it — a chain takes the plane sum of source row r once per TARGET row that blends r in (the shifts of the second
    warp are the target row's): against the oracle at 1e-4, against the dense (per-pixel gather) form and against the single
    warps within fp32 reassociation."""
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    B = 2
    g = torch.Generator().manual_seed(3000 + W + N)
    logits = torch.randn(2 * B, N, H, W, generator=g) * 2
    sigma = torch.rand(2 * B, N, H, W, generator=g) * 0.9 + 0.05
    w = torch.softmax(logits, 1) / sigma
    prob = w / w.sum(1, True)
    base = torch.rand(2 * B, N, 1, 1, generator=g) * min(120.0, W * 0.6)
    rows = base.expand(-1, -1, H, 1).clone()
    rows[:, N - n_xz:] = rows[:, N - n_xz:] * torch.linspace(0.1, 2.5, H).view(1, 1, H, 1) * (0.5 + torch.rand(2 * B, n_xz, 1, 1, generator=g))
    dl = rows.expand(-1, -1, -1, W).contiguous()                      # what the decoder's cat() hands over
    disp = (prob * dl).sum(1, True)
    want = orc.post_process_disp(logits, prob, disp, dl)
    args = (logits.cuda(), prob.cuda(), disp.cuda(), dl.cuda())
    got = ops.post_process_disp(*args, row_uniform=True)
    dense = ops.post_process_disp(*args)                               # no promise: PD_PP_DISP_DENSE
    step = ops.post_process_disp_stepwise(*args, row_uniform=True)
    for a_, b_, c_, d_ in zip(got, want, dense, step):
        assert rel_err(a_.cpu(), b_) < TOL
        assert float((a_ - c_).abs().max()) <= 3e-6 * max(float(c_.abs().max()), 1.0)
        assert float((a_ - d_).abs().max()) <= 3e-6 * max(float(d_.abs().max()), 1.0)
    got_view = ops.post_process_disp(args[0], args[1], args[2], rows.cuda().expand(-1, -1, -1, W))   # an x-expanded view: no promise needed
    for a_, b_ in zip(got, got_view):
        assert torch.equal(a_, b_)


def test_plane_disparities_vs_the_decoders_expression():
    """networks/depth_decoder.py:147-152 (`disp_max * (disp_min / disp_max) ** (levels / (no_levels - 1))`,
    `0.1 * 0.58 * W / disp_layered`) in one launch each way: values and the gradient into the levels against the reference's own
    torch expression on the CPU."""
    from planedepth_amd import ops
    B, N, W = 3, 49, 640
    g = torch.Generator().manual_seed(77)
    res = torch.rand(B, N, 1, 1, generator=g) - 0.5
    lv = (torch.arange(N, dtype=torch.float32)[None, :, None, None] + res).requires_grad_(True)
    disp_min, disp_max = 2.0, 300.0
    want = disp_max * (disp_min / disp_max) ** (lv / (N - 1))
    want_dist = 0.1 * 0.58 * W / want[:, :, 0, 0]
    gd, gdist = torch.randn(B, N, 1, 1, generator=g), torch.randn(B, N, generator=g)
    torch.autograd.backward([want, want_dist], [gd, gdist])
    lv2 = lv.detach().cuda().requires_grad_(True)
    got, got_dist = ops.plane_disparities(lv2, disp_min, disp_max, W)
    assert got.shape == (B, N, 1, 1) and got_dist.shape == (B, N)
    torch.autograd.backward([got, got_dist], [gd.cuda(), gdist.cuda()])
    assert rel_err(got.detach().cpu(), want.detach()) < 2e-6
    assert rel_err(got_dist.detach().cpu(), want_dist.detach()) < 2e-6
    assert rel_err(lv2.grad.cpu(), lv.grad) < 1e-5
    got2, _ = ops.plane_disparities(lv.detach().cuda().requires_grad_(True), disp_min, disp_max, W)   # only disp consumed
    (got2 * gd.cuda()).sum().backward()


def test_add_flip_right_inputs_is_bit_exact():
    """the batch-doubling kernel against the oracle's cat/flip restatement of trainer.py:252-276."""
    import types
    import planedepth_amd as pa
    from oracle import planedepth_oracle as orc
    g = torch.Generator().manual_seed(3)
    B, H, W = 3, 13, 37
    inputs = {(k, s): torch.rand(B, 3, H, W, generator=g) for k in ("color", "color_aug") for s in ("l", "r", -1, 1)}
    inputs.update({("depth_gt", s): torch.rand(B, 1, H, W, generator=g) for s in ("l", "r")})
    inputs["grid"] = torch.randn(B, 2, H, W, generator=g)
    inputs.update({k: torch.randn(B, 4, 4, generator=g) for k in ("K", "inv_K", ("Rt", "l"), ("Rt", "r"))})
    want = orc.add_flip_right_inputs(inputs, novel_frame_ids=(-1, 1))
    ns = types.SimpleNamespace(opt=types.SimpleNamespace(novel_frame_ids=[-1, 1]))
    got = pa.add_flip_right_inputs(ns, {k: v.cuda() for k, v in inputs.items()})
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k].cpu(), want[k]), k


def test_add_flip_right_inputs_takes_the_dataloaders_cpu_batch():
    """The reference calls add_flip_right_inputs on the DataLoader's CPU batch, before process_batch moves it to the
    device (trainer.py:294-295 vs 328-329): CPU tensors in, through a patch_trainer'ed stub whose ``device`` is the GPU —
    the doubled batch comes back on the device, bit-exact against the oracle's cat/flip restatement."""
    import types
    from gpu_cases import make_stub_trainer
    from oracle import planedepth_oracle as orc
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 11, 29
    inputs = {(k, s): torch.rand(B, 3, H, W, generator=g) for k in ("color", "color_aug") for s in ("l", "r", -1, 1)}
    inputs["grid"] = torch.randn(B, 2, H, W, generator=g)
    inputs.update({k: torch.randn(B, 4, 4, generator=g) for k in ("K", "inv_K", ("Rt", "l"), ("Rt", "r"))})
    assert all(not v.is_cuda for v in inputs.values())
    want = orc.add_flip_right_inputs(inputs, novel_frame_ids=(-1, 1))
    trainer = make_stub_trainer(types.SimpleNamespace(novel_frame_ids=[-1, 1]), ["r", -1, 1], "cuda")
    got = trainer.add_flip_right_inputs(inputs)
    assert set(got) == set(want)
    for k in want:
        assert got[k].is_cuda, k
        assert torch.equal(got[k].cpu(), want[k]), k
    assert all(not v.is_cuda for v in inputs.values())     # the caller's batch is left where it was


@pytest.mark.parametrize("gpus,ddp_model", [(2, "r18"), (4, "r50"), (8, "r18")])
def test_bench_spawns_its_own_ranks(gpus, ddp_model):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (bench.spawn_ranks);
    on a one-GPU box PD_BENCH_SHARE_GPU=1 puts all ranks on cuda:0 with gloo for the collectives.  The JSON line must report
    N ranks, a comm block, every rank's shard in `value` and finite numbers — through the HIP path of all ranks — and the
    DDP training-step block (three more DDP re-wraps for the bucket sizes, under its timer) must FINISH: at world 4 with the
    ResNet-50 + dense-ASPP-shaped stand-in (BASELINE configs[2]'s 157 MB of gradients) as well as at world 2 with the
    ResNet-18-shaped one, and at world 8 (eight ranks on the one GPU: what the driver's SCALE run launches on an 8-GPU node,
    plumbing-wise).  A block that hangs or fails says so at the TOP level of the line (`ddp_step_timed_out` /
    `ddp_step_failed`), which this test requires to be false."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PD_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", PD_DDP_STEP_TIMEOUT_S="600")   # (gloo moves 157 MB per step here)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
                          "--no_cpu_baseline", "--no_next_rows", "--batch", "2", "--height", "64", "--width", "128", "--planes", "9",
                          "--ddp_model", ddp_model],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == gpus and res["comm"]["world_size"] == gpus, res
    assert res["value"] > 0 and res["ms_per_step"] > 0 and res["value"] == res["value"]
    assert res["config"]["global_batch"] == gpus * 2      # weak scaling: every rank brings its own shard ...
    assert abs(res["value"] - gpus * 2 * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) <= 1e-3 * res["value"]   # ... and is counted
    assert res["ddp_step_timed_out"] is False and res["ddp_step_failed"] is False, res.get("ddp_step")
    # the DDP training-step block ran on all ranks (gloo carries DDP's all-reduce here): the figures SCALE runs will read
    # at 2/4/8 GPUs exist and are finite before an 8-GPU node ever sees this code
    blk = res["ddp_step"]
    assert "error" not in blk and "skipped" not in blk, blk
    assert blk["world_size"] == gpus and blk["ms_per_step"] > 0 and blk["ms_per_step_without_gradient_sync"] > 0
    assert ("ResNet-50" in blk["network"]) == (ddp_model == "r50")
    import math
    for key in ("allreduce_alone_ms", "allreduce_exposed_ms"):
        assert math.isfinite(blk[key]) and blk[key] >= 0.0, (key, blk[key])
    assert blk["allreduce_hidden_share"] is not None and 0.0 <= blk["allreduce_hidden_share"] <= 1.0
    assert sorted(blk["by_bucket_cap_mb"]) == ["10", "100", "50"]
    for cap, v in blk["by_bucket_cap_mb"].items():
        assert math.isfinite(v["ms_per_step"]) and v["ms_per_step"] > 0 and math.isfinite(v["allreduce_exposed_ms"]), (cap, v)
    assert math.isfinite(blk["sweep_fwd_ms"]) and math.isfinite(blk["sweep_bwd_ms"]) and math.isfinite(blk["tail_bwd_ms"])


def _shard_worker(rank, world, port, ret):
    """One rank of test_two_rank_hip_shards_reproduce_the_full_batch: the HIP sweep on this rank's shard (both ranks share
    cuda:0; gloo carries the barrier)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from gpu_cases import run_product
    from planedepth_amd import parallel
    from planedepth_amd.synthetic import build_case
    parallel.init_process_group_from_env("gloo")
    case = build_case(**SHARD_CASE)
    shard = parallel.shard_batch(case, rank, world, SHARD_CASE["B"])
    out = run_product(shard, {}, device="cuda:0")   # objective: mean(ph_map) over THIS shard + sum(rgb_rec * g)
    parallel.barrier()
    ret[rank] = {k: out[k] for k in ("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp")}
    import torch.distributed as dist
    dist.destroy_process_group()


SHARD_CASE = dict(B=4, N=7, H=20, W=96, seed=91, disp_min=0.5, disp_max=30.0, sigma_interior=True)


def test_two_rank_hip_shards_reproduce_the_full_batch():
    """SURVEY 8e on the product path: two processes each run the HIP sweep on half of the batch (no data-path collective)
    and together reproduce the full-batch run — values exactly per image, gradients up to the loss mean's 1/B vs
    1/(B/2) (DDP's gradient averaging supplies that factor in training)."""
    import torch.multiprocessing as mp
    from gpu_cases import run_product
    from planedepth_amd.synthetic import build_case
    world = 2
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, 29641, ret)) for r in range(world)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(300)
        assert p_.exitcode == 0
    # The shard's photometric head is a mean over B/world images, the full batch's over B: run the full batch with that head
    # scaled by `world` (what DDP's gradient averaging does to the factor in training) and the gradients agree per image.
    full = run_product(build_case(**SHARD_CASE), dict(ph_scale=float(world)))
    per = SHARD_CASE["B"] // world
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        assert torch.equal(ret[r]["rgb_rec"], full["rgb_rec"][sl]) and torch.equal(ret[r]["ph_map"], full["ph_map"][sl])
        for k in ("g_logits", "g_sigma", "g_disp_pp"):
            assert float(full[k][sl].abs().max()) > 0, k
            # (the per-plane disparity gradient is a sum of float atomics over the image's rows: its last bits depend on the order)
            assert rel_err(ret[r][k], full[k][sl]) < (1e-5 if k == "g_disp_pp" else 1e-6), (k, rel_err(ret[r][k], full[k][sl]))
    assert abs(0.5 * (float(ret[0]["ph_loss"]) + float(ret[1]["ph_loss"])) - float(full["ph_loss"])) < 1e-6


@pytest.mark.parametrize("label,case_kw,run,opt_extra", [
    # BASELINE configs[2]/[3] plane count: 49 xy + 14 xz planes, horizon mask, automask (per-row disparities + row masks)
    ("n63_xz", dict(B=1, N=63, H=192, W=640, n_xz=14), dict(automask=True), dict(yz_levels=0, xz_levels=14)),
    # BASELINE config (5): high resolution
    ("hr", dict(B=1, N=49, H=384, W=1280), dict(), dict(yz_levels=0, xz_levels=0)),
    # BASELINE configs[2]: batch 12 per GPU — every tensor of all twelve images (the loss mean runs over the whole batch)
    ("batch12", dict(B=12, N=49, H=192, W=640), dict(), dict(yz_levels=0, xz_levels=0)),
])
def test_other_baseline_configs_fullsize_vs_oracle(label, case_kw, run, opt_extra, row_mode):
    """The other full-size configurations of BASELINE.json / SURVEY §8d against the fp32 oracle (B=1)."""
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import survey_fullsize_case
    case = survey_fullsize_case(sigma_interior=True, **case_kw)
    got = run_product(case, run, opt_extra=opt_extra)
    if label == "n63_xz":
        assert ops.LAST_SWEEP_FLAGS & C.PD_DISP_ROWS and ops.LAST_SWEEP_FLAGS & C.PD_MASK_ROWS
    want = run_oracle(case, run)
    _compare(got, want, keys=("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp"), tag=label, tol=_row_tol(row_mode, label))


@pytest.mark.parametrize("impl", ["rows", "general"])
def test_fused_mean_of_ph_map(impl):
    """The `.mean()` of trainer.py:742 accumulated inside the sweep kernel (ph_mean) equals ph_map.mean(), and a loss
    built on it back-propagates the same gradients as one built on ph_map.mean()."""
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(B=2, N=7, H=18, W=150, seed=77, disp_min=0.5, disp_max=30.0, sigma_interior=True)
    c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.items()}
    B, N, H, W = c["logits"].shape
    ops.SWEEP_IMPL = C.PD_IMPL_GENERAL if impl == "general" else C.PD_IMPL_AUTO
    try:
        grads = []
        for fused in (True, False):
            lg, sg, dp = (c[k].clone().requires_grad_(True) for k in ("logits", "sigma", "disp_pp"))
            rgb, ph_map, ph_mean = ops.plane_sweep_disp(c["color_l"], c["color_r"], lg, sg, dp.expand(B, N, H, W), None,
                                                        automask=True, return_mean=True)
            assert abs(float(ph_mean) - float(ph_map.mean())) < 2e-6 * abs(float(ph_map.mean()))
            loss = (ph_mean if fused else ph_map.mean()) * 1.7 + (rgb * c["g_rgb_rec"]).sum()
            loss.backward()
            grads.append((lg.grad, sg.grad, dp.grad))
        for a, b in zip(*grads):
            assert rel_err(a.cpu(), b.cpu()) < 2e-6
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO


def test_pred_self_images_vs_oracle():
    """trainer.py:605-633 through the HIP modules (backproject, project, border-mode grid_sample), forward and the
    gradient w.r.t. the disparity, against the oracle's restatement of the same chain."""
    import types
    import planedepth_amd as pa
    from oracle import planedepth_oracle as orc
    from planedepth_amd.synthetic import intrinsics, small_pose
    g = torch.Generator().manual_seed(21)
    B, N, H, W = 2, 4, 20, 48
    K, inv_K = intrinsics(B, H, W)
    T = small_pose(g, B, stereo=True)
    disp = (torch.rand(B, 1, H, W, generator=g) * 20 + 2)
    color = torch.rand(B, 3, H, W, generator=g)
    gw = torch.randn(B, 3, H, W, generator=g)

    from cases import three_way

    def chain(dt):
        dd = disp.detach().clone().to(dt).requires_grad_(True)
        cam = orc.backproject_depth(0.1 * 0.58 * W / dd, inv_K.to(dt))
        out = orc.bilinear_sample(color.to(dt), orc.project_3d(cam, K.to(dt), T.to(dt), H, W), "border")
        (out * gw.to(dt)).sum().backward()
        return dd, out

    (d64, want), (d32, _) = chain(torch.float64), chain(torch.float32)

    dg = disp.cuda().requires_grad_(True)
    ns = types.SimpleNamespace(opt=types.SimpleNamespace(match_aug=False))
    outputs = {"disp": dg, "probability": torch.empty(B, N, H, W, device="meta")}
    inputs = {("Rt", "r"): T.cuda(), "K": K.cuda(), "inv_K": inv_K.cuda(), ("color", "r"): color.cuda()}
    pa.pred_self_images(ns, inputs, outputs)
    (outputs["self_rec"] * gw.cuda()).sum().backward()
    assert rel_err(outputs["self_rec"].detach().cpu(), want.detach().float()) < TOL
    # bilinear derivative through fp32 coordinates: as close to fp64 as the same chain in torch fp32 is
    assert three_way(dg.grad.cpu(), d32.grad, d64.grad)[0], three_way(dg.grad.cpu(), d32.grad, d64.grad)


@pytest.mark.parametrize("seed,kw,run", [
    (301, dict(B=1, N=1, H=2, W=2, disp_min=0.2, disp_max=0.9), dict()),                       # smallest legal image
    (302, dict(B=2, N=1, H=3, W=5, disp_min=0.2, disp_max=3.0), dict(use_mixture_loss=False)),  # a single plane
    (303, dict(B=1, N=3, H=5, W=17, disp_min=0.5, disp_max=20.0), dict(automask=True)),          # shifts beyond the row
    (304, dict(B=2, N=4, H=9, W=33, disp_min=0.5, disp_max=12.0), dict(target_side="l")),
    (305, dict(B=1, N=5, H=7, W=63, disp_min=0.5, disp_max=30.0, n_xz=2), dict()),             # one partial segment
    (306, dict(B=1, N=2, H=4, W=3, disp_min=0.1, disp_max=1.5, stereo_T=False), dict(warp_type="homography_warp")),
])
def test_degenerate_shapes_vs_oracle(seed, kw, run, row_mode):
    """Images narrower than one 64-lane segment, a single plane, disparities larger than the row (every tap out of
    view), the 2x2 minimum: the edge cases the reference's code admits (H, W >= 2: its normalisation divides by
    size - 1)."""
    from gpu_cases import run_product
    from planedepth_amd.synthetic import build_case
    case = build_case(seed=seed, sigma_interior=True, **kw)
    got = run_product(case, run)
    if run.get("warp_type") == "homography_warp":
        _compare3(got, case, run, tag="seed%d" % seed)
    else:
        _compare(got, run_oracle(case, run), tag="seed%d" % seed)


@pytest.mark.parametrize("kw,run", [
    (dict(B=2, N=9, H=12, W=640, disp_min=0.5, disp_max=200.0), dict(automask=True)),        # 10 segments on 4 waves: ragged round
    (dict(B=1, N=6, H=9, W=130, disp_min=0.5, disp_max=40.0), dict(target_side="l")),        # partial last segment
    (dict(B=2, N=5, H=7, W=33, disp_min=0.5, disp_max=12.0), dict(use_mixture_loss=False)),  # narrower than a wave, L1
    (dict(B=1, N=12, H=21, W=200, disp_min=0.5, disp_max=60.0, n_xz=4), dict(automask=True)),  # xz planes: per-row shifts + masks
    (dict(B=1, N=2, H=5, W=70, disp_min=0.5, disp_max=9.0), dict()),                         # N = 2: one real alpha + the closing plane
])
def test_render_probability_on_the_row_kernels(kw, run):
    """--render_probability (alpha compositing over the planes, trainer.py:584-591) on the row-shift kernels — running
    transmittance front to back in the forward, transmittance + prefix sum in the backward, g_dists written per pixel —
    against the general kernels (same formulas, atomic scatter) and the oracle."""
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(seed=4100 + kw["W"], sigma_interior=True, render_probability=True, **kw)
    run = dict(run, render_probability=True)
    extra = dict(yz_levels=0, xz_levels=kw.get("n_xz", 0))
    fast = run_product(case, run, opt_extra=extra)
    assert ops.LAST_SWEEP_FLAGS & C.PD_RENDER_PROB
    d = C.SweepDesc(kw["B"], kw["N"], kw["H"], kw["W"], C.PD_WARP_DISP, ops.LAST_SWEEP_FLAGS, 1.0, 0)
    assert C.load().pd_sweep_uses_rowshift(ctypes.byref(d))            # the fast path really served it
    ops.SWEEP_IMPL = C.PD_IMPL_GENERAL
    try:
        slow = run_product(case, run, opt_extra=extra)
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    want = run_oracle(case, run)
    keys = ("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp", "g_dists")
    _compare(fast, {k: want[k] for k in keys if k in want}, tag="rows/render")
    _compare(slow, {k: want[k] for k in keys if k in want}, tag="general/render")
    for k in ("g_logits", "g_sigma", "g_dists"):
        if k in fast and float(slow[k].abs().max()) > 0:
            assert rel_err(fast[k], slow[k]) < 5e-5, (k, rel_err(fast[k], slow[k]))


@pytest.mark.parametrize("view", ["pose_net", "stereo"])
@pytest.mark.parametrize("mix", [True, False])
def test_render_probability_on_the_homography_shortcuts(view, mix):
    """--render_probability with homography_warp: the plane-uniform kernels (pose_net view: zero translation) and the
    per-row-shift form of the stereo view against the general per-plane-homography kernels, all gradients incl. g_dists."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics, small_pose
    B, N, H, W = 2, 7, 24, 80
    g = torch.Generator().manual_seed(515)
    dev = "cuda"
    src, tgt = torch.rand(B, 3, H, W, generator=g).to(dev), torch.rand(B, 3, H, W, generator=g).to(dev)
    # positive logits: alpha = 1 - exp(-relu(l) dist) has a kink at l = 0, and the two stereo paths evaluate the sampling
    # position with different (equally legitimate) fp32 chains — a sample within 1e-5 of zero would flip the relu's gate
    # in one of them (seen: one element in 27 000).  Negative logits are covered against the oracle, where the row
    # kernels' coordinates are bit-exact (test_render_probability_on_the_row_kernels).
    logits = (torch.randn(B, N, H, W, generator=g).abs() + 0.05).to(dev)
    sigma = (0.011 + 0.978 * torch.rand(B, N, H, W, generator=g)).to(dev)
    dists = (torch.rand(B, N - 1, H, W, generator=g) * 2.0).to(dev)
    gw = (torch.randn(B, 3, H, W, generator=g) * 0.1).to(dev)
    distance = (0.5 + 5 * torch.rand(B, N, generator=g)).to(dev)
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1)
    norm[:, N // 2:] = torch.nn.functional.normalize(torch.tensor([0.0, 1.0, 0.07]), dim=0)   # "xz planes"
    norm = norm.to(dev)
    K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
    Rt = (_f8_pose(B, 31, 0.03) if view == "pose_net" else small_pose(None, B, stereo=True)).to(dev)
    res = {}
    for fast in (True, False):
        lg, sg, ds, dd = (t.clone().requires_grad_(True) for t in (logits, sigma, dists, distance))
        T = Rt.clone().requires_grad_(view == "pose_net")
        rgb, ph, ph_mean = ops.plane_sweep_homography(src, tgt, lg, sg if mix else None, dd, norm, T, K, inv_K,
                                                      use_mixture_loss=mix, automask=mix, render_probability=True, dists=ds,
                                                      return_mean=True, plane_uniform=fast and view == "pose_net",
                                                      stereo_rows=fast and view == "stereo")
        (ph_mean * 2.0 + (rgb * gw).sum()).backward()
        res[fast] = dict(rgb=rgb.detach().cpu(), ph=ph.detach().cpu(), g_logits=lg.grad.cpu(), g_dists=ds.grad.cpu(),
                         g_sigma=sg.grad.cpu() if mix else torch.zeros(1))
        if view == "pose_net":
            res[fast]["g_Rt"] = T.grad.cpu()[:, :3]
        else:
            res[fast]["g_distance"] = dd.grad.cpu()
    f, s_ = res[True], res[False]
    assert float(s_["g_dists"].abs().max()) > 0 and float(s_["g_logits"].abs().max()) > 0
    tol = 3e-6 if view == "pose_net" else 2e-4    # same coordinates / the row kernels' own coordinate chain (NOTEBOOK 3.5.2)
    for k in f:
        # g_distance of the stereo view: a bilinear DERIVATIVE through two different fp32 coordinate chains (the accuracy of
        # either against fp64 is what test_stereo_homography_as_row_shifts bounds)
        bound = 1e-3 if k == "g_distance" else (2e-4 if k == "g_Rt" else tol)
        assert rel_err(f[k], s_[k]) < bound, (view, k, rel_err(f[k], s_[k]))


def test_randomised_shapes_rowshift_vs_general():
    """Seeded sweep over odd shapes (heights 2..40 incl. odd ones, widths that are not multiples of 64, fewer planes
    than a plane group, batch 1..3, both sides, L1 / mixture / automask): the specialised kernels (row pairs, plane-axis
    work split, row-major dispatch) against the general kernels on the same inputs."""
    import random
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    rnd = random.Random(2024)
    for trial in range(30):
        B, N = rnd.randint(1, 3), rnd.randint(1, 11)
        H, W = rnd.randint(2, 40), rnd.choice([64, 65, 70, 96, 127, 128, 130, 191, 200])
        run = dict(target_side=rnd.choice(["l", "r"]), use_mixture_loss=rnd.random() < 0.7, automask=rnd.random() < 0.5)
        render = N >= 2 and rnd.random() < 0.35     # alpha compositing over the planes (needs a real alpha: N >= 2)
        if render:
            run["render_probability"] = True
        case = build_case(B=B, N=N, H=H, W=W, seed=900 + trial, disp_min=0.5, disp_max=0.6 * W, sigma_interior=True,
                          render_probability=render)
        fast = run_product(case, run)
        ops.SWEEP_IMPL = C.PD_IMPL_GENERAL
        try:
            slow = run_product(case, run)
        finally:
            ops.SWEEP_IMPL = C.PD_IMPL_AUTO
        tag = "trial%d B%d N%d %dx%d %s" % (trial, B, N, H, W, run)
        _compare(fast, slow, keys=("rgb_rec", "ph_map", "ph_loss", "g_disp_pp"), tag=tag, tol=2e-5)
        # (zero_floor: with one plane g_logits is rounding noise around 0; with target "l" under automask the identity loss
        # wins everywhere and so is g_sigma)
        if N == 1:
            # One plane: the softmax is constant and rgb_rec = the plane's colour whatever sigma is, so g_logits and the
            # rgb_rec part of g_sigma are EXACT zeros; what the kernels return is cancellation noise (gr.c - gr.rgb_rec)
            # amplified by 1/sigma^2 (up to 1e4) — bounded absolutely, not compared relatively.  Only the NLL's g_sigma
            # is a real number there, and under automask with target == source ("l") the identity loss wins everywhere.
            for k in ("g_logits", "g_sigma"):
                assert float((fast[k] - slow[k]).abs().max()) < max(1e-5, 1e-4 * float(slow[k].abs().max())), (tag, k)
        else:
            _compare(fast, slow, keys=("g_logits", "g_sigma", "g_dists"), tag=tag, tol=5e-5)  # eps-weighted cross-row adjoint term


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] as the trainer runs it: every target side, decoder-made xz planes, patch_trainer
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,stereo_constant", [("homo3", False), ("homo3", True), ("homo_nostereo_l1", False),
                                                 ("disp_xz", False)])
def test_trainer_mono_fixture_through_patch_trainer(tag, stereo_constant, monkeypatch):
    """tests/golden/trainer_mono.npz — captured from the reference's DepthDecoder (xz_levels = 3: non-frontal normals,
    horizon mask, per-plane distances) -> Trainer.predict_poses (Rt with zero translation and Rt[3,3] = 0, F8) ->
    Trainer.pred_novel_images over target_sides ["r", -1, 1] -> Trainer.compute_losses — through a stub class that
    adopted the product methods with patch_trainer."""
    from cases import cap_for, load_trainer_fixture, run_oracle_trainer
    from gpu_cases import run_product_trainer
    from planedepth_amd import sweep
    z, meta = load_trainer_fixture(tag)
    taken = []
    rows_sweep = sweep._stereo_rows_sweep
    monkeypatch.setattr(sweep, "_stereo_rows_sweep", lambda *a, **k: (taken.append(1), rows_sweep(*a, **k))[1])
    got = run_product_trainer(z, meta, stereo_constant=stereo_constant)
    # the stereo side runs as per-row shifts on the row-shift kernels exactly when its pose is the dataset's constant
    assert len(taken) == (1 if stereo_constant else 0)
    if stereo_constant:
        got.pop("g_Rt_r")
    homo = meta["warp_type"] == "homography_warp"
    exact = run_oracle_trainer(z, meta, dtype=torch.float64) if homo else None   # the same inputs in fp64 arithmetic
    worst = {}
    for k, v in got.items():
        w = z[k]
        if k == "g_disp_layered":
            # the row-shift kernels take the xz planes' map as per-row disparities (constant along x, depth_decoder.py:
            # 163-181) and hand the row's gradient to column 0; the decoder's expand-backward sums over x either way
            v, w = v.sum(-1), w.sum(-1)
        if float(w.abs().max()) == 0.0:
            assert float(v.abs().max()) < 1e-6, (tag, k)
            continue
        e = rel_err(v, w)
        worst[k] = e
        if not homo:
            assert e < TOL, (tag, k, e)
            continue
        # homography_warp end to end: H_t2s = inverse(K (R + t n^T / d) K^-1) is an fp32 torch.inverse in the reference (LAPACK
        # where the fixture was captured) and, by default, pd_homography_matrices_fwd here (fp64 inside, rounded once);
        # cond(H) ~ 1e3-1e4 turns the reference's last-ulp rounding into ~1e-4 coordinate differences (the
        # reference's own disp_warp / homography_warp twins differ by 1.6e-4).  The fp32 evaluation is not "the" answer, so
        # THIS test's bar is a three-way one: the product must be as close to the fp64 evaluation of the same formulas as the
        # reference's own fp32 run is (x2 + the 1e-4 budget).  The direct pin is the test below it: the reference's OWN
        # matrices (captured in the fixture) through every route at the plain 1e-4
        # (test_trainer_mono_fixture_on_the_references_own_matrices).
        ref_vs_exact = rel_err(w, exact[k].float())
        got_vs_exact = rel_err(v, exact[k].float())
        assert got_vs_exact < 2.0 * ref_vs_exact + TOL, (tag, k, e, got_vs_exact, ref_vs_exact)
        if not (k == "g_Rt_r"):   # (stereo pose: every sample on an integer row, ILL_CONDITIONED) — the absolute cap
            assert got_vs_exact < cap_for(k), (tag, k, got_vs_exact, cap_for(k))
    print(tag, {k: "%.1e" % e for k, e in worst.items()})


@pytest.mark.parametrize("tag", ["homo3", "homo_nostereo_l1"])
def test_trainer_mono_fixture_reference_arithmetic_route(tag, monkeypatch):
    """PD_TORCH_HOMOGRAPHY=1: the matrices formed as the reference forms them (the stock fp32 torch chain with
    torch.inverse, layers.py:206-219) instead of the fp64 kernel — the strict route, against the reference-captured
    fixture itself.  What still separates the two is the inverse's backend (rocSOLVER here, LAPACK where the fixture
    was captured) times cond(H) ~ 1e3: the measured distances are printed and held to the caps; profiles/r03_parity.md
    lists them per tensor next to the default route's."""
    from cases import cap_for, load_trainer_fixture
    from gpu_cases import run_product_trainer
    from planedepth_amd import ops
    z, meta = load_trainer_fixture(tag)
    monkeypatch.setattr(ops, "TORCH_HOMOGRAPHY", True)
    got = run_product_trainer(z, meta, stereo_constant=False)
    worst = {}
    for k, v in got.items():
        w = z[k]
        if float(w.abs().max()) == 0.0 or k == "g_Rt_r":     # (stereo pose: ILL_CONDITIONED)
            continue
        worst[k] = rel_err(v, w)
        assert worst[k] < 2.5 * cap_for(k), (tag, k, worst[k])
    print(tag, "PD_TORCH_HOMOGRAPHY vs reference fixture", {k: "%.1e" % e for k, e in worst.items()})


class _ReferenceInverse:
    """Hands the product the H_t2s THE REFERENCE computed: while active, ``torch.inverse`` — the one call of the reference's
    3x3 chain that differs between backends, and the only place the PD_TORCH_HOMOGRAPHY route calls it — returns the
    fixture's reference-captured matrices (tests/golden/make_golden.py: record_inverse), block by block in call order (one
    per target view, trainer.py:532).  Per-plane and stereo-row routes ask for [B*N,3,3]; the plane-uniform route asks for
    [B,4,3,3] (slice 0 = the image's matrix, 1..3 = the virtual planes of the translation's gradient, left as computed)."""

    def __init__(self, blocks, B, N):
        self.blocks, self.B, self.N, self.calls = list(blocks), B, N, 0

    def __enter__(self):
        self._orig = torch.inverse

        def pinned(x, *a, **k):
            ref = self.blocks[self.calls].to(x.device)
            self.calls += 1
            if x.dim() == 3:
                assert tuple(x.shape) == tuple(ref.shape), (x.shape, ref.shape)
                return ref
            assert tuple(x.shape) == (self.B, 4, 3, 3), x.shape
            own = self._orig(x, *a, **k)
            r0 = ref.reshape(self.B, self.N, 3, 3)
            assert torch.equal(r0, r0[:, :1].expand_as(r0)), "the reference's matrices of a zero-translation view differ between planes"
            return torch.cat([r0[:, :1], own[:, 1:].detach()], 1)
        torch.inverse = pinned
        return self

    def __exit__(self, *exc):
        torch.inverse = self._orig
        return False


@pytest.mark.parametrize("tag,route", [("homo3", "per_plane"), ("homo3", "uniform_and_rows"), ("homo3", "uniform_and_per_plane_stereo"),
                                       ("homo_nostereo_l1", "uniform"), ("homo_nostereo_l1", "per_plane")])
def test_trainer_mono_fixture_on_the_references_own_matrices(tag, route, monkeypatch):
    """BASELINE configs[3] pinned to the reference DIRECTLY.  The fixture carries the H_t2s the reference itself
    computed; fed to the product through the trainer path (every route: one homography per plane, one per image for the
    novel frames, the stereo view as row shifts), every tensor the reference produced from them is met at the plain 1e-4 —
    no three-way bound, no caps: with the 3x3 algebra pinned nothing but the kernels is compared."""
    from cases import load_trainer_fixture, side_key
    from gpu_cases import run_product_trainer
    from planedepth_amd import ops
    z, meta = load_trainer_fixture(tag)
    B, N = z["distance"].shape
    monkeypatch.setattr(ops, "TORCH_HOMOGRAPHY", True)
    extra = dict(pd_uniform_homography=route.startswith("uniform"), pd_stereo_rows=route == "uniform_and_rows")
    blocks = [z["H_t2s_%s" % side_key(s)] for s in meta["target_sides"]]
    with _ReferenceInverse(blocks, B, N) as pin:
        got = run_product_trainer(z, meta, stereo_constant=True, opt_extra=extra)
    assert pin.calls == len(blocks)
    worst = {}
    keys = [k for k in got if k.startswith("rgb_rec")] + ["ph_loss", "total_loss", "g_logits"] + (["g_sigma"] if meta["use_mixture_loss"] else [])
    for k in keys:
        worst[k] = rel_err(got[k], z[k])
        assert worst[k] < TOL, (tag, route, k, worst[k])
    print(tag, route, "vs reference on the reference's matrices", {k: "%.1e" % e for k, e in worst.items()})


@pytest.mark.parametrize("name", ["homo_mix_stereo", "homo_mix_pose", "homo_l1_pose"])
def test_small_homography_fixture_on_the_references_own_matrices(name):
    """The small homography fixtures with the reference's own H_t2s handed straight to the per-plane kernels."""
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    case, want, run = load_fixture(name)
    B, N, H, W = case["logits"].shape
    mix = run.get("use_mixture_loss", True)
    dev = "cuda"
    lg, sg = case["logits"].to(dev).requires_grad_(True), case["sigma"].to(dev).requires_grad_(True)
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].expand(B, N, -1)
    Rn = torch.matmul(case["Rt"][:, None, :3, :3], norm[..., None])[..., 0].reshape(B * N, 3)        # layers.py:223
    flags = (C.PD_MIXTURE if mix else 0) | (C.PD_AUTOMASK if run.get("automask", False) else 0)
    rgb, ph, ph_mean = ops._PlaneSweep.apply(case["color_l"].to(dev), case["color_r"].to(dev), lg, sg if mix else None,
                                             want["H_t2s"].to(dev), Rn.contiguous().to(dev), case["inv_K"][:, :3, :3].contiguous().to(dev),
                                             None, None, C.PD_WARP_HOMOGRAPHY, flags, 0.0)
    (ph_mean + (rgb * case["g_rgb_rec"].to(dev)).sum()).backward()
    assert rel_err(rgb.detach().cpu(), want["rgb_rec"]) < TOL
    assert abs(float(ph_mean) - float(want["ph_loss"])) < TOL * abs(float(want["ph_loss"]))
    assert rel_err(lg.grad.cpu(), want["g_logits"]) < TOL
    if mix:
        assert rel_err(sg.grad.cpu(), want["g_sigma"]) < TOL


@pytest.mark.parametrize("impl", ["auto", "general"])
def test_fullsize_homography_on_the_references_own_matrices(impl, monkeypatch):
    """192 x 640 x 63, one homography per plane (rotation + translation), mixture + automask: the reference's matrices and
    outputs of tests/golden/homography_pinned_fullsize.npz (rgb_rec in full, the gradients on a stride-8 lattice plus their
    L1 norms) against the per-plane kernels — gather backward (auto) and atomic backward (general)."""
    import numpy as np
    from conftest import GOLDEN
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import survey_fullsize_case
    z = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(GOLDEN, "homography_pinned_fullsize.npz")).items()}
    case = survey_fullsize_case(B=1, N=63, sigma_interior=True)
    monkeypatch.setattr(ops, "SWEEP_IMPL", C.PD_IMPL_GENERAL if impl == "general" else C.PD_IMPL_AUTO)
    B, N, H, W = case["logits"].shape
    dev = "cuda"
    lg, sg = case["logits"].to(dev).requires_grad_(True), case["sigma"].to(dev).requires_grad_(True)
    Rn = z["Rt"][:, None, :3, 2].expand(B, N, 3).reshape(B * N, 3)            # R n with n = (0, 0, 1)
    rgb, ph, ph_mean = ops._PlaneSweep.apply(case["color_l"].to(dev), case["color_r"].to(dev), lg, sg, z["H_t2s"].to(dev),
                                             Rn.contiguous().to(dev), case["inv_K"][:, :3, :3].contiguous().to(dev), None, None,
                                             C.PD_WARP_HOMOGRAPHY, C.PD_MIXTURE | C.PD_AUTOMASK, 0.0)
    (ph_mean + (rgb * case["g_rgb_rec"].to(dev)).sum()).backward()
    assert rel_err(rgb.detach().cpu(), z["rgb_rec"]) < TOL, rel_err(rgb.detach().cpu(), z["rgb_rec"])
    assert abs(float(ph_mean) - float(z["ph_loss"])) < TOL * float(z["ph_loss"])
    for k, t in (("g_logits", lg.grad.cpu()), ("g_sigma", sg.grad.cpu())):
        e = float((t[..., ::8, ::8] - z[k + "_sub8"]).abs().max()) / float(z["max_" + k])
        assert e < TOL, (k, e)
        assert abs(float(t.double().abs().sum()) - float(z["l1_" + k])) < TOL * float(z["l1_" + k]), k


@pytest.mark.parametrize("tag,stereo_constant", [("homo3", True), ("homo3", False), ("homo_nostereo_l1", False)])
def test_views_as_one_autograd_node_equal_one_node_per_view(tag, stereo_constant):
    """pred_novel_images issues the target views of a step as ONE autograd node whose backward kernels add into the same
    g_logits / g_sigma (PD_BWD_ACCUMULATE: read-modify-write stores in the plane-uniform kernels, no zero-fill in the
    atomic ones, the row-shift view first) — against one node per view with torch's own gradient accumulation."""
    from cases import load_trainer_fixture
    from gpu_cases import run_product_trainer
    z, meta = load_trainer_fixture(tag)
    one = run_product_trainer(z, meta, stereo_constant=stereo_constant)
    per_view = run_product_trainer(z, meta, stereo_constant=stereo_constant, opt_extra=dict(pd_fuse_sides=False))
    for k, v in one.items():
        w = per_view[k]
        if float(w.abs().max()) == 0.0:
            assert float(v.abs().max()) == 0.0, k
            continue
        tol = 2e-5 if k.startswith("g_Rt") or k == "g_distance" else 3e-6   # atomics / partial sums in another order
        assert rel_err(v, w) < tol, (tag, k, rel_err(v, w))


def test_two_row_shift_views_as_one_node():
    """Two disp_warp views ("r" and "l") over the same logits / sigma as one autograd node: neither row-shift backward can
    add in place, so the second goes through a temporary and one add — same result as two nodes."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(B=2, N=6, H=20, W=96, seed=77, disp_min=0.5, disp_max=25.0, sigma_interior=True)
    c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.items()}
    res = {}
    for fused in (True, False):
        lg, sg, dp = (c[k].clone().requires_grad_(True) for k in ("logits", "sigma", "disp_pp"))
        dl = dp.expand(-1, -1, 20, 96)
        calls = [ops.plane_sweep_disp(c["color_l"], c["color_r"], lg, sg, dl, None, target_side=side, return_mean=True,
                                      defer=True) for side in ("r", "l")]
        outs = ops.plane_sweep_multi(calls) if fused else [ops._PlaneSweep.apply(*cc) for cc in calls]
        loss = sum(o[2] * (i + 1.0) + (o[0] * c["g_rgb_rec"]).sum() for i, o in enumerate(outs))
        loss.backward()
        res[fused] = dict(rgb0=outs[0][0].detach().cpu(), rgb1=outs[1][0].detach().cpu(), g_logits=lg.grad.cpu(),
                          g_sigma=sg.grad.cpu(), g_disp=dp.grad.cpu())
    for k in res[True]:
        assert rel_err(res[True][k], res[False][k]) < 2e-6, (k, rel_err(res[True][k], res[False][k]))


@pytest.mark.parametrize("tag,stereo_constant", [("homo3", False), ("homo3", True), ("disp_xz", False)])
def test_trainer_mono_fixture_general_kernels(tag, stereo_constant):
    """The same fixtures forced onto the general kernels (PD_IMPL_GENERAL); with the stereo pose a constant the stereo
    view's per-row shifts are expanded to a dense disparity map for them."""
    from cases import load_trainer_fixture
    from gpu_cases import run_product_trainer
    from planedepth_amd import _capi as C
    from cases import run_oracle_trainer
    z, meta = load_trainer_fixture(tag)
    got = run_product_trainer(z, meta, impl=C.PD_IMPL_GENERAL, stereo_constant=stereo_constant)
    homo = meta["warp_type"] == "homography_warp"
    exact = run_oracle_trainer(z, meta, dtype=torch.float64) if homo else None
    for k in ("ph_loss", "total_loss", "g_logits") + (("g_sigma",) if meta["use_mixture_loss"] else ()):
        if homo:   # three-way, as in test_trainer_mono_fixture_through_patch_trainer
            assert rel_err(got[k], exact[k].float()) < 2.0 * rel_err(z[k], exact[k].float()) + TOL, (tag, k)
        else:
            assert rel_err(got[k], z[k]) < TOL, (tag, k, rel_err(got[k], z[k]))


def _mono_fullsize_case(N_xy=49, N_xz=14, B=1, H=192, W=640, seed=77):
    """Decoder-shaped inputs of BASELINE configs[3] at full size: 49 frontal planes + 14 ground planes with the normal
    [0, 1, c] / |.| and distances of depth_decoder.py:197-207, a pose_net-like motion per image."""
    from planedepth_amd.synthetic import intrinsics, small_pose
    g = torch.Generator().manual_seed(seed)
    N = N_xy + N_xz
    color_l, color_r = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    logits = torch.randn(B, N, H, W, generator=g)
    sigma = 0.011 + 0.978 * torch.rand(B, N, H, W, generator=g)
    lv = torch.arange(N_xy, dtype=torch.float32)[None] + torch.rand(B, N_xy, generator=g) - 0.5
    disp = 300.0 * (2.0 / 300.0) ** (lv / (N_xy - 1))
    distance = 0.1 * 0.58 * W / disp
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].expand(B, N_xy, -1)
    if N_xz:
        hl = 0.1852 + (0.3704 - 0.1852) * (torch.arange(N_xz, dtype=torch.float32)[None] + torch.rand(B, N_xz, generator=g) - 0.5) / (N_xz - 1)
        c = torch.full((B,), 0.07)
        nz = 1.0 / (1.0 + c ** 2) ** 0.5
        xz_norm = torch.stack([torch.zeros(B), torch.ones(B), c], 1) * nz[:, None]
        norm = torch.cat([norm, xz_norm[:, None].expand(-1, N_xz, -1)], 1)
        distance = torch.cat([distance, hl * nz[:, None]], 1)
    K, inv_K = intrinsics(B, H, W)
    Rt = small_pose(g, B, rot=0.01, trans=0.05)
    gw = torch.randn(B, 3, H, W, generator=g) * 1e-5
    return dict(color_l=color_l, color_r=color_r, logits=logits, sigma=sigma, distance=distance, norm=norm.contiguous(),
                K=K, inv_K=inv_K, Rt=Rt, gw=gw)


# Absolute caps at 192x640 (x up to 639: one ulp of the fp32 coordinate is 6e-5 px, and the reference's chain
# x/(W-1) -> ... -> *(W-1) is reproduced operation by operation): measured distances of the product from the fp64
# evaluation, with a factor ~2 (profiles/r03_parity.md).  A kernel error of a percent fails these by an order of magnitude.
FULLSIZE_CAPS = dict(rgb_rec=6e-4, ph_map=1.5e-3, g_logits=1.5e-3, g_sigma=5e-3, g_H=4e-2, g_distance=3e-3)


@pytest.mark.parametrize("mix,automask", [(True, True), (False, False)])
def test_homography_fullsize_63_planes_pinned_matrices(mix, automask):
    """192x640, 49 + 14 planes (ground planes with non-frontal normals: the facing test (K^-1 p).(R n) > 0 of
    layers.py:223-226 bites), a pose_net-like motion: the fused homography kernels against the fp32 AND the fp64 oracle,
    all fed the SAME fp32 H_t2s (so what is compared is the per-pixel path, not torch.inverse's rounding): three-way
    bound, see below."""
    from oracle import planedepth_oracle as orc
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    c = _mono_fullsize_case()
    B, N, H, W = c["logits"].shape
    ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
    H64, Rn64 = orc.homography_matrices(c["distance"].double(), c["norm"].double(), ex(c["Rt"].double()),
                                        ex(c["K"].double()), ex(c["inv_K"].double()))
    Hm = H64.float()
    def oracle(dt):
        cc = {k: v.to(dt) for k, v in c.items()}
        lg, sg, Hl = (cc["logits"].clone().requires_grad_(True), cc["sigma"].clone().requires_grad_(True),
                      Hm.detach().clone().to(dt).requires_grad_(True))
        r = orc.warp_and_loss(cc["color_l"], cc["color_r"], lg, sg if mix else None, warp_type="homography_warp",
                              distance=cc["distance"], norm=cc["norm"], T=cc["Rt"], K=cc["K"], inv_K=cc["inv_K"],
                              use_mixture_loss=mix, automask=automask, H_t2s=Hl)
        (r["ph_loss"] + (r["rgb_rec"] * cc["gw"]).sum()).backward()
        out = dict(rgb_rec=r["rgb_rec"].detach().float(), ph_map=r["ph_map"].detach().float(), g_logits=lg.grad.float(),
                   g_H=Hl.grad.float(), masked=float((r["sweep"]["logit_rec"] == 0).float().mean()))
        if mix:
            out["g_sigma"] = sg.grad.float()
        return out

    o32, o64 = oracle(torch.float32), oracle(torch.float64)
    dev = "cuda"
    lgd, sgd, Hd = (c["logits"].to(dev).requires_grad_(True), c["sigma"].to(dev).requires_grad_(True),
                    Hm.detach().clone().to(dev).requires_grad_(True))
    flags = (C.PD_MIXTURE if mix else 0) | (C.PD_AUTOMASK if automask else 0)
    rgb, ph, ph_mean = ops._PlaneSweep.apply(c["color_l"].to(dev), c["color_r"].to(dev), lgd, sgd if mix else None, Hd,
                                             Rn64.float().reshape(B * N, 3).to(dev), c["inv_K"][:, :3, :3].to(dev), None,
                                             None, C.PD_WARP_HOMOGRAPHY, flags, 0.0)
    (ph_mean + (rgb * c["gw"].to(dev)).sum()).backward()
    got = dict(rgb_rec=rgb.detach().cpu(), ph_map=ph.detach().cpu(), g_logits=lgd.grad.cpu(), g_H=Hd.grad.cpu())
    if mix:
        got["g_sigma"] = sgd.grad.cpu()
    assert 0.02 < o32["masked"] < 0.9, o32["masked"]     # the facing / z tests and the image border really remove samples
    # At x ~ 600 one ulp of the fp32 coordinate is 6e-5 px and the [BN,3,3] x [3,HW] product of layers.py:221 has no
    # defined summation order (MKL here, rocBLAS/cuBLAS on a GPU), so two fp32 evaluations of the reference's formulas
    # differ by 1e-4..2e-3 on white-noise inputs (measured: oracle fp32 vs fp64 3.4e-4 on rgb_rec, 1.9e-3 on g_sigma).
    # The bar is therefore three-way: the kernel must be as close to the fp64 evaluation as the fp32 oracle is.
    for k, v in got.items():
        if k == "g_H" and not mix:
            # L1: d|rgb_rec - tgt|/dH sums sign(rgb_rec - tgt) * (...) over 368 640 values with random signs; a handful of
            # pixels with |rgb_rec - tgt| below the forward's rounding noise flip their sign and move the (heavily
            # cancelling) sum by percents.  Covered at 24x80 (test_homography_kernel_with_pinned_matrices, 2e-4).
            continue
        e_got, e_ref = rel_err(v, o64[k]), rel_err(o32[k], o64[k])
        assert e_got < 1.5 * e_ref + TOL, (k, e_got, e_ref, rel_err(v, o32[k]))
        assert e_got < FULLSIZE_CAPS[k], (k, e_got, FULLSIZE_CAPS[k])   # absolute, against fp64
        print(k, "product %.1e reference-fp32 %.1e (vs fp64)" % (e_got, e_ref))


@pytest.mark.parametrize("B,N_xy,N_xz,H,W,mix,automask", [(1, 49, 14, 192, 640, True, True), (2, 5, 3, 24, 80, True, False),
                                                          (1, 6, 0, 33, 70, False, False), (2, 4, 4, 40, 150, False, True)])
def test_stereo_homography_as_row_shifts(B, N_xy, N_xz, H, W, mix, automask):
    """homography_warp on the stereo side (identity rotation, x-translation, normals without an x component: the warp is
    a shift h01*y + h02 per (plane, row)) routed through the row-shift kernels (ops._stereo_rows_sweep) against the
    general per-plane-homography kernels and the fp64 oracle, end to end from (distance, norm, T, K): rgb_rec, ph_map and
    every gradient the trainer needs (logits, sigma, distance).  Three-way bound as above: the shortcut must be as close
    to the fp64 evaluation as the general fp32 kernels are."""
    from cases import cap_for
    from oracle import planedepth_oracle as orc
    from planedepth_amd import ops
    from planedepth_amd.synthetic import small_pose
    c = _mono_fullsize_case(N_xy=N_xy, N_xz=N_xz, B=B, H=H, W=W, seed=500 + W)
    c["Rt"] = small_pose(None, B, stereo=True)
    c["gw"] = c["gw"] * (1e3 if H < 100 else 1.0)

    def oracle(dt):
        cc = {k: v.to(dt) for k, v in c.items()}
        lg, sg, dd = (cc["logits"].clone().requires_grad_(True), cc["sigma"].clone().requires_grad_(True),
                      cc["distance"].clone().requires_grad_(True))
        r = orc.warp_and_loss(cc["color_l"], cc["color_r"], lg, sg if mix else None, warp_type="homography_warp",
                              distance=dd, norm=cc["norm"], T=cc["Rt"], K=cc["K"], inv_K=cc["inv_K"],
                              use_mixture_loss=mix, automask=automask)
        (r["ph_loss"] + (r["rgb_rec"] * cc["gw"]).sum()).backward()
        out = dict(rgb_rec=r["rgb_rec"].detach().float(), ph_map=r["ph_map"].detach().float(), g_logits=lg.grad.float(),
                   g_distance=dd.grad.float())
        if mix:
            out["g_sigma"] = sg.grad.float()
        return out, float((r["sweep"]["logit_rec"] == 0).float().mean())

    def product(rows):
        dev = "cuda"
        cc = {k: v.to(dev) for k, v in c.items()}
        lg, sg, dd = (cc["logits"].clone().requires_grad_(True), cc["sigma"].clone().requires_grad_(True),
                      cc["distance"].clone().requires_grad_(True))
        rgb, ph, ph_mean = ops.plane_sweep_homography(cc["color_l"], cc["color_r"], lg, sg if mix else None, dd, cc["norm"],
                                                      cc["Rt"], cc["K"], cc["inv_K"], use_mixture_loss=mix,
                                                      automask=automask, return_mean=True, stereo_rows=rows)
        (ph_mean + (rgb * cc["gw"]).sum()).backward()
        out = dict(rgb_rec=rgb.detach().cpu(), ph_map=ph.detach().cpu(), g_logits=lg.grad.cpu(), g_distance=dd.grad.cpu())
        if mix:
            out["g_sigma"] = sg.grad.cpu()
        return out

    (exact, masked), (ref32, _) = oracle(torch.float64), oracle(torch.float32)
    rows, general = product(True), product(False)
    if N_xz:
        assert 0.02 < masked < 0.9, masked     # ground planes above the horizon face away: the per-row mask really bites
    for k, v in rows.items():
        if k == "g_distance" and not mix:
            continue   # L1: sign flips of |rgb_rec - tgt| at the forward's noise level move this heavily cancelling sum
        # the yardstick is the reference's own arithmetic: its formulas evaluated in fp32 (torch.inverse and all) against
        # the fp64 evaluation; the general kernels (fed matrices rounded once from fp64) are reported next to it
        e_rows, e_gen, e_ref = rel_err(v, exact[k]), rel_err(general[k], exact[k]), rel_err(ref32[k], exact[k])
        assert e_rows < 1.5 * max(e_ref, e_gen) + TOL, (k, e_rows, e_gen, e_ref, rel_err(v, general[k]))
        cap = FULLSIZE_CAPS[k] if H >= 100 else cap_for(k)      # absolute, against fp64
        assert e_rows < cap and e_gen < cap, (k, e_rows, e_gen, cap)
        print(k, "rows %.1e general %.1e reference-fp32 %.1e" % (e_rows, e_gen, e_ref))


@pytest.mark.parametrize("mode", ["planes", "uniform", "stereo_rows"])
def test_homography_matrices_kernel_vs_fp64_chain(mode):
    """pd_homography_matrices_fwd/bwd (the 3x3 algebra of layers.py:206-219, 223-225 in one launch, fp64 inside) against
    the same chain written with stock torch operators in fp64 on the CPU: values at fp32 rounding, every gradient
    (distance, norm, pose) through torch.inverse's autograd at 1e-5."""
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics, small_pose
    B, N, H, W = 3, 11, 37, 120
    g = torch.Generator().manual_seed(4242)
    distance = 0.5 + 5 * torch.rand(B, N, generator=g)
    norm = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g) * 0.4 + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    K, inv_K = intrinsics(B, H, W)
    T = small_pose(g, B, rot=0.05, trans=0.1)
    if mode == "uniform":
        T = _f8_pose(B, 5, rot=0.05)
    if mode == "stereo_rows":
        T = small_pose(None, B, stereo=True)
        norm[..., 0] = 0.0
        norm[:, N // 2:, 1] += 3.0        # ground-like planes: the facing test changes sign inside the image
        norm = torch.nn.functional.normalize(norm, dim=-1)
    d64, n64, T64 = (t.double().requires_grad_(True) for t in (distance, norm, T))
    K64, Ki64 = K.double(), inv_K.double()
    ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
    dev = "cuda"
    dd, nd, Td = (t.to(dev).requires_grad_(True) for t in (distance, norm, T))
    if mode == "planes":
        Hr, Rnr = ops.homography_matrices(d64, n64, ex(T64), ex(K64), ex(Ki64))
        Hg, Rng = ops.homography_matrices_fused(dd, nd, Td, K.to(dev), inv_K.to(dev))
        Hr = Hr.reshape(B, N, 3, 3)
    elif mode == "uniform":
        Rm, t = T64[:, :3, :3], T64[:, :3, 3:4]
        eye = torch.eye(3, dtype=torch.float64)
        Rtnd = torch.cat([(Rm + torch.matmul(t.detach(), n64[:, 0].reshape(B, 1, 3)) / d64[:, 0].reshape(B, 1, 1))[:, None],
                          Rm.detach()[:, None] + t[:, None] * eye.reshape(1, 3, 1, 3)], 1)
        Hr = torch.inverse(torch.matmul(K64[:, None, :3, :3], torch.matmul(Rtnd, Ki64[:, None, :3, :3])))
        Rnr = torch.matmul(Rm[:, None], n64.reshape(B, N, 3, 1))[..., 0]
        Hg, Rng = ops.homography_matrices_fused(dd, nd, Td, K.to(dev), inv_K.to(dev), C.PD_HMAT_UNIFORM)
    else:
        Hm, Rnr = ops.homography_matrices(d64, n64, ex(T64), ex(K64), ex(Ki64))
        Hm = Hm.reshape(B, N, 3, 3)
        y = torch.arange(H, dtype=torch.float64).reshape(1, 1, H)
        Hr = Hm[:, :, 0, 1, None] * y + Hm[:, :, 0, 2, None]
        ray = Ki64[:, None, :3, 1, None] * y[..., None, :] + Ki64[:, None, :3, 2, None]
        facing = (ray * Rnr.reshape(B, N, 3, 1)).sum(2)
        Hg, maskg, Rng = ops.homography_matrices_fused(dd, nd.detach(), Td.detach(), K.to(dev), inv_K.to(dev),
                                                       C.PD_HMAT_STEREO_ROWS, rows=H)
        sure = facing.abs() > 1e-5       # away from the knife edge the per-row mask is the facing test's sign
        assert torch.equal(maskg.cpu()[sure] > 0, facing[sure] > 0)
        assert 0.02 < float(maskg.mean()) < 0.98
    assert rel_err(Hg.detach().cpu(), Hr.detach().float().reshape(Hg.shape)) < 3e-7
    assert rel_err(Rng.detach().cpu().reshape(B, N, 3), Rnr.detach().float().reshape(B, N, 3)) < 3e-7
    gH = torch.randn(Hr.shape, generator=g, dtype=torch.float64)
    (Hr * gH).sum().backward()
    (Hg * gH.float().to(dev).reshape(Hg.shape)).sum().backward()
    if mode == "uniform":   # zero translation: planes carry no gradient; the pose gets rotation AND translation columns
        assert dd.grad is None or float(dd.grad.abs().max()) == 0.0
        assert rel_err(Td.grad.cpu()[:, :3], T64.grad.float()[:, :3]) < 1e-5
        assert float(T64.grad[:, :3, 3].abs().max()) > 0
        return
    assert rel_err(dd.grad.cpu(), d64.grad.float()) < 1e-5
    if mode == "planes":
        assert rel_err(nd.grad.cpu(), n64.grad.float()) < 1e-5
        assert rel_err(Td.grad.cpu()[:, :3], T64.grad.float()[:, :3]) < 1e-5
        assert float(Td.grad[:, 3].abs().max()) == 0.0


@pytest.mark.parametrize("mix,automask,with_mask", [(False, False, True), (False, True, True), (True, False, True),
                                                    (False, True, False)])
def test_masked_photometric_loss_vs_torch_expression(mix, automask, with_mask):
    """pd_masked_photometric_fwd/bwd against the reference's own statements (trainer.py:724-742) written in torch: the
    blended prediction, the scalar loss, and the gradients into rgb_rec / ph_map, with an upstream gradient on the
    blended prediction (the perceptual net's) on top of the loss's."""
    from planedepth_amd import ops
    g = torch.Generator().manual_seed(77)
    B, H, W = 3, 37, 150
    dev = "cuda"
    rgb, tgt, src = (torch.rand(B, 3, H, W, generator=g).to(dev) for _ in range(3))
    ph_map = (torch.rand(B, 1, H, W, generator=g) * 5).to(dev)
    mask = torch.rand(B, 1, H, W, generator=g).clamp(0.1, 0.8).sub(0.1).div(0.7).to(dev) if with_mask else None  # zeros and ones included
    gw = torch.randn(B, 3, H, W, generator=g).to(dev) * 1e-5

    def reference(r, pm):
        m = mask if mask is not None else torch.ones(B, 1, H, W, device=dev)
        pred = r * m + tgt * (1.0 - m)                                         # trainer.py:726
        if mix:
            ph = pm * m                                                        # :736
        else:
            ph = torch.abs(pred - tgt).mean(1, True)                          # :738
            if automask:
                ph_auto = torch.abs(src - tgt).mean(1, True)                   # :740
                ph, _ = torch.cat([ph, ph_auto], dim=1).min(1, True)           # :741
        return pred, ph.mean()                                                 # :742

    def product(r, pm):
        return ops.masked_photometric(r, tgt, mask, source=src if (automask and not mix) else None,
                                      ph_map=pm if mix else None)

    out = {}
    for name, fn in (("ref", reference), ("hip", product)):
        r, pm = rgb.clone().requires_grad_(True), ph_map.clone().requires_grad_(True)
        pred, loss = fn(r, pm)
        (loss * 3.0 + (pred * gw).sum()).backward()
        out[name] = dict(pred=pred.detach().cpu(), loss=loss.detach().cpu(), g_rgb=r.grad.cpu(),
                         g_ph=pm.grad.cpu() if pm.grad is not None else torch.zeros(1))
    assert torch.equal(out["hip"]["pred"], out["ref"]["pred"])                 # same three roundings per element
    assert abs(float(out["hip"]["loss"]) - float(out["ref"]["loss"])) < 2e-6 * abs(float(out["ref"]["loss"]))
    assert rel_err(out["hip"]["g_rgb"], out["ref"]["g_rgb"]) < 1e-6
    if mix:
        assert rel_err(out["hip"]["g_ph"], out["ref"]["g_ph"]) < 1e-6


def _wild_homographies(B, N, H, W, seed):
    """[B*N,3,3] target->source homographies well away from the identity: in-plane rotation up to ~12 degrees, zoom
    0.8-1.25, shear, a perspective term, shifts of up to a third of the image — plus their (K^-1, R n) companions."""
    g = torch.Generator().manual_seed(seed)
    M = B * N
    ang = (torch.rand(M, generator=g) - 0.5) * 0.4
    zoom = 0.8 + 0.45 * torch.rand(M, generator=g)
    A = torch.zeros(M, 3, 3)
    A[:, 0, 0], A[:, 0, 1] = zoom * torch.cos(ang), -zoom * torch.sin(ang) + 0.05 * torch.randn(M, generator=g)
    A[:, 1, 0], A[:, 1, 1] = zoom * torch.sin(ang), zoom * torch.cos(ang)
    A[:, 0, 2] = (torch.rand(M, generator=g) - 0.5) * 0.66 * W
    A[:, 1, 2] = (torch.rand(M, generator=g) - 0.5) * 0.66 * H
    A[:, 2, 0] = torch.randn(M, generator=g) * 2e-4
    A[:, 2, 1] = torch.randn(M, generator=g) * 2e-4
    A[:, 2, 2] = 1.0
    # rotate about the image centre rather than the corner
    C = torch.eye(3)[None].repeat(M, 1, 1)
    C[:, 0, 2], C[:, 1, 2] = W / 2.0, H / 2.0
    Ci = C.clone()
    Ci[:, 0, 2], Ci[:, 1, 2] = -W / 2.0, -H / 2.0
    Hm = C @ A @ Ci
    Hm[:, :2, 2] += A[:, :2, 2] * 0.0
    Rn = torch.nn.functional.normalize(torch.randn(M, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    return Hm.contiguous(), Rn.contiguous()


def _gather_case(B, N, H, W, seed, irregular=False):
    """Inputs of a per-plane homography sweep far from the identity; ``irregular``: some planes the gather backward must
    hand to its atomic fix-up (4x minification, the line at infinity inside the view, a NaN and a singular matrix)."""
    from planedepth_amd.synthetic import intrinsics
    g = torch.Generator().manual_seed(seed)
    src, tgt = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    logits = torch.randn(B, N, H, W, generator=g)
    sigma = 0.011 + 0.978 * torch.rand(B, N, H, W, generator=g)
    gw = torch.randn(B, 3, H, W, generator=g)
    Hm, Rn = _wild_homographies(B, N, H, W, seed + 1)
    if irregular:
        Hm = Hm.clone()
        Hm[0] = torch.tensor([[0.25, 0.0, 0.3 * W], [0.0, 0.25, 0.3 * H], [0.0, 0.0, 1.0]])      # 16 target pixels per source pixel
        Hm[1] = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [2.0 / W, 0.0, 0.2]])          # strong perspective
        if Hm.shape[0] > 3:
            Hm[2] = float("nan")
            Hm[3] = 0.0
        Rn = Rn.clone()
        Rn[:2] = torch.tensor([0.0, 0.0, 1.0])
    _, inv_K = intrinsics(B, H, W)
    return src, tgt, logits, sigma, gw, Hm, Rn, inv_K[:, :3, :3].contiguous()


def _run_gather_case(case, impl, mix=True, render=False, dists=None):
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    src, tgt, logits, sigma, gw, Hm, Rn, iK = [t.cuda() for t in case]
    flags = (C.PD_MIXTURE if mix else 0) | C.PD_AUTOMASK | (C.PD_RENDER_PROB if render else 0)
    ops.SWEEP_IMPL = impl
    ops.DEBUG_WORKSPACE = []
    try:
        lg, sg, Hd = logits.clone().requires_grad_(True), sigma.clone().requires_grad_(True), Hm.clone().requires_grad_(True)
        dd = dists.cuda().clone().requires_grad_(True) if render else None
        rgb, ph, ph_mean = ops._PlaneSweep.apply(src, tgt, lg, sg if mix else None, Hd, Rn, iK, None, dd,
                                                 C.PD_WARP_HOMOGRAPHY, flags, 0.0)
        (ph_mean * 3.0 + (rgb * gw).sum()).backward()
        out = dict(g_logits=lg.grad.cpu(), g_sigma=sg.grad.cpu() if mix else None, g_H=Hd.grad.cpu(),
                   g_dists=dd.grad.cpu() if render else None)
        flags_out = None
        if impl in (C.PD_IMPL_AUTO, C.PD_IMPL_UNIFORM_DIRECT):
            import ctypes
            d, ws = ops.DEBUG_WORKSPACE[-1]
            host = (ctypes.c_int * 2)()
            C.check(C.load().pd_debug_gather_flags(ctypes.byref(d), C.ptr(ws), host, C.stream_handle(ws.device)),
                    "pd_debug_gather_flags")
            flags_out = (host[0], host[1])
        return out, flags_out
    finally:
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
        ops.DEBUG_WORKSPACE = None


@pytest.mark.parametrize("B,N,H,W,mix,irregular", [
    (2, 5, 40, 150, True, False), (1, 9, 33, 70, True, False), (1, 3, 50, 200, False, False), (1, 4, 5, 7, True, False),
    (2, 6, 64, 64, True, False), (1, 5, 96, 320, True, False), (1, 49, 192, 640, True, False),
    (1, 1, 16, 40, True, False), (2, 2, 16, 40, True, False),   # one / two planes: the staging pipeline's prologue alone
    (2, 5, 40, 150, True, True), (1, 4, 33, 70, False, True), (1, 6, 96, 320, True, True)])
def test_gather_backward_equals_atomic_backward(B, N, H, W, mix, irregular):
    """homography_warp with one matrix per plane (6-DoF poses): the two-pass gather backward (pd_plane_sweep_gather.hip,
    the default) against the atomic scatter (PD_IMPL_GENERAL) — two independent adjoints of the same gather, equal to
    summation order — on homographies far from the identity, ragged sizes, and with planes the gather hands to its
    atomic fix-up (minification, line at infinity, NaN and singular matrices)."""
    from planedepth_amd import _capi as C
    case = _gather_case(B, N, H, W, 300 + W + N, irregular)
    new, fl = _run_gather_case(case, C.PD_IMPL_AUTO, mix)
    new2, _ = _run_gather_case(case, C.PD_IMPL_AUTO, mix)
    direct, fl_direct = _run_gather_case(case, C.PD_IMPL_UNIFORM_DIRECT, mix)   # pass 2 without the LDS staging
    old, _ = _run_gather_case(case, C.PD_IMPL_GENERAL, mix)
    assert fl == ((1, 0) if irregular else (0, 0)) and fl_direct == fl, (fl, fl_direct)
    assert float(old["g_logits"].abs().max()) > 0

    def clean(t):   # the NaN matrix's plane carries NaN homography gradients in both forms
        return torch.nan_to_num(t, nan=0.0)
    for k in ("g_logits", "g_sigma"):
        if new[k] is None:
            continue
        assert torch.isfinite(new[k]).all()
        assert rel_err(new[k], old[k]) < 2e-6, (k, rel_err(new[k], old[k]))
        if not irregular:
            assert torch.equal(new[k], new2[k]), k + ": the gather form is deterministic"
            assert torch.equal(new[k], direct[k]), k + ": staged and direct pass 2 add the same numbers in the same order"
    assert rel_err(clean(new["g_H"]), clean(old["g_H"])) < 5e-5


def test_gather_backward_accumulates_and_serves_render_probability():
    """PD_BWD_ACCUMULATE (a second target view adds into the first one's gradients) and PD_RENDER_PROB on the gather
    backward, against the atomic kernels."""
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    B, N, H, W = 2, 6, 48, 160
    case = _gather_case(B, N, H, W, 77, irregular=True)
    dists = torch.rand(B, N - 1, H, W, generator=torch.Generator().manual_seed(5)) * 2.0
    new, fl = _run_gather_case(case, C.PD_IMPL_AUTO, True, render=True, dists=dists)
    old, _ = _run_gather_case(case, C.PD_IMPL_GENERAL, True, render=True, dists=dists)
    assert fl == (1, 0)
    for k in ("g_logits", "g_sigma", "g_dists"):
        assert rel_err(new[k], old[k]) < 2e-6, (k, rel_err(new[k], old[k]))
    # accumulate: base + gradient
    src, tgt, logits, sigma, gw, Hm, Rn, iK = [t.cuda() for t in case]
    flags = C.PD_MIXTURE | C.PD_AUTOMASK
    (rgb, ph, ph_mean), saved = ops._sweep_forward(src, tgt, logits, sigma, Hm, Rn, iK, None, None, C.PD_WARP_HOMOGRAPHY, flags, 0.0)
    grads = (gw, None, torch.full((1,), 3.0, device="cuda"))
    cfg = (C.PD_WARP_HOMOGRAPHY, flags, 0.0)
    gl, gs, _, _ = ops._sweep_backward(saved, cfg, grads, (True, True, False, False))
    base_l, base_s = torch.randn_like(logits) * gl.abs().max(), torch.randn_like(sigma) * gs.abs().max()
    into = (base_l.clone(), base_s.clone())
    ops._sweep_backward(saved, cfg, grads, (True, True, False, False), into=into, accumulate=True)
    assert rel_err(into[0] - base_l, gl) < 1e-5 and rel_err(into[1] - base_s, gs) < 1e-5   # (base + g) - base in fp32


def test_rowshift_adjoint_cross_row_term_is_bounded_at_large_height():
    """the row kernels' adjoint drops the eps-weighted contribution to the neighbouring source row (the vertical
    round trip y -> normalise -> un-normalise returns y + e; e grows with H).  Bound the difference to the general
    kernels, which keep the term, at H = 1024 (the largest height of any BASELINE config is 384)."""
    from gpu_cases import run_product
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(B=1, N=4, H=1024, W=96, seed=55, disp_min=0.5, disp_max=30.0, sigma_interior=True)
    run = dict(automask=True)
    res = {}
    for impl in (C.PD_IMPL_AUTO, C.PD_IMPL_ROWS1, C.PD_IMPL_GENERAL):
        ops.SWEEP_IMPL = impl
        try:
            res[impl] = run_product(case, run)
        finally:
            ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    gen = res[C.PD_IMPL_GENERAL]
    for impl in (C.PD_IMPL_AUTO, C.PD_IMPL_ROWS1):
        _compare(res[impl], gen, keys=("rgb_rec", "ph_map", "g_disp_pp"), tag="H1024/%d" % impl, tol=2e-5)
        _compare(res[impl], gen, keys=("g_logits", "g_sigma"), tag="H1024/%d" % impl, tol=1e-4)


def test_contract_check_catches_tensors_that_disagree_with_the_options():
    """pred_novel_images takes its mask / row-uniformity shortcuts from opt; opt.pd_check_contract verifies them
    on the tensors and raises instead of warping silently wrong."""
    import types
    import planedepth_amd
    from planedepth_amd.synthetic import build_case
    c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in
         build_case(B=1, N=4, H=8, W=32, seed=3, disp_min=0.5, disp_max=5.0, sigma_interior=True).items()}
    B, N, H, W = c["logits"].shape
    opt = types.SimpleNamespace(warp_type="disp_warp", match_aug=False, use_mixture_loss=True, automask=False,
                                render_probability=False, xz_levels=0, yz_levels=0, pd_check_contract=True)
    ns = types.SimpleNamespace(opt=opt, target_sides=["r"])
    inputs = {("color", "l"): c["color_l"], ("color", "r"): c["color_r"]}
    base = {"probability": torch.empty(B, N, H, W, device="meta"), "logits": c["logits"], "sigma": c["sigma"],
            "disp_layered": c["disp_pp"].expand(-1, -1, H, W), "padding_mask": torch.ones(B, N, H, W, device="cuda")}
    planedepth_amd.pred_novel_images(ns, inputs, dict(base))                      # consistent: fine
    bad_mask = dict(base, padding_mask=base["padding_mask"].clone())
    bad_mask["padding_mask"][0, 1, 2, 3] = 0.0
    with pytest.raises(ValueError, match="all-ones padding_mask"):
        planedepth_amd.pred_novel_images(ns, inputs, bad_mask)
    bad_disp = dict(base, disp_layered=(c["disp_pp"].expand(-1, -1, H, W) + torch.rand(B, N, H, W, device="cuda")))
    with pytest.raises(ValueError, match="constant along x"):
        planedepth_amd.pred_novel_images(ns, inputs, bad_disp)
    # without the option: the FIRST call of a trainer object checks (one host sync), later calls trust the cached verdict
    opt2 = types.SimpleNamespace(**{k: v for k, v in vars(opt).items() if k != "pd_check_contract"})
    fresh = types.SimpleNamespace(opt=opt2, target_sides=["r"])
    with pytest.raises(ValueError, match="all-ones padding_mask"):
        planedepth_amd.pred_novel_images(fresh, inputs, dict(bad_mask))
    seasoned = types.SimpleNamespace(opt=opt2, target_sides=["r"])
    planedepth_amd.pred_novel_images(seasoned, inputs, dict(base))
    assert seasoned._pd_contract_checked
    planedepth_amd.pred_novel_images(seasoned, inputs, dict(bad_mask))            # no check any more: no sync either


@pytest.mark.parametrize("mode", ["planes", "uniform", "stereo_rows"])
def test_stock_torch_homography_algebra_still_serves_every_route(mode, monkeypatch):
    """PD_TORCH_HOMOGRAPHY=1 (ops.TORCH_HOMOGRAPHY): the 3x3 algebra as the reference's stock torch chain (fp32
    torch.inverse) instead of pd_homography_matrices_*, on all three routes of plane_sweep_homography.  Loose bound: the
    two differ by the inverse's fp32 rounding times cond(H); what is tested is that the comparison
    switch works end to end, gradients included."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics, small_pose
    B, N, H, W = 2, 5, 16, 64
    g = torch.Generator().manual_seed(808)
    dev = "cuda"
    src, tgt = torch.rand(B, 3, H, W, generator=g).to(dev), torch.rand(B, 3, H, W, generator=g).to(dev)
    logits = torch.randn(B, N, H, W, generator=g).to(dev)
    sigma = (0.05 + 0.9 * torch.rand(B, N, H, W, generator=g)).to(dev)
    distance = (1.0 + 4 * torch.rand(B, N, generator=g)).to(dev)
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1).to(dev)
    K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
    Rt = {"planes": small_pose(g, B, rot=0.02, trans=0.05), "uniform": _f8_pose(B, 3, 0.02),
          "stereo_rows": small_pose(None, B, stereo=True)}[mode].to(dev)
    res = {}
    for stock in (False, True):
        monkeypatch.setattr(ops, "TORCH_HOMOGRAPHY", stock)
        lg, sg, dd = (t.clone().requires_grad_(True) for t in (logits, sigma, distance))
        T = Rt.clone().requires_grad_(mode != "stereo_rows")
        rgb, ph, ph_mean = ops.plane_sweep_homography(src, tgt, lg, sg, dd, norm, T, K, inv_K, return_mean=True,
                                                      plane_uniform=mode == "uniform", stereo_rows=mode == "stereo_rows")
        (ph_mean + rgb.sum() * 1e-3).backward()
        res[stock] = dict(rgb=rgb.detach().cpu(), g_logits=lg.grad.cpu(), g_sigma=sg.grad.cpu())
        if mode != "uniform":
            res[stock]["g_distance"] = dd.grad.cpu()
        if mode != "stereo_rows":
            res[stock]["g_Rt"] = T.grad.cpu()[:, :3]
    for k in res[True]:
        assert rel_err(res[True][k], res[False][k]) < 5e-3, (mode, k, rel_err(res[True][k], res[False][k]))


@pytest.mark.parametrize("rot,zoom", [(0.4, 1.0), (0.25, 0.6), (0.05, 1.0)])
def test_uniform_backward_never_reads_shared_memory_it_did_not_write(rot, zoom):
    """Regression: a source tile whose pre-image lies outside the target image (the view turned away from it) staged
    nothing into LDS, yet unused gather entries (weight 0, slot 0) were still read: 0 * whatever the CU's LDS held, i.e.
    NaN once in a while (fresh boxes hold ~1e-5 NaN patterns in LDS, scripts/diag_lds.py).  With the LDS of the device
    poisoned with NaNs first, any such read shows up deterministically."""
    from planedepth_amd import _capi as C
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics
    B, N, H, W = 1, 4, 30, 90
    g = torch.Generator().manual_seed(990)
    dev = "cuda"
    src, tgt = torch.rand(B, 3, H, W, generator=g).to(dev), torch.rand(B, 3, H, W, generator=g).to(dev)
    K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1).to(dev)
    distance = (1.0 + torch.rand(B, N, generator=g)).to(dev)
    Rt = _f8_pose(B, 61, rot, dev)
    Rt[:, :2, :3] *= zoom
    for _ in range(3):
        lg = torch.randn(B, N, H, W, generator=g).to(dev).requires_grad_(True)
        sg = (0.05 + 0.9 * torch.rand(B, N, H, W, generator=g)).to(dev).requires_grad_(True)
        rgb, ph, ph_mean = ops.plane_sweep_homography(src, tgt, lg, sg, distance, norm, Rt, K, inv_K, return_mean=True,
                                                      plane_uniform=True)
        C.check(C.load().pd_debug_poison_lds(C.stream_handle()), "pd_debug_poison_lds")
        (ph_mean + rgb.sum() * 1e-3).backward()
        assert bool(torch.isfinite(lg.grad).all()) and bool(torch.isfinite(sg.grad).all())


def test_two_host_threads_on_one_device_equal_the_serial_result():
    """The header's threading statement (include/planedepth_hip.h: per-device caches in atomics, the rare dynamic-LDS raise
    under a mutex): two Python threads drive the path on the SAME device at the same time, each on its own stream, at a shape
    whose kernels need more than the default 64 KB of LDS per workgroup (384 x 1280: the persistent forward's double row
    buffers, the packed row-stream backward) — the first call of each thread races the other for the one-time
    hipFuncSetAttribute.  Results must be bit-identical to the same cases run one after the other (the per-plane disparity gradient, summed with LDS
    float atomics, to rounding)."""
    import threading
    from gpu_cases import run_product
    from planedepth_amd.synthetic import survey_fullsize_case
    cases = [survey_fullsize_case(B=1, N=11, H=96, W=1280, seed=500 + i, sigma_interior=True) for i in range(2)]
    extra = dict(xz_levels=0, yz_levels=0)
    out, errs = [None, None], []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    out[i] = run_product(cases[i], {}, opt_extra=extra)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(2):
        serial = run_product(cases[i], {}, opt_extra=extra)
        for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma"):
            assert torch.equal(out[i][k], serial[k]), (i, k)
        assert rel_err(out[i]["g_disp_pp"], serial["g_disp_pp"]) < 1e-5   # (summed with float atomics: order-dependent rounding)


def test_plane_gradient_added_into_a_zeroed_block_equals_the_reduced_partials(monkeypatch):
    """PD_BWD_PLANE_ZEROED (include/planedepth_hip.h): the row-stream backward adds every row's share of the per-plane
    disparity gradient into the caller's pre-zeroed [B, N] block instead of writing per-row partials for a reduction launch.
    Both forms against each other (rounding: the order of the float adds differs) and, through the usual parity cases, against
    the oracle; blocks handed out by the pool are never written twice (an earlier step's gradient keeps its value)."""
    import ctypes
    from gpu_cases import run_product
    from planedepth_amd import _capi as C, ops
    from planedepth_amd.synthetic import survey_fullsize_case
    case = survey_fullsize_case(B=3, N=13, H=48, W=320, seed=77, sigma_interior=True)
    extra = dict(xz_levels=0, yz_levels=0)
    d = ops._desc(3, 13, 48, 320, C.PD_WARP_DISP, C.PD_MIXTURE | C.PD_AUTOMASK, -1.0)
    assert C.load().pd_sweep_bwd_plane_adds(ctypes.byref(d)) == 1
    monkeypatch.setattr(ops, "PLANE_ADDS", True)
    first = run_product(case, {}, opt_extra=extra)
    kept = first["g_disp_pp"].clone()
    again = run_product(case, {}, opt_extra=extra)
    monkeypatch.setattr(ops, "PLANE_ADDS", False)
    ref = run_product(case, {}, opt_extra=extra)
    assert float(ref["g_disp_pp"].abs().max()) > 0
    assert rel_err(first["g_disp_pp"], ref["g_disp_pp"]) < 1e-5
    assert rel_err(again["g_disp_pp"], ref["g_disp_pp"]) < 1e-5
    assert torch.equal(first["g_disp_pp"], kept)   # the second call got a block of its own
    for k in ("g_logits", "g_sigma"):
        assert torch.equal(first[k], ref[k]), k


def test_launches_follow_torchs_current_stream():
    """The C ABI takes an explicit stream and the Python layer hands it torch's CURRENT one: the whole path run inside a
    side-stream context (inputs produced on that stream right before, no synchronisation in between) gives the result
    of the default stream.  A launch on the wrong stream would race with the producer kernels."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(B=2, N=9, H=24, W=200, seed=321, disp_min=0.5, disp_max=40.0, sigma_interior=True)
    c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.items()}

    def run():
        # the inputs are (re)computed on the current stream immediately before the sweep: a misplaced launch reads them early
        lg = (c["logits"] * 1.0 + 0.0).requires_grad_(True)
        sg = (c["sigma"] * 1.0).requires_grad_(True)
        dp = c["disp_pp"].clone().requires_grad_(True)
        rgb, ph, ph_mean = ops.plane_sweep_disp(c["color_l"] * 1.0, c["color_r"] * 1.0, lg, sg, dp.expand(-1, -1, 24, 200),
                                                None, automask=True, return_mean=True)
        (ph_mean + (rgb * c["g_rgb_rec"]).sum()).backward()
        return [t.detach().clone() for t in (rgb, ph, lg.grad, sg.grad, dp.grad)]

    want = run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            got = run()
    side.synchronize()
    for i_, (a_, b_) in enumerate(zip(got, want)):
        assert rel_err(a_.cpu(), b_.cpu()) < (1e-5 if i_ == 4 else 1e-6)   # (4: the per-plane disparity gradient, float atomics)


def test_step_replayed_from_a_hip_graph_equals_the_eager_step():
    """bench.py may time the step as a HIP-graph replay (``--launch auto`` takes the faster form) and a trainer may capture it:
    the whole step — forward, fused mean, backward with the per-plane disparity gradient — recorded once in a torch CUDAGraph
    and replayed three times gives the eager result every time.  What could break it: the pre-zeroed slots (``ph_mean`` under
    PD_PH_MEAN_ZEROED, ``g_plane`` under PD_BWD_PLANE_ZEROED) are added INTO by the kernels, so a replay has to zero them in
    the captured work itself (ops._zero_scalar / _zero_block hand out a captured ``zeros`` while the stream is capturing)."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_replay_check.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "graph replay equals eager: ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_randomised_shapes_homography_shortcuts_vs_general():
    """Seeded sweep over odd shapes (heights / widths from 2 up, not multiples of the 32 x 8 tiles, fewer planes than a
    staging group, batch 1..3, rotations up to ~15 degrees, zooms, L1 / mixture / automask / compositing, one or two
    novel views as one autograd node with in-kernel accumulation): the plane-uniform kernels and the stereo view's
    per-row-shift route against the general per-plane-homography kernels on the same inputs."""
    import random
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics, small_pose
    rnd = random.Random(77)
    dev = "cuda"
    for trial in range(48):
        B, N = rnd.randint(1, 3), rnd.randint(1, 9)
        H, W = rnd.choice([2, 3, 7, 8, 9, 17, 33]), rnd.choice([2, 5, 31, 32, 33, 64, 70, 130])
        mix, automask, render = rnd.random() < 0.7, rnd.random() < 0.5, rnd.random() < 0.4 and N >= 2
        views = rnd.choice([["pose"], ["pose", "pose2"], ["stereo", "pose"]])
        g = torch.Generator().manual_seed(5000 + trial)
        src = torch.rand(B, 3, H, W, generator=g).to(dev)
        tgts = [torch.rand(B, 3, H, W, generator=g).to(dev) for _ in views]
        logits = torch.randn(B, N, H, W, generator=g)
        if render:
            logits = logits.abs() + 0.05     # away from the relu's kink (see test_render_probability_on_the_homography_shortcuts)
        logits = logits.to(dev)
        sigma = (0.011 + 0.978 * torch.rand(B, N, H, W, generator=g)).to(dev)
        dists = (torch.rand(B, max(N - 1, 1), H, W, generator=g) * 2.0).to(dev) if render else None
        gws = [(torch.randn(B, 3, H, W, generator=g) * 0.1).to(dev) for _ in views]
        distance = (0.5 + 5 * torch.rand(B, N, generator=g)).to(dev)
        norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1)
        if N >= 2:
            norm[:, N // 2:] = torch.nn.functional.normalize(torch.tensor([0.0, 1.0, 0.07]), dim=0)
        norm = norm.to(dev)
        K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
        poses = []
        for i, v in enumerate(views):
            if v == "stereo":
                poses.append(small_pose(None, B, stereo=True).to(dev))
            else:
                P = _f8_pose(B, 100 + trial + i, rnd.choice([0.01, 0.05, 0.25]), dev)
                P[:, :2, :3] *= rnd.choice([1.0, 1.0, 0.7, 1.6])
                poses.append(P)
        res = {}
        for fast in (True, False):
            lg, sg = logits.clone().requires_grad_(True), sigma.clone().requires_grad_(True)
            ds = dists.clone().requires_grad_(True) if render else None
            calls = [ops.plane_sweep_homography(src, t, lg, sg if mix else None, distance, norm, P, K, inv_K,
                                                use_mixture_loss=mix, automask=automask and mix, render_probability=render,
                                                dists=ds, return_mean=True, defer=True,
                                                plane_uniform=fast and v != "stereo", stereo_rows=fast and v == "stereo")
                     for t, P, v in zip(tgts, poses, views)]
            outs = ops.plane_sweep_multi(calls) if fast else [ops._PlaneSweep.apply(*c) for c in calls]
            loss = sum(o[2] * (1.0 + i) + (o[0] * gw).sum() for i, (o, gw) in enumerate(zip(outs, gws)))
            loss.backward()
            res[fast] = dict(rgb=torch.stack([o[0].detach() for o in outs]).cpu(), g_logits=lg.grad.cpu(),
                             g_sigma=sg.grad.cpu() if mix else torch.zeros(1),
                             g_dists=ds.grad.cpu() if render else torch.zeros(1))
        tag = "trial%d B%d N%d %dx%d mix%d am%d render%d %s" % (trial, B, N, H, W, mix, automask, render, views)
        for k in res[True]:
            a_, b_ = res[True][k], res[False][k]
            assert bool(torch.isfinite(a_).all()), (tag, k)
            if float(b_.abs().max()) == 0.0:
                assert float(a_.abs().max()) < 1e-6, (tag, k)
                continue
            # the stereo view's row kernels follow their own (the reference's disp_warp) coordinate chain: 2e-4 (NOTEBOOK 3.5.2)
            tol = 2e-4 if "stereo" in views else 5e-6
            assert rel_err(a_, b_) < tol, (tag, k, rel_err(a_, b_))


def test_non_finite_pose_does_not_stall_the_uniform_backward():
    """A diverged pose net hands over NaNs: the plane-uniform backward must come back at once (empty gather windows for
    that image) instead of scanning the whole image per source pixel, and the healthy image of the batch must be
    unaffected."""
    import time
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics
    B, N, H, W = 2, 5, 48, 160
    g = torch.Generator().manual_seed(12)
    dev = "cuda"
    src, tgt = torch.rand(B, 3, H, W, generator=g).to(dev), torch.rand(B, 3, H, W, generator=g).to(dev)
    logits = torch.randn(B, N, H, W, generator=g).to(dev)
    sigma = (0.05 + 0.9 * torch.rand(B, N, H, W, generator=g)).to(dev)
    K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1).to(dev)
    distance = (1.0 + torch.rand(B, N, generator=g)).to(dev)
    good = _f8_pose(B, 7, 0.03, dev)
    res = {}
    for poisoned in (False, True):
        Rt = good.clone()
        if poisoned:
            Rt[1, 0, 0] = float("nan")
        lg, sg = logits.clone().requires_grad_(True), sigma.clone().requires_grad_(True)
        rgb, ph, ph_mean = ops.plane_sweep_homography(src, tgt, lg, sg, distance, norm, Rt, K, inv_K, return_mean=True,
                                                      plane_uniform=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        (rgb[0].sum() * 1e-3 + ph[0].mean()).backward()      # a loss over the healthy image only
        torch.cuda.synchronize()
        res[poisoned] = (lg.grad[0].cpu(), sg.grad[0].cpu(), time.perf_counter() - t0)
    assert res[True][2] < 20 * res[False][2] + 0.05
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_contract_check_covers_the_homography_shortcuts():
    """The round-2 shortcuts of homography_warp are taken from what the reference's code guarantees (zero translation of
    a novel frame without COLMAP; the stereo pose is a pure x-translation and no normal has an x component):
    opt.pd_check_contract verifies each on the tensors."""
    import types
    import planedepth_amd
    from planedepth_amd.synthetic import intrinsics, small_pose
    B, N, H, W = 1, 4, 8, 32
    g = torch.Generator().manual_seed(9)
    dev = "cuda"
    K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1).to(dev)
    base = {"probability": torch.empty(B, N, H, W, device="meta"), "logits": torch.randn(B, N, H, W, generator=g).to(dev),
            "sigma": (0.1 + 0.8 * torch.rand(B, N, H, W, generator=g)).to(dev),
            "distance": (1.0 + torch.rand(B, N, generator=g)).to(dev), "norm": norm}
    imgs = {s: torch.rand(B, 3, H, W, generator=g).to(dev) for s in ("l", "r", -1)}
    inputs = {("color", s): v for s, v in imgs.items()}
    inputs.update(K=K, inv_K=inv_K)

    def run(side, Rt, **over):
        opt = types.SimpleNamespace(warp_type="homography_warp", match_aug=False, use_mixture_loss=True, automask=False,
                                    render_probability=False, xz_levels=0, yz_levels=0, use_colmap=False,
                                    pd_check_contract=True)
        ns = types.SimpleNamespace(opt=opt, target_sides=[side])
        out = dict(base, **over)
        out[("Rt", side)] = Rt.to(dev)
        planedepth_amd.pred_novel_images(ns, inputs, out)

    pose = small_pose(g, B, rot=0.02, trans=0.0)
    run(-1, pose)                                                     # zero translation: fine
    with pytest.raises(ValueError, match="has a translation"):
        run(-1, small_pose(g, B, rot=0.02, trans=0.05))
    run("r", small_pose(None, B, stereo=True))                        # the dataset's stereo extrinsic: fine
    with pytest.raises(ValueError, match="pure x-translation"):
        run("r", small_pose(g, B, rot=0.02, trans=0.05))
    tilted = norm.clone()
    tilted[:, 1] = torch.nn.functional.normalize(torch.tensor([0.3, 0.0, 1.0]), dim=0).to(dev)
    with pytest.raises(ValueError, match="x component"):
        run("r", small_pose(None, B, stereo=True), norm=tilted)


def test_on_device_grid_matches_the_reference_pipeline_to_one_ulp():
    """SURVEY §8f rank 4 / VERDICT r1 F4: inputs["grid"] generated on the device (pd_crop_grid) against the grids the
    reference's RandomResizeCrop / Resize produced (tests/golden/pipeline.npz); then the whole on-device minibatch has
    the reference's keys, shapes and conventions.

    Tolerance: one ulp of the coordinate (1.2e-7).  torch.linspace's CPU kernel evaluates `start + step * idx` per
    VECTOR (ATen RangeFactoriesKernel: Vectorized::arange(start + step * idx0, step), 8 lanes under AVX2, 16 under
    AVX-512), so the reference's own grid differs in the last bit between host CPUs; the kernel implements the scalar
    formula (start + step * i below the midpoint, end - step * (steps - 1 - i) above)."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import kitti_like_inputs, kitti_like_inputs_on_device
    z = np.load(os.path.join(GOLDEN, "pipeline.npz"))
    tags = sorted({k.split("/")[0] for k in z.files})
    for tag in tags:
        H, W = (int(v) for v in z[tag + "/hw"])
        params = torch.from_numpy(z[tag + "/params"])[None].repeat(3, 1).cuda()
        grid = ops.crop_grid(params, H, W)
        want = torch.from_numpy(z[tag + "/grid"])
        for b in range(3):
            assert float((grid[b].cpu() - want).abs().max()) <= 1.2e-7, (tag, float((grid[b].cpu() - want).abs().max()))
    dev = kitti_like_inputs_on_device(4, 24, 80, seed=7, novel_frame_ids=(-1, 1))
    cpu = kitti_like_inputs(4, 24, 80, seed=7, novel_frame_ids=(-1, 1))
    assert set(dev) == set(cpu)
    for k in cpu:
        assert dev[k].is_cuda and tuple(dev[k].shape) == tuple(cpu[k].shape), k
    for k in ("K", "inv_K", ("Rt", "l"), ("Rt", "r")):   # deterministic parts agree exactly with the CPU pipeline
        assert torch.equal(dev[k].cpu(), cpu[k]), k
    assert float((dev["grid"].cpu() - cpu["grid"]).abs().max()) <= 1.2e-7


def _f8_pose(B, seed, rot=0.02, device="cpu"):
    """Rt as Trainer.predict_poses leaves it for a novel frame without COLMAP (trainer.py:386-400): a rotation
    conjugated by a crop matrix, ZERO translation, Rt[3,3] = 0."""
    from planedepth_amd.synthetic import small_pose
    g = torch.Generator().manual_seed(seed)
    R = small_pose(g, B, rot=rot, trans=0.0)[:, :3, :3]
    Rc = torch.eye(3)[None].repeat(B, 1, 1)
    Rc[:, 0, 2] = torch.randn(B, generator=g) * 0.05
    Rc[:, 1, 2] = torch.randn(B, generator=g) * 0.05
    Rc[:, 2, 2] = 0.7 + 0.3 * torch.rand(B, generator=g)
    Rt = torch.zeros(B, 4, 4)
    Rt[:, :3, :3] = Rc @ R @ torch.inverse(Rc)
    return Rt.to(device)


_PAIR_CASES = [
    (2, 7, 24, 80, True, (0.02, 0.03), (1.0, 1.0), True),       # two pose-net frames behind a disp_warp view (accumulate)
    (2, 7, 24, 80, True, (0.02, 0.03), (1.0, 1.0), False),      # the pair starts the sum (plain stores)
    (1, 5, 40, 150, False, (0.05, 0.01), (1.0, 1.0), True),     # L1 loss: float scratch
    (1, 9, 64, 200, True, (0.5, 0.02), (1.0, 1.0), True),       # one view rotated by ~29 degrees: its box does not fit -> direct gathers
    (1, 4, 30, 90, True, (0.05, 0.02), (2.6, 1.0), False),      # one view 2.6x denser than the source: lists overflow -> follow-up
    (1, 4, 30, 90, True, (0.05, 0.02), (1.6, 1.55), True),      # 9-12 contributors: around the pair kernel's 10 slots
    (1, 49, 96, 320, True, (0.01, 0.012), (1.0, 1.0), True),
    (1, 3, 5, 7, True, (0.3, 0.1), (1.0, 1.0), False),          # smaller than one tile, odd sizes
    (1, 4, 33, 71, True, (0.04, 0.02), (1.0, 1.0), False),      # ragged tiles on both axes
    (2, 1, 20, 48, True, (0.03, 0.02), (1.0, 1.0), False)]      # a single plane


# (render_probability: the small cases cover the pair kernel's independence of the compositing mode)
@pytest.mark.parametrize("B,N,H,W,mix,rots,zooms,with_stereo,render",
                         [c + (False,) for c in _PAIR_CASES] + [c + (True,) for c in _PAIR_CASES if c[3] <= 100 and 2 <= c[1] <= 9])
def test_two_plane_uniform_views_gather_in_one_kernel(B, N, H, W, mix, rots, zooms, with_stereo, render, monkeypatch):
    """pd_uniform_fwd_pair / pd_uniform_bwd_pair (the two novel frames of a step: forwards in one launch, first passes in one
    launch, second passes in one kernel with one store per gradient element) against the same node with the views one
    after the other (PD_PAIR_FORWARD=0, PD_PAIR_GATHER=0: read-modify-write per view).  The forward's tensors: the same
    bits (the same arithmetic per view).  Gradients: same contributions added in the same order -> the same bits in
    g_logits / g_sigma, unless a list overflows the 12 register slots (then the follow-up kernels add those shares last)."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics
    g = torch.Generator().manual_seed(4200 + W + N)
    dev = "cuda"
    src = torch.rand(B, 3, H, W, generator=g).to(dev)
    tgts = [torch.rand(B, 3, H, W, generator=g).to(dev) for _ in range(3)]
    logits = torch.randn(B, N, H, W, generator=g).to(dev)
    sigma = (0.011 + 0.978 * torch.rand(B, N, H, W, generator=g)).to(dev)
    gws = [(torch.randn(B, 3, H, W, generator=g) * 0.1).to(dev) for _ in range(3)]
    distance = (0.5 + 5 * torch.rand(B, N, generator=g)).to(dev)
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1).to(dev)
    disp = (torch.rand(B, N, 1, 1, generator=g) * 20 + 0.5).to(dev)
    K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
    dists = (torch.rand(B, N - 1, H, W, generator=g) * 2.0).to(dev) if render else None
    res = {}
    for pair in (True, False):
        monkeypatch.setattr(ops, "PAIR_GATHER", pair)
        monkeypatch.setattr(ops, "PAIR_FORWARD", pair)
        lg, sg = logits.clone().requires_grad_(True), sigma.clone().requires_grad_(True)
        Rts = []
        calls = []
        if with_stereo:
            calls.append(ops.plane_sweep_disp(src, tgts[2], lg, sg if mix else None, disp.expand(-1, -1, H, W), None,
                                              target_side="r", use_mixture_loss=mix, return_mean=True, defer=True,
                                              render_probability=render, dists=dists))
        for v in range(2):
            Rt = _f8_pose(B, 31 + H + v, rots[v], dev)
            Rt[:, :2, :3] *= zooms[v]
            Rt.requires_grad_(True)
            Rts.append(Rt)
            calls.append(ops.plane_sweep_homography(src, tgts[v], lg, sg if mix else None, distance, norm, Rt, K, inv_K,
                                                    use_mixture_loss=mix, automask=True, return_mean=True,
                                                    plane_uniform=True, defer=True, render_probability=render, dists=dists))
        outs = ops.plane_sweep_multi(calls)
        loss = sum(o[2] * (i + 1.0) + (o[0] * gws[i]).sum() for i, o in enumerate(outs))
        loss.backward()
        res[pair] = dict(g_logits=lg.grad.cpu(), g_sigma=sg.grad.cpu() if mix else torch.zeros(1),
                         g_Rt0=Rts[0].grad.cpu(), g_Rt1=Rts[1].grad.cpu())
        for i, o in enumerate(outs):
            res[pair].update({"rgb_rec%d" % i: o[0].detach().cpu(), "ph_map%d" % i: o[1].detach().cpu(),
                              "ph_mean%d" % i: o[2].detach().cpu()})
    a, b = res[True], res[False]
    assert float(b["g_logits"].abs().max()) > 0
    exact = max(zooms) == 1.0
    for k in a:
        if k.startswith(("rgb_rec", "ph_map")) or (exact and k.startswith(("g_logits", "g_sigma"))):
            assert torch.equal(a[k], b[k]), (k, rel_err(a[k], b[k]))
        else:    # the pose gradients pass through torch's own reductions; ph_mean is a sum of atomics in arrival order
            assert rel_err(a[k], b[k]) < 2e-6, (k, rel_err(a[k], b[k]))


@pytest.mark.parametrize("B,N,H,W,mix,automask,rot,zoom", [
    (2, 7, 24, 80, True, True, 0.02, 1.0), (1, 9, 33, 70, True, False, 0.15, 1.0), (2, 5, 40, 150, False, True, 0.05, 1.0),
    (1, 3, 5, 7, True, False, 0.3, 1.0), (1, 63, 192, 640, True, True, 0.01, 1.0),
    (1, 4, 30, 90, True, False, 0.05, 2.6),     # target 2.6x denser than the source: ~27 contributors per source pixel
    (1, 4, 30, 90, True, False, 0.05, 0.45)])   # the other way round: most source pixels get none
@pytest.mark.parametrize("bwd", ["staged", "direct"])
def test_plane_uniform_homography_kernels_equal_the_general_ones(B, N, H, W, mix, automask, rot, zoom, bwd, monkeypatch):
    """PD_HOMO_UNIFORM (pd_plane_sweep_uniform.hip: geometry once per pixel, two-pass atomic-free backward) against the
    general kernels on poses shaped like predict_poses' output (zero translation): one homography per image, planes with
    two different normals (the facing test still differs per plane), rotations up to 17 degrees, ragged sizes, and
    zooms that push the per-source-pixel gather list past its 8 register slots (the follow-up re-scan kernel)."""
    from planedepth_amd import ops
    from planedepth_amd.synthetic import intrinsics
    # backward variants: pass 2 with the scratch staged through LDS (default), the direct-gather pass 2
    # (PD_IMPL_UNIFORM_DIRECT); tests/experiments runs the same body with the one-kernel form (PD_UNI_FUSED)
    from planedepth_amd import _capi as C
    if bwd != "fused":
        monkeypatch.delenv("PD_UNI_FUSED", raising=False)
    monkeypatch.setattr(ops, "SWEEP_IMPL", C.PD_IMPL_UNIFORM_DIRECT if bwd == "direct" else C.PD_IMPL_AUTO)
    g = torch.Generator().manual_seed(900 + W + N)
    dev = "cuda"
    src, tgt = torch.rand(B, 3, H, W, generator=g).to(dev), torch.rand(B, 3, H, W, generator=g).to(dev)
    logits = torch.randn(B, N, H, W, generator=g).to(dev)
    sigma = (0.011 + 0.978 * torch.rand(B, N, H, W, generator=g)).to(dev)
    gw = (torch.randn(B, 3, H, W, generator=g) * 0.1).to(dev)
    distance = (0.5 + 5 * torch.rand(B, N, generator=g)).to(dev)
    norm = torch.tensor([0.0, 0.0, 1.0])[None, None].repeat(B, N, 1)
    norm[:, N // 2:] = torch.nn.functional.normalize(torch.tensor([0.0, 1.0, 0.07]), dim=0)   # "xz planes"
    norm = norm.to(dev)
    K, inv_K = (t.to(dev) for t in intrinsics(B, H, W))
    res = {}
    for uniform in (True, False):
        lg, sg = logits.clone().requires_grad_(True), sigma.clone().requires_grad_(True)
        Rt = _f8_pose(B, 31 + H, rot, dev)
        Rt[:, :2, :3] *= zoom              # not a rotation any more: the kernels take whatever the 3x3 block holds
        Rt.requires_grad_(True)
        dd = distance.clone().requires_grad_(True)
        rgb, ph, ph_mean = ops.plane_sweep_homography(src, tgt, lg, sg if mix else None, dd, norm, Rt, K, inv_K,
                                                      use_mixture_loss=mix, automask=automask, return_mean=True,
                                                      plane_uniform=uniform)
        (ph_mean * 2.0 + (rgb * gw).sum()).backward()
        res[uniform] = dict(rgb=rgb.detach().cpu(), ph=ph.detach().cpu(), g_logits=lg.grad.cpu(), g_Rt=Rt.grad.cpu(),
                            g_sigma=sg.grad.cpu() if mix else torch.zeros(1),
                            g_dist=(dd.grad if dd.grad is not None else torch.zeros_like(dd)).cpu())
    u, gen = res[True], res[False]
    assert float(gen["g_logits"].abs().max()) > 0
    assert torch.equal(u["rgb"], gen["rgb"]) or rel_err(u["rgb"], gen["rgb"]) < 2e-6
    assert rel_err(u["ph"], gen["ph"]) < 2e-6
    assert rel_err(u["g_logits"], gen["g_logits"]) < 3e-6, rel_err(u["g_logits"], gen["g_logits"])
    if mix:
        assert rel_err(u["g_sigma"], gen["g_sigma"]) < 3e-6, rel_err(u["g_sigma"], gen["g_sigma"])
    # rotation block AND translation column (the weighted plane sums of PD_HOMO_UNIFORM); sums in a different order
    assert rel_err(u["g_Rt"][:, :3, :3], gen["g_Rt"][:, :3, :3]) < 2e-4, rel_err(u["g_Rt"][:, :3, :3], gen["g_Rt"][:, :3, :3])
    assert rel_err(u["g_Rt"][:, :3, 3], gen["g_Rt"][:, :3, 3]) < 2e-4, rel_err(u["g_Rt"][:, :3, 3], gen["g_Rt"][:, :3, 3])
    assert float(u["g_dist"].abs().max()) == 0.0 and float(gen["g_dist"].abs().max()) < 1e-6   # t = 0: exactly no gradient
    if (N, H, W) == (63, 192, 640) and bwd == "staged":
        # at the benchmark size the plane-uniform kernels are also held to the ORACLE directly (not only to the
        # general kernels): the fp32 oracle — the reference's arithmetic, facing test and all — on the matrices the product itself
        # formed (its fp64-rounded-once H_t2s and rotated normals), 1e-4.  (An fp64 oracle disagrees on which side of the horizon line a handful of
        # pixels fall: ph_map 2e-3 there.)
        from oracle import planedepth_oracle as orc
        with torch.no_grad():
            Rt0 = _f8_pose(B, 31 + H, rot, "cpu")
            Rt0[:, :2, :3] *= zoom
            Hm, Rn = ops.homography_matrices_fused(distance, norm, Rt0.to(dev), K, inv_K)
        Hm, Rn = Hm.reshape(B * N, 3, 3).cpu(), Rn.reshape(B * N, 3).cpu()
        lgo, sgo = logits.cpu().requires_grad_(True), sigma.cpu().requires_grad_(True)
        r = orc.warp_and_loss(src.cpu(), tgt.cpu(), lgo, sgo if mix else None, warp_type="homography_warp",
                              distance=distance.cpu(), norm=norm.cpu(), T=Rt0, K=K.cpu(),
                              inv_K=inv_K.cpu(), use_mixture_loss=mix, automask=automask, H_t2s=Hm, Rn=Rn)
        (r["ph_loss"] * 2.0 + (r["rgb_rec"] * gw.cpu()).sum()).backward()
        assert rel_err(u["rgb"], r["rgb_rec"].detach().float()) < TOL
        assert rel_err(u["ph"], r["ph_map"].detach().float()) < TOL
        assert rel_err(u["g_logits"], lgo.grad.float()) < TOL, rel_err(u["g_logits"], lgo.grad.float())
        # g_sigma: the clamp's gate (trainer.py:597, inclusive bounds) is a knife edge where a border sample's sigma — a source
        # sigma times a partial bilinear weight — lands within an ulp of 0.01: one evaluation passes the (1 / sigma^2-sized,
        # i.e. near-maximal) gradient on, the other closes the gate.  7.7 M samples here: a handful of such elements are
        # allowed, every other element is held to 1e-4 of the tensor's range
        gs, ws = u["g_sigma"].double(), sgo.grad.double()
        beyond = int(((gs - ws).abs() > TOL * ws.abs().max()).sum())
        assert beyond <= 4 and rel_err(u["g_sigma"], sgo.grad.float()) < 5e-3, (beyond, rel_err(u["g_sigma"], sgo.grad.float()))
