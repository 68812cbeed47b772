"""Error budget of homography_warp on the GPU box: the product against the reference-captured trainer fixtures, and —
at 192x640x63 with pinned matrices — product vs fp32 oracle vs fp64 oracle (three-way).  Prints one line per tensor."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import TRAINER_MONO, load_trainer_fixture, rel_err  # noqa: E402
from gpu_cases import run_product_trainer  # noqa: E402
from oracle import planedepth_oracle as orc  # noqa: E402
from planedepth_amd import _capi as C  # noqa: E402
from planedepth_amd import ops  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def elementwise(a, b, floor_frac=1e-3):
    """max |a-b| / max(|b|, floor) element by element, floor = floor_frac * max|b|."""
    a, b = a.double(), b.double()
    fl = float(b.abs().max()) * floor_frac
    return float(((a - b).abs() / b.abs().clamp_min(fl)).max())


for tag in TRAINER_MONO:
    z, meta = load_trainer_fixture(tag)
    for impl in (None, C.PD_IMPL_TILE):
        got = run_product_trainer(z, meta, impl=impl)
        print(tag, "impl", impl, {k: "%.1e" % rel_err(v if k != "g_disp_layered" else v.sum(-1),
                                                      z[k] if k != "g_disp_layered" else z[k].sum(-1))
                                  for k, v in got.items() if float(z[k].abs().max()) > 0})

c = T._mono_fullsize_case()
B, N, H, W = c["logits"].shape
ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
H64, Rn64 = orc.homography_matrices(c["distance"].double(), c["norm"].double(), ex(c["Rt"].double()),
                                    ex(c["K"].double()), ex(c["inv_K"].double()))
Hm = H64.float()
for mix, automask in ((True, True), (False, False)):
    outs = {}
    for dt in (torch.float32, torch.float64):
        t0 = time.time()
        cc = {k: v.to(dt) for k, v in c.items()}
        lg, sg, Hl = cc["logits"].clone().requires_grad_(True), cc["sigma"].clone().requires_grad_(True), Hm.detach().clone().to(dt).requires_grad_(True)
        r = orc.warp_and_loss(cc["color_l"], cc["color_r"], lg, sg if mix else None, warp_type="homography_warp",
                              distance=cc["distance"], norm=cc["norm"], T=cc["Rt"], K=cc["K"], inv_K=cc["inv_K"],
                              use_mixture_loss=mix, automask=automask, H_t2s=Hl)
        (r["ph_loss"] + (r["rgb_rec"] * cc["gw"]).sum()).backward()
        outs[dt] = dict(rgb_rec=r["rgb_rec"].detach(), ph_map=r["ph_map"].detach(), g_logits=lg.grad, g_H=Hl.grad)
        if mix:
            outs[dt]["g_sigma"] = sg.grad
        print("oracle", dt, "%.1fs" % (time.time() - t0))
    dev = "cuda"
    lgd, sgd, Hd = (c["logits"].to(dev).requires_grad_(True), c["sigma"].to(dev).requires_grad_(True), Hm.detach().clone().to(dev).requires_grad_(True))
    flags = (C.PD_MIXTURE if mix else 0) | (C.PD_AUTOMASK if automask else 0)
    rgb, ph, ph_mean = ops._PlaneSweep.apply(c["color_l"].to(dev), c["color_r"].to(dev), lgd, sgd if mix else None, Hd,
                                             Rn64.float().reshape(B * N, 3).to(dev), c["inv_K"][:, :3, :3].to(dev), None,
                                             None, C.PD_WARP_HOMOGRAPHY, flags, 0.0)
    (ph_mean + (rgb * c["gw"].to(dev)).sum()).backward()
    got = dict(rgb_rec=rgb.detach().cpu(), ph_map=ph.detach().cpu(), g_logits=lgd.grad.cpu(), g_H=Hd.grad.cpu())
    if mix:
        got["g_sigma"] = sgd.grad.cpu()
    o32, o64 = outs[torch.float32], outs[torch.float64]
    for k in got:
        print("mix=%d %-9s hip-o32 %.2e  hip-o64 %.2e  o32-o64 %.2e   | elementwise(floor 1e-3): hip-o32 %.2e hip-o64 %.2e o32-o64 %.2e"
              % (mix, k, rel_err(got[k], o32[k]), rel_err(got[k], o64[k].float()), rel_err(o32[k], o64[k].float()),
                 elementwise(got[k], o32[k]), elementwise(got[k], o64[k]), elementwise(o32[k], o64[k])))
