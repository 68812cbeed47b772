"""Shared helpers: load golden fixtures and drive the ORACLE on them (test infrastructure)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import planedepth_oracle as orc

SMALL = ["disp_mix_r", "disp_mix_l", "disp_mix_automask", "disp_l1", "disp_l1_automask", "disp_mix_xz",
         "disp_mix_oob", "disp_mix_integer_d", "disp_mix_masknovel", "disp_l1_masknovel",
         "homo_mix_stereo", "homo_mix_pose", "homo_l1_pose", "disp_mix_render"]


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    inp = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    out = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out_")}
    return inp, out, meta["run"]


def run_oracle(case, run, dtype=torch.float32, sampler=orc.bilinear_sample, timing=None, H_t2s=None):
    """Mirror of make_golden.run_reference, but through the oracle.  Returns the same keys.  `timing` (a dict) receives
    `fwd_s` (inputs -> loss, autograd recording as in training) and `fwd_bwd_s` (.. -> gradients): bench.py's cpu_baseline.
    `H_t2s` [B*N,3,3] pins homography_warp's matrices (the reference-captured ones of the fixtures: no gradient to distance / Rt
    then)."""
    import time
    t_start = time.perf_counter()
    c = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in case.items()}
    B, N, H, W = c["logits"].shape
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    logits, sigma, disp_pp, Rt = leaf(c["logits"]), leaf(c["sigma"]), leaf(c["disp_pp"]), leaf(c["Rt"])
    disp_layered = disp_pp.expand(-1, -1, H, W) * c["row_gain"]
    distance = 0.1 * 0.58 * W / disp_pp[:, :, 0, 0]
    norm = torch.tensor([0.0, 0.0, 1.0], dtype=dtype)[None, None].expand(B, N, -1)
    mix = run.get("use_mixture_loss", True)
    dists = c["dists"].clone().requires_grad_(True) if run.get("render_probability", False) else None
    # the reference reads the target from inputs[("color", side)]; for side "l" that IS the source image
    tgt = c["color_l"] if run.get("target_side", "r") == "l" else c["color_r"]
    r = orc.warp_and_loss(c["color_l"], tgt, logits, sigma if mix else None,
                          warp_type=run.get("warp_type", "disp_warp"), target_side=run.get("target_side", "r"),
                          disp_layered=disp_layered, padding_mask=c["padding_mask"], distance=distance, norm=norm,
                          T=Rt, K=c["K"], inv_K=c["inv_K"], use_mixture_loss=mix, automask=run.get("automask", False),
                          mask_novel=c.get("mask_novel"), render_probability=run.get("render_probability", False),
                          dists=dists, sampler=sampler, H_t2s=None if H_t2s is None else H_t2s.to(dtype))
    objective = r["ph_loss"] + (r["rgb_rec"] * c["g_rgb_rec"]).sum()
    if timing is not None:
        timing["fwd_s"] = time.perf_counter() - t_start
    objective.backward()
    if timing is not None:
        timing["fwd_bwd_s"] = time.perf_counter() - t_start
    z = torch.zeros_like
    res = dict(rgb_rec=r["rgb_rec"], ph_loss=r["ph_loss"], ph_map=r["ph_map"],
               rgb_rec_layered=r["sweep"]["rgb_rec_layered"], logit_rec=r["sweep"]["logit_rec"],
               probability_rec=r["sweep"]["probability_rec"],
               g_logits=logits.grad, g_sigma=sigma.grad if sigma.grad is not None else z(sigma),
               g_disp_pp=disp_pp.grad if disp_pp.grad is not None else z(disp_pp),
               g_Rt=Rt.grad if Rt.grad is not None else z(Rt))
    if dists is not None:
        res["g_dists"] = dists.grad if dists.grad is not None else z(dists)
    if run.get("warp_type", "disp_warp") == "homography_warp":   # the matrices this evaluation formed itself (layers.py:206-219)
        ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
        res["H_t2s"] = orc.homography_matrices(distance, norm, ex(Rt), ex(c["K"]), ex(c["inv_K"]))[0]
    if mix:
        res["sigma_rec"] = r["sweep"]["sigma_rec"]
        res["pi_rec"] = r["sweep"]["pi_rec"]
    return {k: v.detach() for k, v in res.items()}


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny): the normalised max error used for every fp32 parity check."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# Absolute caps against the fp64 evaluation of the reference's formulas, next to the relative three-way bound: where the
# reference's own fp32 run is 1e-2 away from fp64 (torch.inverse at cond(H) ~ 1e4) the relative bound alone would let a
# kernel error of several percent through.  The product forms its matrices in fp64 and rounds once, so only the per-pixel
# fp32 chain separates it from the fp64 oracle: measured <= 2.4e-4 (forward tensors) / 6.4e-4 (gradients) over every
# homography test of the suite (profiles/r03_parity.md); the caps leave a factor of ~2-3.
FWD_CAP, GRAD_CAP = 5e-4, 2e-3


def cap_for(key):
    return GRAD_CAP if key.startswith("g_") else FWD_CAP


def three_way(got, ref32, exact, factor=2.0, tol=1e-4, cap=None):
    """The parity bar where two fp32 evaluations of the reference's formulas legitimately differ by more than 1e-4
    (fp32 matrix inverses, 1/sigma^3-type amplification, bilinear derivatives through fp32 coordinates): the product must
    be as close to the fp64 evaluation as the reference's own fp32 arithmetic is — err(got, fp64) <= factor *
    err(ref32, fp64) + tol — and, where ``cap`` is given, no further from it than that absolute amount.
    Returns (ok, err_got, err_ref)."""
    e_got, e_ref = rel_err(got, exact.float()), rel_err(ref32, exact.float())
    return e_got <= factor * e_ref + tol and (cap is None or e_got <= cap), e_got, e_ref


def elementwise_report(a, b, rtol=1e-4, floor=1e-4):
    """Element-wise companion of rel_err: the share of elements with |a-b| > rtol*|b| + floor*max|b| and the worst
    element's error in units of that allowance."""
    a, b = a.double().flatten(), b.double().flatten()
    allow = rtol * b.abs() + floor * b.abs().max().clamp_min(1e-30)
    r = (a - b).abs() / allow
    return dict(max_norm_err=rel_err(a, b), frac_beyond=float((r > 1).double().mean()), worst_over_allowance=float(r.max()))


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] as the trainer runs it: all target sides, decoder-made xz planes (tests/golden/trainer_mono.npz)
# ---------------------------------------------------------------------------------------------------------------------
TRAINER_MONO = ["homo3", "homo_nostereo_l1", "disp_xz"]


def load_trainer_fixture(tag):
    z = np.load(os.path.join(GOLDEN, "trainer_mono.npz"))
    meta = json.loads(bytes(z[tag + "/meta"]).decode())
    t = {k.split("/", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/") and not k.endswith("/meta")}
    return t, meta


def side_key(s):
    return s if isinstance(s, str) else int(s)


def run_oracle_trainer(z, meta, dtype=torch.float32, pin_matrices=False):
    """pred_novel_images + compute_losses over every target side, restated with the oracle's building blocks
    (trainer.py:532, 717, 765-771: the loss dict is divided by len(target_sides) BEFORE the smoothness term)."""
    c = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in z.items()}
    sides = [side_key(s) for s in meta["target_sides"]]
    mix, homo = meta["use_mixture_loss"], meta["warp_type"] == "homography_warp"
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    logits = leaf(c["logits"])
    sigma = leaf(c["sigma"]) if mix else None
    distance, disp_layered = leaf(c["distance"]), leaf(c["disp_layered"])
    Rts = {s: leaf(c["Rt_%s" % s]) for s in sides}
    B, N, H, W = logits.shape
    res, total, ph_total = {}, 0.0, 0.0
    for s in sides:
        r = orc.warp_and_loss(c["color_l"], c["color_%s" % s], logits, sigma,
                              warp_type=meta["warp_type"], target_side=s, disp_layered=disp_layered,
                              padding_mask=c["padding_mask"], distance=distance, norm=c["norm"], T=Rts[s], K=c["K"],
                              inv_K=c["inv_K"], use_mixture_loss=mix, automask=meta["automask"],
                              H_t2s=c["H_t2s_%s" % s] if (pin_matrices and homo) else None)   # the reference's own matrices
        res["rgb_rec_%s" % s] = r["rgb_rec"]
        ph_total = ph_total + r["ph_loss"]
        total = total + r["ph_loss"] + (r["rgb_rec"] * c["gw_%s" % s]).sum() * len(sides)  # the test objective adds it undivided
    ph_loss = ph_total / len(sides)
    x0 = int(0.2 * W)
    smooth = orc.smooth_loss_disp(c["disp"][..., x0:], c["color_l"][..., x0:], 2.0)
    total_loss = ph_loss + 0.04 * smooth
    obj = total / len(sides) + 0.04 * smooth
    obj.backward()
    zz = torch.zeros_like
    res.update(ph_loss=ph_loss, total_loss=total_loss, smooth_loss=smooth, g_logits=logits.grad)
    if mix:
        res["g_sigma"] = sigma.grad
    if homo:
        res["g_distance"] = distance.grad if distance.grad is not None else zz(distance)
    else:
        res["g_disp_layered"] = disp_layered.grad
    for s in sides:
        res["g_Rt_%s" % s] = Rts[s].grad if Rts[s].grad is not None else zz(Rts[s])
    return {k: v.detach() for k, v in res.items()}
