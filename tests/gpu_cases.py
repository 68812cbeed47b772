"""Drive the PRODUCT path (planedepth_amd, HIP kernels via the C ABI) on a fixture / synthetic case."""
import types

import torch

import planedepth_amd
from planedepth_amd import ops


def run_product(case, run, device="cuda", force_dense=False, through_trainer=True, opt_extra=None):
    """Same contract as cases.run_oracle / make_golden.run_reference, but on the GPU through the product API."""
    c = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in case.items()}
    B, N, H, W = c["logits"].shape
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    logits, sigma, disp_pp, Rt = leaf(c["logits"]), leaf(c["sigma"]), leaf(c["disp_pp"]), leaf(c["Rt"])
    if case["dense_disp"] or force_dense:
        disp_layered = disp_pp.expand(-1, -1, H, W) * c["row_gain"]
    else:
        disp_layered = disp_pp.expand(-1, -1, H, W)  # a view, as the decoder makes it for xy planes
    distance = 0.1 * 0.58 * W / disp_pp[:, :, 0, 0]
    norm = torch.tensor([0.0, 0.0, 1.0], device=device)[None, None].expand(B, N, -1)
    side = run.get("target_side", "r")
    mix = run.get("use_mixture_loss", True)
    warp = run.get("warp_type", "disp_warp")
    inputs = {("color", "l"): c["color_l"], "K": c["K"], "inv_K": c["inv_K"]}
    if side != "l":
        inputs[("color", side)] = c["color_r"]
    outputs = {"probability": torch.empty(B, N, H, W, device="meta"), "logits": logits, "sigma": sigma,
               "disp_layered": disp_layered, "padding_mask": c["padding_mask"], "distance": distance, "norm": norm,
               ("Rt", side): Rt,
               "disp": (torch.softmax(logits.detach(), 1) * disp_layered.detach()).sum(1, True)}  # as make_golden.py
    if "mask_novel" in c:
        outputs["mask_novel"] = c["mask_novel"]
    dists = None
    if run.get("render_probability", False):
        dists = c["dists"].clone().requires_grad_(True)
        outputs["dists"] = dists
    opt = types.SimpleNamespace(warp_type=warp, match_aug=False, use_mixture_loss=mix, automask=run.get("automask", False),
                                render_probability=run.get("render_probability", False), alpha_pc=0.0, alpha_self=0.0,
                                self_distillation=0.0, gamma_smooth=2.0, alpha_smooth=0.04, use_ssim=True,
                                materialize_layers=True, **(opt_extra or {}))
    ns = types.SimpleNamespace(opt=opt, target_sides=[side],
                               perceptual_loss=lambda *a, **k: torch.zeros((), device=device))
    planedepth_amd.pred_novel_images(ns, inputs, outputs)
    losses = planedepth_amd.compute_losses(ns, inputs, outputs)
    rgb_rec = outputs[("rgb_rec", side)]
    # (run["ph_scale"]: weight of the photometric head in the test objective — a full batch against its data-parallel shards)
    (losses["loss/ph_loss"] * run.get("ph_scale", 1.0) + (rgb_rec * c["g_rgb_rec"]).sum()).backward()
    z = torch.zeros_like
    res = dict(rgb_rec=rgb_rec, ph_loss=losses["loss/ph_loss"], ph_map=outputs[("ph_map", side)],
               smooth_loss=losses["loss/smooth_loss"], total_loss=losses["loss/total_loss"],
               g_logits=logits.grad, g_sigma=sigma.grad if sigma.grad is not None else z(sigma),
               g_disp_pp=disp_pp.grad if disp_pp.grad is not None else z(disp_pp),
               g_Rt=Rt.grad if Rt.grad is not None else z(Rt))
    if dists is not None:
        res["g_dists"] = dists.grad if dists.grad is not None else z(dists)
    for k in ("rgb_rec_layered", "logit_rec", "probability_rec", "sigma_rec", "pi_rec"):
        if (k, side) in outputs:
            res[k] = outputs[(k, side)]
    return {k: v.detach().cpu() for k, v in res.items()}


def make_stub_trainer(opt, target_sides, device="cuda"):
    """A reference-shaped Trainer class with nothing but what the hot path reads, adopted through ``patch_trainer``
    exactly as INTEGRATION.md tells a maintainer to do with the real one."""
    class StubTrainer:
        def __init__(self):
            self.opt = opt
            self.target_sides = target_sides
            self.device = torch.device(device)

        def perceptual_loss(self, *a, **k):
            return torch.zeros((), device=device)

    planedepth_amd.patch_trainer(StubTrainer)
    return StubTrainer()


def run_product_trainer(z, meta, device="cuda", impl=None, stereo_constant=False, opt_extra=None):
    """tests/golden/trainer_mono.npz through the patched Trainer methods (pred_novel_images + compute_losses over every
    target side).  Same keys as cases.run_oracle_trainer."""
    from cases import side_key
    c = {k: v.to(device) for k, v in z.items()}
    sides = [side_key(s) for s in meta["target_sides"]]
    mix, homo = meta["use_mixture_loss"], meta["warp_type"] == "homography_warp"
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    logits = leaf(c["logits"])
    sigma = leaf(c["sigma"]) if mix else None
    distance, disp_layered = leaf(c["distance"]), leaf(c["disp_layered"])
    B, N, H, W = logits.shape
    inputs = {("color", "l"): c["color_l"], "K": c["K"], "inv_K": c["inv_K"], "grid": c["grid"]}
    outputs = {"probability": torch.empty(B, N, H, W, device="meta"), "logits": logits, "disp_layered": disp_layered,
               "padding_mask": c["padding_mask"], "distance": distance, "norm": c["norm"], "disp": c["disp"]}
    if mix:
        outputs["sigma"] = sigma
    Rts = {}
    for s in sides:
        inputs[("color", s)] = c["color_%s" % s]
        # stereo_constant: inputs[("Rt", "r")] as the dataset hands it over (a constant; trainer.py:364), which is what
        # lets the stereo side run as per-row shifts.  Otherwise every pose is a leaf so that g_Rt can be compared.
        Rts[s] = outputs[("Rt", s)] = c["Rt_%s" % s] if (stereo_constant and s == "r") else leaf(c["Rt_%s" % s])
    opt = types.SimpleNamespace(warp_type=meta["warp_type"], match_aug=False, use_mixture_loss=mix,
                                automask=meta["automask"], render_probability=False, alpha_pc=0.0, alpha_self=0.0,
                                self_distillation=0.0, gamma_smooth=2.0, alpha_smooth=0.04, use_ssim=True,
                                xz_levels=meta["xz_levels"], yz_levels=0, novel_frame_ids=[s for s in sides if s != "r"],
                                **(opt_extra or {}))
    trainer = make_stub_trainer(opt, sides, device)
    if impl is not None:
        ops.SWEEP_IMPL = impl
    try:
        trainer.pred_novel_images(inputs, outputs)
        losses = trainer.compute_losses(inputs, outputs)
        obj = losses["loss/total_loss"] + sum((outputs[("rgb_rec", s)] * c["gw_%s" % s]).sum() for s in sides)
        obj.backward()
    finally:
        ops.SWEEP_IMPL = 0
    zz = torch.zeros_like
    res = {"rgb_rec_%s" % s: outputs[("rgb_rec", s)] for s in sides}
    res.update(ph_loss=losses["loss/ph_loss"], total_loss=losses["loss/total_loss"],
               smooth_loss=losses["loss/smooth_loss"], g_logits=logits.grad)
    if mix:
        res["g_sigma"] = sigma.grad
    if homo:
        res["g_distance"] = distance.grad if distance.grad is not None else zz(distance)
    else:
        res["g_disp_layered"] = disp_layered.grad
    for s in sides:
        res["g_Rt_%s" % s] = Rts[s].grad if Rts[s].grad is not None else zz(Rts[s])
    return {k: v.detach().cpu() for k, v in res.items()}
