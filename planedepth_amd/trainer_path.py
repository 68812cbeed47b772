"""The reference Trainer's hot-path methods, re-implemented on the fused HIP operators.

``pred_novel_images``, ``compute_reprojection_loss`` and ``compute_losses`` keep the reference's names, argument
lists and the ``inputs`` / ``outputs`` / ``losses`` dict contracts (trainer.py:523-603, 687-699, 701-773;
SURVEY.md row B1), so a reference ``Trainer`` can adopt them verbatim:

    import planedepth_amd
    planedepth_amd.patch_trainer(trainer_module.Trainer)

``self`` only needs ``opt`` (warp_type, match_aug, use_mixture_loss, automask, alpha_pc, alpha_self,
self_distillation, alpha_smooth, gamma_smooth, use_ssim), ``target_sides`` and ``perceptual_loss``.

What differs from the reference (all documented in DESIGN.md):
  * the [B,N,*,H,W] per-plane tensors (``rgb_rec_layered``, ``logit_rec``, ``probability_rec``, ``sigma_rec``,
    ``pi_rec``) are NOT materialised: nothing live consumes them once the loss is fused.  ``outputs[("sweep", side)]``
    holds a handle whose ``.layers()`` produces them on demand; ``opt.materialize_layers=True`` stores them eagerly
    under the reference's keys.
  * the per-pixel photometric loss is produced by the same kernel that warps (``outputs[("ph_map", side)]``) and
    ``compute_losses`` only averages it.
  * ``depth_warp`` is dead code in the reference (its branch raises UnboundLocalError): ``pred_novel_images``
    raises for it as well, naming the reason; the modules it would use (BackprojectDepth / Project3D) are available
    from ``planedepth_amd.layers`` and are pinned against the reference by direct module calls.
"""
import os

import torch

from . import ops


class SweepHandle:
    """Lazy access to the per-plane tensors of one target view (reference keys of trainer.py:582-602)."""

    def __init__(self, **kw):
        self._kw = kw
        self._cache = None

    def layers(self):
        if self._cache is None:
            kw = dict(self._kw)
            if callable(kw.get("homography")):   # the 3x3 algebra is only redone when somebody asks for the layers
                kw["homography"] = kw["homography"]()
            self._cache = ops.plane_sweep_layers(**kw)
        return self._cache

    def __getitem__(self, key):
        return self.layers()[key]


def _color_name(opt):
    return "color_aug" if getattr(opt, "match_aug", False) else "color"


def pred_novel_images(self, inputs, outputs):
    """Generate the warped (reprojected) colour images for a minibatch (reference trainer.py:523-603).

    Writes ``outputs[("rgb_rec", side)]`` (+ ``("ph_map", side)`` and ``("sweep", side)``) for every target side.
    """
    opt = self.opt
    B, N, H, W = outputs["probability"].shape
    source_side = "l"
    cname = _color_name(opt)
    src = inputs[(cname, source_side)]
    mix = bool(opt.use_mixture_loss)
    automask = bool(getattr(opt, "automask", False))
    render = bool(getattr(opt, "render_probability", False))
    dists = outputs["dists"] if render else None  # trainer.py:585-587: the decoder's inter-plane distances
    # With xy planes only the decoder's padding mask is torch.ones_like(disp_layered) (networks/depth_decoder.py:157;
    # zeros only enter with xz/yz planes, :163-207, :224-247).  Reading N*H*W ones is 1/3 of the forward's HBM traffic,
    # so when the options say there are no xz/yz planes the mask is not read at all.
    padding_mask = outputs.get("padding_mask")
    if getattr(opt, "xz_levels", None) == 0 and getattr(opt, "yz_levels", None) == 0:
        padding_mask = None
    # xy and xz planes have disparities that are constant along x (depth_decoder.py:153-181); yz planes do not (:221-236)
    row_uniform = getattr(opt, "yz_levels", None) == 0
    # The shortcuts (no mask read, row-uniform disparities, one homography per image, the stereo view as row shifts) are
    # taken from the OPTIONS (what the reference's three networks guarantee), not from the tensors: a custom decoder or
    # pose source whose outputs disagree with opt would get silently wrong warps.  They are therefore verified on the
    # data ONCE per trainer object — on its first call (a few reductions and one host sync), the verdict cached on the
    # object; opt.pd_check_contract = True (or PD_CHECK_CONTRACT=1) checks every call, = False never.
    check = getattr(opt, "pd_check_contract", None)
    if os.environ.get("PD_CHECK_CONTRACT"):
        check = True
    if check is None:
        check = not getattr(self, "_pd_contract_checked", False)
    if check:
        pm, dl = outputs.get("padding_mask"), outputs.get("disp_layered")   # (homography_warp does not read disp_layered)
        if padding_mask is None and pm is not None and not bool((pm == 1).all()):
            raise ValueError("opt.xz_levels == opt.yz_levels == 0 promises an all-ones padding_mask, but it has zeros")
        if row_uniform and dl is not None and dl.dim() == 4 and dl.shape[-1] > 1 and not bool((dl == dl[..., :1]).all()):
            raise ValueError("opt.yz_levels == 0 promises disparities that are constant along x, but disp_layered is not")
        if row_uniform and pm is not None and pm.dim() == 4 and pm.shape[-1] > 1 and not bool((pm == pm[..., :1]).all()):
            raise ValueError("opt.yz_levels == 0 promises a padding_mask that is constant along x, but it is not")
    # Several target views (trainer.py:532; mono training: ["r", -1, 1]) sweep the same logits / sigma: they are issued as
    # ONE autograd node whose backward kernels add their gradients in place (ops._MultiPlaneSweep) instead of one node
    # per view with [B,N,H,W]-sized adds in between.  opt.pd_fuse_sides = False keeps one node per view.
    fuse_sides = len(self.target_sides) > 1 and getattr(opt, "pd_fuse_sides", True)
    try:
        self._pd_contract_checked = True     # (set before the per-view checks run: a failing check raises anyway)
    except AttributeError:
        pass
    calls, handles = [], []
    for target_side in self.target_sides:
        tgt = inputs[(cname, target_side)]
        sigma = outputs["sigma"] if mix else None
        if opt.warp_type == "disp_warp":
            call = ops.plane_sweep_disp(src, tgt, outputs["logits"], sigma, outputs["disp_layered"],
                                        padding_mask, target_side=target_side,
                                        use_mixture_loss=mix, automask=automask,
                                        render_probability=render, dists=dists,
                                        row_uniform=row_uniform, return_mean=True, defer=True)
            handle = SweepHandle(src=src, logits=outputs["logits"], sigma=sigma, disp_layered=outputs["disp_layered"],
                                 padding_mask=padding_mask, target_side=target_side, use_mixture_loss=mix,
                                 render_probability=render, dists=dists)
        elif opt.warp_type == "homography_warp":
            T = outputs[("Rt", target_side)]
            # Novel frames without COLMAP: predict_poses leaves the translation at zero (trainer.py:386-400, only
            # `if self.opt.use_colmap` writes it), so one homography serves all planes of an image.
            uniform = (target_side != "r" and not getattr(opt, "use_colmap", False)
                       and getattr(opt, "pd_uniform_homography", True))
            if uniform and check:
                if not bool((T[:, :3, 3] == 0).all()):
                    raise ValueError("outputs[('Rt', %r)] has a translation although opt.use_colmap is off" % (target_side,))
            # The stereo side: inputs[("Rt", "r")] is the dataset's pure x-translation (mono_dataset.py:203-211, copied
            # to outputs at trainer.py:364) and xy / xz planes have no x component in their normals, so the warp is a
            # per-row horizontal shift and runs on the row-shift kernels.
            stereo_rows = (target_side in ("l", "r") and row_uniform
                           and getattr(opt, "pd_stereo_rows", True))
            if stereo_rows and check:
                eye = torch.eye(3, device=T.device, dtype=T.dtype)
                if not (bool((T[:, :3, :3] == eye).all()) and bool((T[:, 1:3, 3] == 0).all())
                        and bool((outputs["norm"][..., 0] == 0).all())):
                    raise ValueError("outputs[('Rt', %r)] is not a pure x-translation, or a plane normal has an x "
                                     "component although opt.yz_levels == 0" % (target_side,))
            call = ops.plane_sweep_homography(src, tgt, outputs["logits"], sigma,
                                              outputs["distance"], outputs["norm"], T, inputs["K"],
                                              inputs["inv_K"], use_mixture_loss=mix,
                                              automask=automask, render_probability=render,
                                              dists=dists, return_mean=True, plane_uniform=uniform,
                                              stereo_rows=stereo_rows, defer=True)

            def matrices(T=T):   # the same matrices the sweep itself used (pd_homography_matrices_fwd unless PD_TORCH_HOMOGRAPHY)
                with torch.no_grad():
                    if ops.TORCH_HOMOGRAPHY:
                        ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
                        H_t2s, Rn = ops.homography_matrices(outputs["distance"], outputs["norm"], ex(T), ex(inputs["K"]),
                                                            ex(inputs["inv_K"]))
                    else:
                        H_t2s, Rn = ops.homography_matrices_fused(outputs["distance"], outputs["norm"], T, inputs["K"],
                                                                  inputs["inv_K"])
                        H_t2s, Rn = H_t2s.reshape(B * N, 3, 3), Rn.reshape(B * N, 3)
                return H_t2s, Rn, inputs["inv_K"][:, :3, :3]

            handle = SweepHandle(src=src, logits=outputs["logits"], sigma=sigma, homography=matrices,
                                 use_mixture_loss=mix,
                                 render_probability=render, dists=dists)
        else:
            raise NotImplementedError("warp_type %r: the reference's depth_warp branch raises UnboundLocalError "
                                      "(padding_mask is never assigned, trainer.py:533-538/580); use disp_warp or "
                                      "homography_warp" % (opt.warp_type,))
        calls.append(call)
        handles.append(handle)
    results = ops.plane_sweep_multi(calls) if fuse_sides else [ops._PlaneSweep.apply(*c) for c in calls]
    if not fuse_sides:
        ops.tail_taps(outputs)   # a linked fused decoder tail: disp / depth go through gradient taps from here on (ops.TailLink)
    for target_side, handle, (rgb_rec, ph_map, ph_mean) in zip(self.target_sides, handles, results):
        outputs[("rgb_rec", target_side)] = rgb_rec
        outputs[("ph_map", target_side)] = ph_map
        outputs[("ph_mean", target_side)] = ph_mean  # ph_map.mean(), accumulated by the sweep kernel itself
        outputs[("sweep", target_side)] = handle
        if getattr(opt, "materialize_layers", False):
            for k, v in handle.layers().items():
                outputs[(k, target_side)] = v


def compute_reprojection_loss(self, pred, target):
    """Computes reprojection loss between a batch of predicted and target images (reference trainer.py:687-699)."""
    return ops.reprojection_loss(pred, target, use_ssim=bool(self.opt.use_ssim))


def compute_losses(self, inputs, outputs):
    """Compute the reprojection and smoothness losses for a minibatch (reference trainer.py:701-773)."""
    opt = self.opt
    B, N, H, W = outputs["probability"].shape
    losses = {"loss/ph_loss": 0, "loss/pc_loss": 0}
    if opt.alpha_self > 0.0:
        losses["loss/self_loss"] = 0
    losses["loss/total_loss"] = 0
    cname = _color_name(opt)
    for target_side in self.target_sides:
        total_loss = 0
        pred = outputs[("rgb_rec", target_side)]
        target = inputs[(cname, target_side)]
        mask = outputs["mask_novel"] if "mask_novel" in outputs else None
        if mask is None:
            # mixture NLL or mean_c |rgb_rec - target| (+ automask min) AND its `.mean()` (trainer.py:742) come out
            # of the sweep kernel; the backward takes the scalar's gradient directly
            ph_loss = outputs[("ph_mean", target_side)]
        else:
            # mask_novel is produced after pred_novel_images (trainer.py:342-349), so the blend, the masked loss and its
            # mean are one small kernel here (mixture: the sweep's ph_map * mask; L1: on the blended prediction, with
            # the automask's min) instead of a chain of [B,3,H,W] torch operators
            mix = bool(opt.use_mixture_loss)
            pred, ph_loss = ops.masked_photometric(
                pred, target, mask, ph_map=outputs[("ph_map", target_side)] if mix else None,
                source=inputs[(cname, "l")] if (opt.automask and not mix) else None)
        losses["loss/ph_loss"] += ph_loss
        total_loss += ph_loss

        # perceptual net (stock VGG, out of scope) — called exactly as the reference does; it back-propagates
        # into rgb_rec, which is why the fused op takes an upstream gradient for rgb_rec.
        if not opt.automask:
            pc_loss = self.perceptual_loss(pred, target).mean()
        else:
            pc_loss = self.perceptual_loss(pred, target, inputs[(cname, "l")]).mean()
        losses["loss/pc_loss"] += pc_loss
        total_loss += opt.alpha_pc * pc_loss

        if opt.alpha_self > 0.0:
            self_loss = compute_reprojection_loss(self, outputs[("self_rec", target_side)], inputs[(cname, "l")]).mean()
            losses["loss/self_loss"] += self_loss
            total_loss += opt.alpha_self * self_loss

        if opt.self_distillation > 0:
            disp_loss = torch.abs(outputs["disp"] - outputs["disp_pp"]).mean()
            losses["loss/disp_loss"] = disp_loss
            total_loss += opt.self_distillation * disp_loss

        losses["loss/total_loss"] += total_loss

    n_sides = len(self.target_sides)
    for k in list(losses.keys()):
        losses[k] = losses[k] / n_sides

    # trainer.py:768 crops both operands at 0.2 W; the crop goes into the operator (ops.smooth_loss_disp, x0) so that
    # autograd has no slice to undo
    smooth_loss = ops.smooth_loss_disp(outputs["disp"], inputs[("color", "l")], opt.gamma_smooth, x0=int(0.2 * W))
    losses["loss/smooth_loss"] = smooth_loss
    losses["loss/total_loss"] = losses["loss/total_loss"] + opt.alpha_smooth * smooth_loss
    return losses


def generate_post_process_disp(self, inputs):
    """Self-distillation targets (reference trainer.py:404-466): run the FIXED networks on cat([image, mirrored image])
    exactly as the reference does (:405-419 — the networks are the reference's own, untouched), then the occlusion-aware
    blend of the two predictions through the fused warp kernels instead of five grid_samples."""
    opt = self.opt
    image = inputs[("color_aug", "l")]
    input_images = ops.cat_flip(image, image)                                  # [image ; mirrored image]
    # the mirrored crop sees the scene with x negated (trainer.py:406-419): the batch-doubling kernel does both at once
    input_grids = ops.cat_flip(inputs["grid"], inputs["grid"], negate_c0=True) if opt.num_ep > 0 else None
    if opt.net_type == "ResNet":
        features = self.fixed_models["encoder"](input_images)
        outputs = self.fixed_models["depth"](features, input_grids)
    elif opt.net_type == "PladeNet":
        outputs = self.models["plade"](input_images, input_grids)
    elif opt.net_type == "FalNet":
        outputs = self.models["fal"](input_images)
    else:
        raise ValueError("unknown net_type %r" % (opt.net_type,))
    # xy and xz planes have disparities that are constant along x (depth_decoder.py:153-181); yz planes do not (:221-236): taken
    # from the options and verified on the data once per trainer object, as pred_novel_images does (opt.pd_check_contract)
    dl = outputs["disp_layered"]
    row_uniform = getattr(opt, "yz_levels", None) == 0
    check = getattr(opt, "pd_check_contract", None)
    if os.environ.get("PD_CHECK_CONTRACT"):
        check = True
    if check is None:
        check = not getattr(self, "_pd_pp_contract_checked", False)
    if check and row_uniform and dl.dim() == 4 and dl.shape[-1] > 1 and not bool((dl == dl[..., :1]).all()):
        raise ValueError("opt.yz_levels == 0 promises disparities that are constant along x, but disp_layered is not")
    try:
        self._pd_pp_contract_checked = True
    except AttributeError:
        pass
    disp_pp, mask_novel = ops.post_process_disp(outputs["logits"], outputs["probability"], outputs["disp"], dl,
                                                row_uniform=row_uniform)
    return disp_pp.detach(), mask_novel.detach()


def pred_self_images(self, inputs, outputs):
    """Reproject the right view into the left one through the predicted depth (reference trainer.py:605-633): depth from
    ``outputs["disp"]`` -> BackprojectDepth -> Project3D -> ``F.grid_sample(padding_mode="border")``, each a HIP kernel
    here.  Writes ``outputs["self_rec"]`` as the reference does.  (Only live with ``--alpha_self`` > 0, where the
    reference itself then fails on a key mismatch in compute_losses; kept for the module-level parity.)"""
    from .layers import BackprojectDepth, Project3D
    disp = outputs["disp"]
    B, N, H, W = outputs["probability"].shape
    depth = 0.1 * 0.58 * W / disp
    T = inputs[("Rt", "r")]
    bp = getattr(self, "backproject_depth", None) or BackprojectDepth(H, W)
    pj = getattr(self, "project_3d", None) or Project3D(H, W)
    cam_points = bp(depth, inputs["inv_K"])
    pix_coords = pj(cam_points, inputs["K"], T)
    features = inputs[(_color_name(self.opt), "r")]
    outputs["self_rec"] = ops.grid_sample(features, pix_coords, padding_mode="border")


def add_flip_right_inputs(self, inputs):
    """Batch doubling with the mirrored other view (reference trainer.py:252-276): the same dict, one kernel per image
    tensor instead of a flip copy plus a cat copy.  Inputs are data: no gradients."""
    # The reference calls this on the DataLoader's CPU batch, BEFORE process_batch moves it to the device
    # (trainer.py:294-295 vs 328-329): take the tensors to the trainer's device first (a no-op when they already are
    # there; process_batch's own .to(device) then is one as well).
    dev = getattr(self, "device", None)
    if dev is None:
        dev = next((v.device for v in inputs.values() if torch.is_tensor(v) and v.is_cuda), torch.device("cuda"))
    inputs = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in inputs.items()}
    new_inputs = {}
    for key in ("color", "color_aug", "depth_gt"):
        if (key, "l") in inputs and (key, "r") in inputs:
            new_inputs[(key, "l")] = ops.cat_flip(inputs[(key, "l")], inputs[(key, "r")])
            new_inputs[(key, "r")] = ops.cat_flip(inputs[(key, "r")], inputs[(key, "l")])
    new_inputs["grid"] = ops.cat_flip(inputs["grid"], inputs["grid"], negate_c0=True)   # :258-261
    for key in ("K", "inv_K", ("Rt", "l"), ("Rt", "r")):
        new_inputs[key] = inputs[key].repeat(2, 1, 1)
    # frames -1 / +1 are doubled with their own mirror image (:271-274)
    for f in self.opt.novel_frame_ids:
        for key in ("color", "color_aug"):
            new_inputs[(key, f)] = ops.cat_flip(inputs[(key, f)], inputs[(key, f)])
    return new_inputs


def patch_trainer(trainer_cls):
    """Bind the fused hot path onto a reference-style Trainer class (drop-in)."""
    trainer_cls.pred_novel_images = pred_novel_images
    trainer_cls.compute_reprojection_loss = compute_reprojection_loss
    trainer_cls.compute_losses = compute_losses
    trainer_cls.generate_post_process_disp = generate_post_process_disp
    trainer_cls.add_flip_right_inputs = add_flip_right_inputs
    trainer_cls.pred_self_images = pred_self_images
    return trainer_cls
