#!/usr/bin/env python
"""bench.py — images/sec of the warp+loss hot path (forward + backward) on MI355X.

Metric (BASELINE.json): images/sec (warp+loss fwd+bwd), 192x640x49 planes, 1/2/4/8 GPU.
A "step" = one pass of the hot path over one synthetic minibatch per GPU:
    pred_novel_images (plane sweep, disp_warp, stereo target "r", mixture loss) -> compute_losses (photometric part)
    -> backward to logits / sigma / per-plane disparities, with an upstream gradient on rgb_rec as well.
Workload at N=1 = BASELINE.json configs[1]: batch 8, 192x640, 49 planes, --use_mixture_loss --plane_residual.
Multi-GPU: the path shards over batch elements with no data-path collective: every rank runs its own
shard (weak scaling); ranks only meet in the timing barrier and the max-over-ranks reduction.

One JSON line is printed by rank 0.  Besides the driver's contract it carries
  "roofline":     HBM roofline of the dominant kernel (algorithmic bytes / measured kernel time, HIP events)
  "cpu_baseline": the CPU oracle (a port of the reference's op sequence) timed on this box's host cores
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

with open(os.path.join(ROOT, "BASELINE.json")) as _f:
    METRIC = json.load(_f)["metric"]  # "images/sec (warp+loss fwd+bwd), 192x640x49 planes, 1/2/4/8 GPU"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--planes", type=int, default=49)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--warp_type", default="disp_warp", choices=["disp_warp", "homography_warp"])
    ap.add_argument("--mono_pose", action="store_true",
                    help="homography_warp only: the pose of a novel frame as Trainer.predict_poses produces it without "
                         "COLMAP (BASELINE configs[3]: pose_net): small rotation, zero translation, Rt[3,3] = 0")
    ap.add_argument("--launch", default="auto", choices=["auto", "eager", "graph"],
                    help="how the timed steps are issued: eager (one host launch per kernel), graph (HIP-graph replay of one "
                         "captured step), auto (default): both are timed for 30 steps before the window and the faster one "
                         "runs it — eager on a box whose host keeps ahead of the device (the eager step costs the host "
                         "0.24 ms, the device 0.29), graph replay (0.02 ms of host per step) on one that does not")
    ap.add_argument("--hip_graph", action="store_true",
                    help="capture one step (forward + backward, every launch of it) in a HIP graph and time replays: "
                         "takes the host-side launch cost of the small torch operators around the sweep out of the step")
    ap.add_argument("--mono_sides", action="store_true",
                    help="homography_warp only: BASELINE configs[3] as the trainer runs it — target_sides = ['r', -1, 1] "
                         "(trainer.py:532, 717): the stereo view plus two novel frames with pose_net-like poses, three "
                         "sweeps per step over the same decoder outputs")
    ap.add_argument("--render_probability", action="store_true",
                    help="alpha compositing over the planes instead of the softmax (trainer.py:584-591; needs the decoder's "
                         "outputs['dists'], a synthetic [B,N-1,H,W] leaf here)")
    ap.add_argument("--per_view_nodes", action="store_true",
                    help="--mono_sides: one autograd node per target view (opt.pd_fuse_sides = False) instead of one node "
                         "for all views with in-kernel gradient accumulation")
    ap.add_argument("--general_stereo", action="store_true",
                    help="homography_warp, stereo target: keep the general per-plane-homography kernels instead of the "
                         "per-row-shift form the stereo extrinsic allows (opt.pd_stereo_rows = False)")
    ap.add_argument("--colmap_pose", action="store_true",
                    help="homography_warp only: small rotation + translation per image (--use_colmap poses): a "
                         "different homography per plane, every sample with 4 live taps")
    ap.add_argument("--no_mixture", action="store_true")
    ap.add_argument("--automask", action="store_true")
    ap.add_argument("--xz_levels", type=int, default=0,
                    help="reference option: 0 (BASELINE's '49 planes') lets the path skip the decoder's all-ones padding "
                         "mask; >0 feeds it as a dense [B,N,H,W] tensor like the decoder with ground planes does")
    ap.add_argument("--yz_levels", type=int, default=0,
                    help="append K yz (vertical-wall) planes: disparities that vary along x and half-image masks "
                         "(depth_decoder.py:209-252) — a genuinely dense [B,N,H,W] map: the general forward + the row-dense backward")
    ap.add_argument("--no_padding_mask", action="store_true", help="never pass a padding mask (diagnostics)")
    ap.add_argument("--no_plane_grad", action="store_true", help="diagnostics: disparities do not require grad")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_next_rows", action="store_true", help="skip the timings of the SURVEY 8f operators")
    ap.add_argument("--skip_context", default="", help="comma list of context measurements to skip: copy")
    ap.add_argument("--no_ddp_step", action="store_true", help="skip the end-to-end DDP training-step block")
    ap.add_argument("--ddp_model", default="r18", choices=["r18", "r50"],
                    help="stand-in depth network of the ddp_step block: ResNet-18-shaped (59.6 MB of gradients, BASELINE configs[1]) "
                         "or ResNet-50 + dense-ASPP-shaped (156.6 MB, configs[2])")
    ap.add_argument("--cpu_seconds", type=float, default=20.0, help="budget for the CPU baseline sample")
    ap.add_argument("--keep_gc", action="store_true", help="leave Python's cyclic garbage collector on during the timed windows (A/B)")
    ap.add_argument("--autograd_threads", action="store_true",
                    help="leave autograd's device thread on for the timed windows (A/B).  Default: torch.autograd.set_multithreading_enabled(False) "
                         "around the launch probe and the timed windows — one process per GPU has no use for the hand-over of every "
                         "backward to another thread, which costs the host 0.1 ms per step whenever the two threads sit on distant cores "
                         "(scripts/diag_step_host.py: 97-116 us per step against 118-210)")
    ap.add_argument("--windows", type=int, default=7, help="further timed windows of --steps steps after the one `value` reports (spread)")
    return ap.parse_args()


def make_batch(args, device, seed):
    from planedepth_amd.synthetic import survey_fullsize_case
    torch.manual_seed(seed)
    c = survey_fullsize_case(B=args.batch, N=args.planes + args.xz_levels + args.yz_levels, H=args.height, W=args.width,
                             seed=1234 + seed, n_xz=args.xz_levels + args.yz_levels)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in c.items()}


def bench_pose(args, c, device):
    """[B,4,4] pose of the target view: the rectified stereo baseline (inputs[("Rt","r")], mono_dataset.py:203-211), or
    --mono_pose: what Trainer.predict_poses hands over for a novel frame WITHOUT COLMAP (trainer.py:386-400):
    a small rotation conjugated by the crop matrix, ZERO translation and Rt[3,3] = 0 — the homography is then the same
    for every plane; --colmap_pose: rotation + translation (trainer.py:397-398), a different homography per plane."""
    B = c["logits"].shape[0]
    if not (args.mono_pose or args.colmap_pose):
        return c["Rt"].clone()
    from planedepth_amd.synthetic import small_pose
    Rt = small_pose(torch.Generator().manual_seed(77), B, stereo=False).to(device)
    if args.mono_pose:
        Rt[:, :3, 3] = 0.0
        Rt[:, 3, 3] = 0.0
    return Rt


def build_step(args, c, device):
    """Returns step() running the product path exactly as a patched Trainer would (dict contract of trainer.py)."""
    import planedepth_amd
    B, N, H, W = c["logits"].shape
    mix = not args.no_mixture
    logits = c["logits"].clone().requires_grad_(True)
    sigma = c["sigma"].clone().requires_grad_(True)
    disp_pp = c["disp_pp"].clone().requires_grad_(not args.no_plane_grad)  # per-plane disparities incl. the learnt residual
    Rt = bench_pose(args, c, device)
    opt = types.SimpleNamespace(warp_type=args.warp_type, match_aug=False, use_mixture_loss=mix, automask=args.automask,
                                render_probability=bool(args.render_probability), alpha_pc=0.0, alpha_self=0.0, self_distillation=0.0,
                                gamma_smooth=2.0, alpha_smooth=0.0, use_ssim=False, materialize_layers=False,
                                xz_levels=args.xz_levels, yz_levels=args.yz_levels)
    zero = torch.zeros((), device=device)
    # --mono_pose: the target is a novel frame (-1), whose pose predict_poses builds without translation -> the
    # plane-uniform kernels; --colmap_pose: a novel frame with a translation (opt.use_colmap) -> the general kernels
    side = -1 if (args.mono_pose or args.colmap_pose) else "r"
    sides = ["r", -1, 1] if args.mono_sides else [side]
    opt.use_colmap = bool(args.colmap_pose)
    opt.pd_stereo_rows = not args.general_stereo
    opt.pd_fuse_sides = not args.per_view_nodes
    ns = types.SimpleNamespace(opt=opt, target_sides=sides, perceptual_loss=lambda *a, **k: zero)
    inputs = {("color", "l"): c["color_l"], "K": c["K"], "inv_K": c["inv_K"]}
    poses = {}
    for i, sd in enumerate(sides):   # the novel frames get their own images (a shifted copy) and poses
        inputs[("color", sd)] = c["color_r"] if i == 0 else torch.roll(c["color_r"], shifts=7 * i, dims=3).contiguous()
        poses[sd] = Rt
    if args.mono_sides:
        margs = argparse.Namespace(**dict(vars(args), mono_pose=True))
        poses[-1] = bench_pose(margs, c, device)
        poses[1] = bench_pose(margs, c, device).transpose(1, 2).contiguous()   # the opposite rotation
    norm = torch.tensor([0.0, 0.0, 1.0], device=device)[None, None].expand(B, N, -1).contiguous()
    # outputs["distance"] is a DECODER output (networks/depth_decoder.py:254), like logits and sigma: a leaf that wants a
    # gradient, handed to the path as the decoder hands it over — not re-derived from the disparities inside the timed step
    # (that was ten 2-5 us launches of decoder arithmetic and its autograd per step: 10-15 % of a homography step)
    distance = (0.1 * 0.58 * W / disp_pp.detach()[:, :, 0, 0]).contiguous().requires_grad_(not args.no_plane_grad)
    decoder_geometry = None
    if args.warp_type == "homography_warp" and args.xz_levels and not args.yz_levels:
        # SURVEY.md 8d D-inputs (4): the reference decoder's own plane set — `planes` xy planes + `xz_levels` ground planes with
        # their normals [0, 1, t] / |.| and distances h / |.| (networks/depth_decoder.py:158-207), the horizon mask from
        # grid = linspace(-1, 1) (the Resize grid, pair_transforms.py:63-64) multiplied into the logits as the decoder does
        # (:255-256), level residuals as in C-golden (rand - 0.5)
        from planedepth_amd.synthetic import crop_grid, decoder_plane_geometry
        grid = crop_grid(H, W, H, W, 0, 0)[None].expand(B, -1, -1, -1).to(device)
        res = (torch.rand(B, N, generator=torch.Generator().manual_seed(4242)) - 0.5).to(device)
        decoder_geometry = decoder_plane_geometry(grid, res, no_levels=args.planes, xz_levels=args.xz_levels, rows=True)
        norm = decoder_geometry["norm"].contiguous()
        distance = decoder_geometry["distance"].contiguous().requires_grad_(not args.no_plane_grad)
        logits = (c["logits"] * decoder_geometry["padding_mask"]).contiguous().requires_grad_(True)
    shape_probe = torch.empty(B, N, H, W, device="meta")
    g_rgb = c["g_rgb_rec"]
    one = torch.ones((), device=device)  # d loss / d ph_loss
    pm = None if args.no_padding_mask else c["padding_mask"]
    if decoder_geometry is not None and pm is not None:   # (homography_warp computes its own facing mask; the decoder's mask is
        pm = decoder_geometry["padding_mask"].expand(-1, -1, -1, W)   # still part of `outputs`, constant along x: an expand view)
    if pm is None:
        pm_arg = None
    else:
        pm_arg = pm

    dists = None
    if args.render_probability:
        dists = (torch.rand(B, N - 1, H, W, generator=torch.Generator().manual_seed(5)) * 2.0).to(device).requires_grad_(True)
    dense_disp = None
    if args.xz_levels or args.yz_levels:  # the decoder cat()s xy and xz planes into a dense [B,N,H,W] map (depth_decoder.py:182): that is
        # decoder work, so it is built ONCE here and handed to the path as the decoder would hand it over (a dense
        # tensor that wants a gradient), not re-materialised inside the timed step
        dense_disp = (disp_pp.detach().expand(-1, -1, H, W) * c["row_gain"]).contiguous()
        if args.yz_levels:   # the last K planes as vertical walls: disparity linear in x, visible on one half of the image each
            K = args.yz_levels
            xr = torch.linspace(-1.0, 1.0, W, device=device)[None, None, None, :]
            dense_disp[:, N - K:] = dense_disp[:, N - K:, :1, :1] * (0.05 + xr.abs())
            pm_arg = pm_arg.clone() if torch.is_tensor(pm_arg) else torch.ones(B, N, H, W, device=device)
            pm_arg[:, N - K:N - K // 2] = (xr >= 1e-7).float()
            pm_arg[:, N - K // 2:] = (xr <= -1e-7).float()
        dense_disp.requires_grad_(not args.no_plane_grad)

    def step():
        logits.grad = sigma.grad = disp_pp.grad = None
        if dists is not None:
            dists.grad = None
        if dense_disp is not None:
            disp_layered = _DecoderSide.apply(dense_disp)
        else:
            disp_layered = disp_pp.expand(-1, -1, H, W)
        outputs = {"probability": shape_probe, "logits": logits, "sigma": sigma,
                   "disp_layered": disp_layered, "padding_mask": pm_arg, "norm": norm}
        for sd in sides:
            outputs[("Rt", sd)] = poses[sd]
        if dists is not None:
            outputs["dists"] = dists
        if args.warp_type == "homography_warp":  # only the homography reads the plane distances (trainer.py:557)
            distance.grad = None
            outputs["distance"] = distance
        planedepth_amd.pred_novel_images(ns, inputs, outputs)
        # photometric part of compute_losses (trainer.py:717-742) + a stand-in for the perceptual net's gradient
        # ph_mean = ph_map.mean() (trainer.py:742), accumulated by the sweep kernel
        heads = [outputs[(k, sd)] for sd in sides for k in ("ph_mean", "rgb_rec")]
        torch.autograd.backward(heads, [one, g_rgb] * len(sides))
        return heads[0]

    return step, (logits, sigma, disp_pp, distance)


class _DecoderSide(torch.autograd.Function):
    """The decoder's side of ``outputs["disp_layered"]`` in the --xz_levels configuration: hands the dense map over as the
    non-leaf tensor it is in the trainer (depth_decoder.py:182, a cat) and takes the path's gradient without doing anything
    with it — the decoder's backward is not part of the path, just as no backward of the networks runs behind
    logits.grad / sigma.grad.  (As a LEAF the map would make autograd's AccumulateGrad clone the path's stride-0 gradient
    into 248 MB of memory per step, which no trainer does.)"""

    @staticmethod
    def forward(ctx, dense):
        return dense.view_as(dense)

    @staticmethod
    def backward(ctx, g):
        return None


def algorithmic_bytes(args):
    """per image per target side, fp32.  mixture: fwd (2N+9) HW 4, bwd (4N+9) HW 4."""
    HW = args.height * args.width
    N = args.planes + args.xz_levels + args.yz_levels
    k = 2 if not args.no_mixture else 1
    fwd = (k * N + 9) * HW * 4
    bwd = (2 * k * N + 9) * HW * 4
    return fwd, bwd


def kernel_times(args, c, device, iters):
    """Average duration of the forward and backward launches, HIP events on the stream the kernels run on."""
    from planedepth_amd import _capi as C
    lib = C.load()
    B, N, H, W = c["logits"].shape
    mix = not args.no_mixture
    if args.xz_levels or args.yz_levels or args.render_probability:
        return None  # direct-launch timing is wired for the xy-plane softmax configurations only
    if args.warp_type == "homography_warp" and (args.mono_sides or not (args.mono_pose or args.colmap_pose or args.general_stereo)):
        return None  # stereo target: runs as per-row shifts on the row-shift kernels; the in-step events time those
    flags = (C.PD_MIXTURE if mix else 0) | (C.PD_AUTOMASK if args.automask else 0)
    pm = None
    aux = k3 = None
    if args.warp_type == "disp_warp":
        mode, sign = C.PD_WARP_DISP, 1.0
        plane = c["disp_pp"][:, :, 0, 0].contiguous()
    else:  # the [B*N,3,3] algebra of HomographyWarp stays in torch (ops.homography_matrices); the kernels take H_t2s
        from planedepth_amd import ops
        mode, sign = C.PD_WARP_HOMOGRAPHY, 0.0
        ex = lambda M: M[:, None].expand(-1, N, -1, -1).reshape(B * N, 4, 4)  # noqa: E731
        norm = torch.tensor([0.0, 0.0, 1.0], device=device)[None, None].expand(B, N, -1)
        Tp = bench_pose(args, c, device)
        dist_ = 0.1 * 0.58 * W / c["disp_pp"][:, :, 0, 0]
        plane, aux = ops.homography_matrices(dist_, norm, ex(Tp), ex(c["K"]), ex(c["inv_K"]))
        if args.mono_pose:   # zero translation: one homography per image, the plane-uniform kernels ([B,4,3,3] layout)
            h1, _ = ops.homography_matrices(dist_[:, :1], norm[:, :1], Tp, c["K"], c["inv_K"])
            plane = h1[:, None].expand(-1, 4, -1, -1)
            pm = (norm / dist_[..., None]).contiguous()   # translation weights n/d
            flags |= C.PD_HOMO_UNIFORM
        plane, aux, k3 = plane.contiguous(), aux.contiguous(), c["inv_K"][:, :3, :3].contiguous()
    d = C.SweepDesc(B, N, H, W, mode, flags, sign, int(os.environ.get("PD_SWEEP_IMPL", 0)))
    k = lib.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
    rgb = torch.empty(B, 3, H, W, device=device)
    ph = torch.empty(B, 1, H, W, device=device)
    stash = torch.empty(B, k, H, W, device=device)
    gl, gs, gp = torch.empty_like(c["logits"]), torch.empty_like(c["sigma"]), torch.empty_like(plane)
    ws = torch.empty(lib.pd_sweep_bwd_workspace_floats(ctypes.byref(d)), device=device)
    phm = torch.empty(1, device=device)     # fused mean of ph_map and its upstream gradient (d loss / d mean = 1)
    gphm = torch.ones(1, device=device)
    st = C.stream_handle(device)
    sig = c["sigma"] if mix else None

    def fwd():
        C.check(lib.pd_plane_sweep_fwd(ctypes.byref(d), C.ptr(c["color_l"]), C.ptr(c["color_r"]), C.ptr(c["logits"]),
                                       C.ptr(sig), C.ptr(plane), C.ptr(aux), C.ptr(k3), C.ptr(pm), None, C.ptr(rgb),
                                       C.ptr(ph), C.ptr(phm), C.ptr(stash), st), "fwd")

    def bwd():
        C.check(lib.pd_plane_sweep_bwd(ctypes.byref(d), C.ptr(c["color_l"]), C.ptr(c["color_r"]), C.ptr(c["logits"]),
                                       C.ptr(sig), C.ptr(plane), C.ptr(aux), C.ptr(k3), C.ptr(pm), None, C.ptr(rgb),
                                       C.ptr(stash), C.ptr(c["g_rgb_rec"]), None, C.ptr(gphm), C.ptr(gl),
                                       C.ptr(gs if mix else None), C.ptr(None if args.no_plane_grad else gp), None,
                                       C.ptr(ws), st), "bwd")

    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        fn()
        torch.cuda.synchronize(device)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize(device)
        out[name] = sum(a.elapsed_time(b) for a, b in ev) / iters  # ms
    return out


def in_step_kernel_times(step, device, iters, skip=0):
    """Average duration of the sweep's forward / backward launches INSIDE the training step (CUDA events recorded on the
    launch stream around the C-ABI calls, planedepth_amd.ops.KERNEL_EVENTS): what the kernels take with the caches in
    the state the step leaves them in, as opposed to an isolated launch loop."""
    from planedepth_amd import ops
    ops.KERNEL_EVENTS = {"fwd": [], "bwd": []}
    try:
        for _ in range(iters):
            step()
        torch.cuda.synchronize(device)
        ev = ops.KERNEL_EVENTS
    finally:
        ops.KERNEL_EVENTS = None
    if not ev["fwd"] or not ev["bwd"]:
        return None
    # (the first `skip` steps are left out of the average: a fresh process ramps for its first ~50 steps, main())
    return {k: sum(a.elapsed_time(b) for a, b in v[skip:]) / len(v[skip:]) for k, v in ev.items()}


def next_rows_times(args, device, iters=100):
    """SURVEY.md §8f rows (decoder tail, smoothness loss, post-process) at the headline shape: average ms per call from
    CUDA events around the public operators (the launches are on torch's current stream), with the algorithmic bytes
    each moves.  Reported next to the headline number; not part of `value`."""
    from planedepth_amd import ops
    B, N, H, W = args.batch, args.planes, args.height, args.width
    g = torch.Generator().manual_seed(99)
    mk = lambda *shape: torch.randn(*shape, generator=g).to(device)  # noqa: E731
    raw_l, raw_s = mk(B, N, H, W).requires_grad_(True), mk(B, N, H, W).requires_grad_(True)
    lv = (torch.arange(N, dtype=torch.float32)[None, :, None, None] + torch.rand(B, N, 1, 1, generator=g) - 0.5).to(device)
    lv.requires_grad_(True)
    gl, gs, gd = mk(B, N, H, W), mk(B, N, H, W), mk(B, 1, H, W)
    img = torch.rand(B, 3, H, W, generator=g).to(device)
    x0 = int(0.2 * W)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):      # back to back, one pair of events around the lot: what a training step sees (the host
            fn()                    # enqueues ahead of the device where it can; an event pair per call would add its own cost)
        b.record()
        torch.cuda.synchronize(device)
        return a.elapsed_time(b) / iters

    def tail_fwd():
        with torch.no_grad():
            dl = ops.plane_disparities(lv, 2.0, 300.0, W)[0].expand(-1, -1, H, W)
            return ops.decoder_tail(raw_l, raw_s, None, dl)

    def tail_fwd_bwd():
        dl = ops.plane_disparities(lv, 2.0, 300.0, W)[0].expand(-1, -1, H, W)   # (depth_decoder.py:150-152 in one launch each way)
        logits, sigma, disp, depth, _ = ops.decoder_tail(raw_l, raw_s, None, dl)
        torch.autograd.backward([logits, sigma, disp], [gl, gs, gd])
        raw_l.grad = raw_s.grad = lv.grad = None

    disp_leaf = (torch.rand(B, 1, H, W, generator=g) * 50).to(device).requires_grad_(True)

    def smooth_fwd_bwd():
        ops.smooth_loss_disp(disp_leaf, img, 2.0, x0=x0).backward()     # the call compute_losses makes (trainer.py:768)
        disp_leaf.grad = None

    Bp = max(1, B // 2)
    pl, pp = mk(2 * Bp, N, H, W), torch.softmax(mk(2 * Bp, N, H, W), 1)
    pdl = (300.0 * (2.0 / 300.0) ** (lv.detach()[:1].expand(2 * Bp, -1, -1, -1) / (N - 1))).expand(-1, -1, H, W)
    pdisp = (pp * pdl).sum(1, True)

    def post():
        ops.post_process_disp(pl, pp, pdisp, pdl)

    pred_leaf = torch.rand(B, 3, H, W, generator=g).to(device).requires_grad_(True)
    g_rl = mk(B, 1, H, W)

    def reproj_fwd_bwd():   # 0.85 * SSIM(3x3) + 0.15 * L1, the non-mixture photometric loss (layers.py:276-306)
        ops.reprojection_loss(pred_leaf, img, True).backward(g_rl)
        pred_leaf.grad = None

    def timed_graph(fn):
        """The same call captured once in a HIP graph and replayed: what the DEVICE spends on it.  The eager figure of an
        operator whose kernels take 30-50 us is the host's enqueue rate on whatever CPU the box has (0.05-0.16 ms measured
        across boxes); inside a training step those launches are enqueued far ahead of the device."""
        try:
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(3):
                    fn()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fn()
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize(device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                graph.replay()
            b.record()
            torch.cuda.synchronize(device)
            return round(a.elapsed_time(b) / iters, 4)
        except Exception as e:   # a capture that fails must not take the bench line with it
            torch.cuda.synchronize(device)
            return "graph capture failed: %s" % type(e).__name__

    raw_lp = raw_l.detach()[:, :N - 1].contiguous().requires_grad_(True)     # PladeNet's conv0: N - 1 logit channels
    gt = mk(B, N - 1, H, W)

    def plade_fwd_bwd():   # PladeNet's tail with --render_probability (plade_net.py:309-341): the producer of outputs["dists"]
        dl = ops.plane_disparities(lv, 2.0, 300.0, W)[0].expand(-1, -1, H, W)   # (plade_net.py:280-285 in one launch each way)
        logits, dists, sigma, disp, depth, _ = ops.plade_tail(raw_lp, raw_s, dl)
        torch.autograd.backward([logits, dists, sigma, disp], [gl, gt, gs, gd])
        raw_lp.grad = raw_s.grad = lv.grad = None

    t_f, t_fb, t_s, t_p = timed(tail_fwd), timed(tail_fwd_bwd), timed(smooth_fwd_bwd), timed(post)
    t_pl = timed(plade_fwd_bwd)
    t_r = timed(reproj_fwd_bwd)
    g_s, g_r, g_pl = timed_graph(smooth_fwd_bwd), timed_graph(reproj_fwd_bwd), timed_graph(plade_fwd_bwd)
    hw4 = H * W * 4
    tail_f_bytes, tail_b_bytes = (3 * N + 4) * hw4 * B, (6 * N + 4) * hw4 * B
    post_bytes = (2 * (2 * N) + 2 * N + N + 3) * hw4 * Bp   # 2 warp-softmax (read N, write N) + 3 warp-sums (read N)
    return {
        "decoder_tail": {"fwd_ms": round(t_f, 4), "fwd_bwd_ms": round(t_fb, 4),
                         "fwd_GBs": round(tail_f_bytes / (t_f * 1e-3) / 1e9, 1),
                         "fwd_bwd_GBs": round((tail_f_bytes + tail_b_bytes) / (t_fb * 1e-3) / 1e9, 1),
                         "shape": [B, N, H, W]},
        # fwd reads 2N-1, writes 3N-1 (+3) planes; bwd reads 2N-1 + 3N-1 (+5), writes 2N-1
        "plade_tail_render": {"fwd_bwd_ms": round(t_pl, 4), "fwd_bwd_device_ms": g_pl,
                              "fwd_bwd_GBs": round((12 * N - 4 + 8) * hw4 * B / ((g_pl if isinstance(g_pl, float) else t_pl) * 1e-3) / 1e9, 1),
                              "shape": [B, N, H, W]},
        "smooth_loss": {"fwd_bwd_ms": round(t_s, 4), "fwd_bwd_device_ms": g_s, "shape": [B, 1, H, W - x0]},
        "reprojection_loss_ssim_l1": {"fwd_bwd_ms": round(t_r, 4), "fwd_bwd_device_ms": g_r, "shape": [B, 3, H, W]},
        "post_process": {"ms": round(t_p, 4), "GBs": round(post_bytes / (t_p * 1e-3) / 1e9, 1),
                         "shape": [2 * Bp, N, H, W]},
        "note": "average over %d back-to-back calls of the public operators, one CUDA-event pair around the lot; fwd_bwd_device_ms is the same "
                "call replayed from a HIP graph, i.e. the device's share (and what plade_tail_render's GB/s is computed from).  With autograd on "
                "the calling thread (host_autograd; torch.autograd.set_multithreading_enabled(False)) the eager figures of the small operators sit "
                "at their device times; under torch's device thread (--autograd_threads) they are host-paced at 0.09-0.15 ms (the hand-over of "
                "every backward to another thread, NOTEBOOK 11.6).  PladeNet's per-plane disparities come from ops.plane_disparities (one launch "
                "each way); "
                "what plade_tail_render's GB/s is computed from)" % iters,
    }


def resnet_shaped_depth_net(kind, n_planes):
    """Stand-in for the reference's depth network with the REAL gradient volume: torchvision is not in the
    image, so the ResNet is restated with stock nn.Conv2d / nn.BatchNorm2d blocks (BasicBlock stacks [2,2,2,2] for "r18",
    Bottleneck stacks [3,4,6,3] for "r50") under a monodepth-style skip decoder (networks/depth_decoder.py:30-60:
    num_ch_dec = [16,32,64,128,256]) that ends in the decoder's three heads.  r18: 14.9 M parameters = 59.6 MB of fp32
    gradients; r50 adds a dense-ASPP-sized block on the bottleneck: 39.2 M = 156.6 MB.  Random init, synthetic inputs:
    what is measured is the step's structure and its gradient all-reduce, not accuracy."""
    import torch.nn as nn

    def cbr(ci, co, k=3, s=1):
        return nn.Sequential(nn.Conv2d(ci, co, k, s, k // 2, bias=False), nn.BatchNorm2d(co), nn.ReLU(inplace=True))

    class Basic(nn.Module):
        def __init__(self, ci, co, s):
            super().__init__()
            self.a, self.b = cbr(ci, co, 3, s), nn.Sequential(nn.Conv2d(co, co, 3, 1, 1, bias=False), nn.BatchNorm2d(co))
            self.sc = None if (s == 1 and ci == co) else nn.Sequential(nn.Conv2d(ci, co, 1, s, bias=False), nn.BatchNorm2d(co))

        def forward(self, x):
            return torch.relu(self.b(self.a(x)) + (x if self.sc is None else self.sc(x)))

    class Bottle(nn.Module):
        def __init__(self, ci, co, s):
            super().__init__()
            m = co // 4
            self.a, self.b = cbr(ci, m, 1), cbr(m, m, 3, s)
            self.c = nn.Sequential(nn.Conv2d(m, co, 1, bias=False), nn.BatchNorm2d(co))
            self.sc = None if (s == 1 and ci == co) else nn.Sequential(nn.Conv2d(ci, co, 1, s, bias=False), nn.BatchNorm2d(co))

        def forward(self, x):
            return torch.relu(self.c(self.b(self.a(x))) + (x if self.sc is None else self.sc(x)))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            deep = kind == "r50"
            block, depths, mul = (Bottle, [3, 4, 6, 3], 4) if deep else (Basic, [2, 2, 2, 2], 1)
            self.stem = cbr(3, 64, 7, 2)
            self.pool = nn.MaxPool2d(3, 2, 1)
            chans, ci, stages = [64 * mul, 128 * mul, 256 * mul, 512 * mul], 64, []
            for i, (co, d) in enumerate(zip(chans, depths)):
                stages.append(nn.Sequential(*[block(ci if j == 0 else co, co, (2 if (j == 0 and i > 0) else 1)) for j in range(d)]))
                ci = co
            self.stages = nn.ModuleList(stages)
            enc = [64] + chans
            self.aspp = None
            if deep:   # DenseASPP-sized context block on the bottleneck (networks/depth_decoder.py with --use_denseaspp)
                self.aspp = nn.Sequential(cbr(enc[-1], 512, 1), *[cbr(512, 512, 3) for _ in range(2)], cbr(512, enc[-1], 1))
            dec = [16, 32, 64, 128, 256]
            self.neck = None if deep else nn.Sequential(nn.Conv2d(256, 256, 3, 1, 1), nn.ELU(inplace=True))   # (brings r18 to 59.7 MB)
            self.up0, self.up1 = nn.ModuleList(), nn.ModuleList()
            for i in range(4, -1, -1):
                cin = enc[-1] if i == 4 else dec[i + 1]
                self.up0.append(nn.Sequential(nn.Conv2d(cin, dec[i], 3, 1, 1), nn.ELU(inplace=True)))
                self.up1.append(nn.Sequential(nn.Conv2d(dec[i] + (enc[i - 1] if i > 0 else 0), dec[i], 3, 1, 1), nn.ELU(inplace=True)))
            self.dispconv = nn.Conv2d(16, n_planes, 3, 1, 1)
            self.sigmaconv = nn.Conv2d(16, n_planes, 3, 1, 1)
            self.residualconv = nn.Conv2d(16, n_planes, 1)

        def forward(self, x):
            x = (x - 0.45) / 0.225
            f = [self.stem(x)]
            y = self.pool(f[0])
            for st in self.stages:
                y = st(y)
                f.append(y)
            x = f[-1] if self.aspp is None else f[-1] + self.aspp(f[-1])
            for k, i in enumerate(range(4, -1, -1)):
                x = self.up0[k](x)
                if k == 0 and self.neck is not None:
                    x = self.neck(x)
                x = nn.functional.interpolate(x, scale_factor=2, mode="nearest")
                if i > 0:
                    x = torch.cat([x, f[i - 1]], 1)
                x = self.up1[k](x)
            res = torch.sigmoid(self.residualconv(x).mean((2, 3), keepdim=True)) - 0.5      # depth_decoder.py:151
            return self.dispconv(x), self.sigmaconv(x), res

    return Net()


def ddp_step_block(args, device, rank, world, steps=8, warmup=3):
    """End-to-end training step with the reference's structure (trainer.py:278-323) on synthetic KITTI-like inputs made
    ON the device: add_flip_right_inputs (B/2 -> B) -> a stand-in conv encoder/decoder wrapped in
    DistributedDataParallel (stock DDP over RCCL; torchvision is not in the image, so the ResNet is replaced by a small
    conv U-net that emits the decoder's conv outputs) -> fused decoder tail -> pred_novel_images -> compute_losses ->
    backward (DDP all-reduces the network gradients while the autograd engine is still in the sweep's backward) -> Adam.
    Reported next to the headline number; `value` stays the hot path alone."""
    import torch.distributed as dist
    import torch.nn as nn
    import planedepth_amd
    from planedepth_amd import ops
    from planedepth_amd.decoder_tail import fused_decoder_tail
    from planedepth_amd.synthetic import kitti_like_inputs_on_device
    N, H, W, B = args.planes, args.height, args.width, args.batch
    if B % 2:
        return {"skipped": "--flip_right doubling needs an even per-GPU batch"}
    shared = bool(os.environ.get("PD_BENCH_SHARE_GPU"))   # functional check (all ranks on one GPU, gloo): fewer steps per leg
    if shared:
        steps, warmup = 3, 1
    hot_steps, cap_steps = (4, 2) if shared else (10, 5)

    model_kind = args.ddp_model
    torch.manual_seed(100 + rank)
    model = resnet_shaped_depth_net(model_kind, N)
    if world > 1 or os.environ.get("PD_BENCH_SYNC_BN"):
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)                          # trainer.py:98
    model = model.to(device)
    own_group = False
    if not dist.is_initialized():          # single GPU: a one-rank group, so that the step really runs under DDP
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(os.environ.get("PD_BENCH_BACKEND", "nccl"), rank=0, world_size=1,
                                init_method="tcp://127.0.0.1:%d" % port)
        own_group = True
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=[device.index], find_unused_parameters=True)  # trainer.py:99
    optim = torch.optim.Adam(ddp.parameters(), 1e-4, betas=(0.5, 0.999))                                       # :102
    n_params = sum(p.numel() for p in model.parameters())
    opt = types.SimpleNamespace(warp_type="disp_warp", match_aug=False, use_mixture_loss=True, automask=False,
                                render_probability=bool(args.render_probability), alpha_pc=0.0, alpha_self=0.0, self_distillation=0.0,
                                gamma_smooth=2.0, alpha_smooth=0.04, use_ssim=False, xz_levels=0, yz_levels=0,
                                novel_frame_ids=[], flip_right=True)
    zero = torch.zeros((), device=device)

    class Stub:
        pass
    planedepth_amd.patch_trainer(Stub)
    trainer = Stub()
    trainer.opt, trainer.target_sides, trainer.device = opt, ["r"], device
    trainer.perceptual_loss = lambda *a, **k: zero
    base = kitti_like_inputs_on_device(B // 2, H, W, seed=1000 + rank, device=device)
    levels0 = torch.arange(N, device=device, dtype=torch.float32)[None, :, None, None]

    fuse_tail = [True]   # the sweep's backward applies the fused decoder tail's backward (pd_plane_sweep_bwd_tail); [False]: two kernels

    def step(sync=True):
        inputs = trainer.add_flip_right_inputs(base)                                    # trainer.py:294-295
        ctx = ddp if sync else ddp.module
        raw_l, raw_s, res = ctx(inputs[("color_aug", "l")])
        disp_layered = (300.0 * (2.0 / 300.0) ** ((levels0 + res) / (N - 1))).expand(-1, -1, H, W)   # depth_decoder.py:148-156
        outputs = {"disp_layered": disp_layered, "padding_mask": None}
        fused_decoder_tail(outputs, raw_l, raw_s, use_mixture_loss=True, all_ones_mask=True, fuse_sweep_backward=fuse_tail[0])
        trainer.pred_novel_images(inputs, outputs)                                      # :342
        losses = trainer.compute_losses(inputs, outputs)                                # :354
        optim.zero_grad(set_to_none=True)                                               # :299
        losses["loss/total_loss"].backward()                                            # :300
        optim.step()                                                                    # :301
        return losses["loss/total_loss"]

    def timed(fn, n):
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / n

    for _ in range(warmup):
        step()
    t_step = timed(step, steps)
    t_nosync = timed(lambda: step(False), steps)      # the same step without DDP's hooks: what the all-reduce adds
    def hot_kernels():
        ops.KERNEL_EVENTS = {"fwd": [], "bwd": []}
        try:
            for _ in range(hot_steps):
                step()
            torch.cuda.synchronize(device)
            return {k: sum(a.elapsed_time(b) for a, b in v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in ops.KERNEL_EVENTS.items() if v}
        finally:
            ops.KERNEL_EVENTS = None
    hot = hot_kernels()
    fuse_tail[0] = False          # the same step with the tail's backward as its own kernel (round 4's form), for comparison
    for _ in range(2):
        step()
    t_unfused = timed(step, steps)
    hot_unfused = hot_kernels()
    fuse_tail[0] = True
    flat = torch.empty(n_params, device=device)
    t_allreduce = timed(lambda: dist.all_reduce(flat), 5) if world > 1 else 0.0
    if world > 1:
        t_step = parallel_max(t_step, device)
    block = {"images_per_sec": round(B * world / t_step, 1), "ms_per_step": round(t_step * 1e3, 3),
             "ms_per_step_without_gradient_sync": round(t_nosync * 1e3, 3),
             "sweep_fwd_ms": round(hot["fwd"], 4), "sweep_bwd_ms": round(hot["bwd"], 4),
             # the sweep's producer and consumer in a real step: the fused decoder tail writes logits / sigma right before the
             # sweep's forward and reads g_logits / g_sigma right after its backward
             "tail_fwd_ms": round(hot.get("tail_fwd", float("nan")), 4),
             # 0: the tail's backward rides in the sweep's backward kernel (fused_decoder_tail(..., fuse_sweep_backward=True))
             "tail_bwd_ms": round(hot.get("tail_bwd", 0.0), 4),
             "tail_backward_as_its_own_kernel": {"ms_per_step": round((parallel_max(t_unfused, device) if world > 1 else t_unfused) * 1e3, 3),
                                                 "sweep_bwd_ms": round(hot_unfused["bwd"], 4),
                                                 "tail_bwd_ms": round(hot_unfused.get("tail_bwd", float("nan")), 4)},
             "hot_path_share_of_step": round((hot["fwd"] + hot["bwd"]) * 1e-3 / t_step, 4),
             "network": "%s-shaped stand-in (stock Conv2d/BatchNorm2d blocks + skip decoder + the decoder's three heads), %d "
                        "parameters = %.1f MB of fp32 gradients (SURVEY C1: 59.6 / 156.6 MB), %s, "
                        "DistributedDataParallel(find_unused_parameters=True) as trainer.py:98-99"
                        % ("ResNet-18" if model_kind == "r18" else "ResNet-50 + dense-ASPP", n_params, n_params * 4 / 1e6,
                           "SyncBatchNorm" if any(isinstance(m, nn.SyncBatchNorm) for m in model.modules()) else
                           "BatchNorm2d (one rank: convert_sync_batchnorm applies from 2 ranks on)"),
             "batch_per_gpu": B, "backend": dist.get_backend(), "world_size": dist.get_world_size(),
             "structure": "flip_right doubling -> DDP(conv net) -> fused decoder tail -> plane sweep -> losses -> backward -> Adam"}
    if world > 1:
        exposed = max(t_step - t_nosync, 0.0)
        block.update(allreduce_alone_ms=round(t_allreduce * 1e3, 3), allreduce_exposed_ms=round(exposed * 1e3, 3),
                     allreduce_hidden_share=round(1.0 - min(exposed / t_allreduce, 1.0), 3) if t_allreduce > 0 else None,
                     bucket_cap_mb=25)
        # xGMI rings are per-link bound: how much of the gradient all-reduce stays exposed as the bucket size changes
        by_bucket = {}
        for cap in (10, 50, 100):
            del ddp
            ddp = nn.parallel.DistributedDataParallel(model, device_ids=[device.index], find_unused_parameters=True,
                                                      bucket_cap_mb=cap)
            for _ in range(2):
                step()
            t_cap = parallel_max(timed(step, cap_steps), device)
            by_bucket[str(cap)] = {"ms_per_step": round(t_cap * 1e3, 3),
                                   "allreduce_exposed_ms": round(max(t_cap - t_nosync, 0.0) * 1e3, 3)}
        block["by_bucket_cap_mb"] = by_bucket
    else:
        block["allreduce_hidden_share"] = None   # one rank: nothing to overlap; measured from --gpus 2 upwards
    del ddp
    if own_group:
        dist.destroy_process_group()
    return block


def measured_copy_rate(device, mbytes=384, iters=20):
    """What this box's HBM gives a plain device-to-device copy right now (torch's own copy kernel; bytes read + bytes
    written per second): context for the roofline fractions, measured in this run — the spec's 8 TB/s is the denominator
    everywhere, this is what a kernel with no arithmetic and perfect access order reaches on the day."""
    n = mbytes * (1 << 20) // 4
    a, b = torch.empty(n, device=device), torch.empty(n, device=device)
    a.fill_(1.0)
    b.copy_(a)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    return {"GBs": round(2 * n * 4 / (ms * 1e-3) / 1e9, 1), "frac_of_peak": round(2 * n * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "what": "torch copy_ of %d MiB (read + write bytes), %d launches, measured in this run" % (mbytes, iters)}


def parallel_max(v, device):
    from planedepth_amd import parallel
    return parallel.max_over_ranks(v, device)


def measured_traffic(args, kernel):
    """(HBM bytes per launch, where the figure comes from).  bench.py cannot run the profiler around itself, so this is
    NOT a measurement of this run: it is read from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    scripts/gpu_profile.sh on this exact workload) and labelled as such; (None, reason) for any other workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        same = (w["batch"] == args.batch and w["planes"] == args.planes and w["height"] == args.height and
                w["width"] == args.width and w["mixture"] == (not args.no_mixture) and args.xz_levels == 0 and args.yz_levels == 0 and
                not args.automask and args.warp_type == "disp_warp")
        if not same:
            return None, "no PMC pass committed for this workload"
        return int(t[kernel]), "profiles/traffic.json: rocprofv3 FETCH_SIZE/WRITE_SIZE passes of %s (%s), not of this run" % (
            t.get("taken", "round 1"), t.get("note", "see profiles/"))
    except Exception:
        return None, "profiles/traffic.json unreadable"


def cpu_baseline(args, budget_s):
    """The oracle (port of the reference's op-by-op PyTorch path) on the host cores: B=1 sample of the same workload.
    Sampling goes through F.grid_sample — the operator the reference itself calls (trainer.py:573-577) and the one that
    ships with torch on this box; the oracle's own gather-based restatement of it (oracle.bilinear_sample, 3-4x slower:
    it exists to pin the formula, not to be fast) is timed once and reported beside it."""
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import run_oracle
    from planedepth_amd.synthetic import survey_fullsize_case
    ncpu = os.cpu_count() or 1
    case = survey_fullsize_case(B=1, N=args.planes, H=args.height, W=args.width)
    run = dict(warp_type=args.warp_type, use_mixture_loss=not args.no_mixture, automask=args.automask)
    torch_sampler = lambda f, g, pm: F.grid_sample(f, g, mode="bilinear", padding_mode=pm, align_corners=True)  # noqa: E731

    def one(sampler=torch_sampler):
        tm = {}
        run_oracle(case, run, sampler=sampler, timing=tm)
        return tm["fwd_s"], tm["fwd_bwd_s"]

    # Thread count.  SURVEY 8d asks for os.cpu_count() threads; torch's intra-op pool degrades badly when oversubscribed on the
    # pool's shared hosts (256 threads: 36 s / image), so the sample runs at the best of {8, 32, every core when there are at
    # most 64} — one warm-up + one probe each — and the line states the host's core count AND the threads used.
    probes = {}
    for th in sorted({min(8, ncpu), min(32, ncpu)} | ({ncpu} if ncpu <= 64 else set())):
        torch.set_num_threads(th)
        one()   # warm-up at this setting
        probes[th] = one()[1]
    threads = min(probes, key=probes.get)
    torch.set_num_threads(threads)
    fwd, both = [], []
    t_end = time.perf_counter() + budget_s
    while len(both) < 9 and (time.perf_counter() < t_end or len(both) < 5):
        f, fb = one()
        fwd.append(f); both.append(fb)
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    t0 = time.perf_counter()
    run_oracle(case, run)   # the restated sampler, once (warm pool): the second figure
    restated = time.perf_counter() - t0
    # one thread: one warm-up + up to three runs inside a budget of its own
    torch.set_num_threads(1)
    one()
    f1, fb1 = [], []
    t_end = time.perf_counter() + max(6.0, budget_s / 2)
    while len(fb1) < 3 and (time.perf_counter() < t_end or not fb1):
        f, fb = one()
        f1.append(f); fb1.append(fb)
    torch.set_num_threads(threads)
    return {"value": round(1.0 / med(both), 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "host_cores": ncpu, "threads": threads,
            "thread_probe_ms": {str(k): round(v * 1e3, 1) for k, v in sorted(probes.items())},
            "sample": "oracle (torch CPU restatement of trainer.py:523-603,717-742, sampling through F.grid_sample as the "
                      "reference does) B=1, N=%d, %dx%d, one warm-up + median of %d at %d threads of %d host cores; value = fwd+bwd"
                      % (args.planes, args.height, args.width, len(both), threads, ncpu),
            "fwd_ms": round(med(fwd) * 1e3, 2), "fwd_bwd_ms": round(med(both) * 1e3, 2),
            "ms_per_image": round(med(both) * 1e3, 2),
            "one_thread": {"value": round(1.0 / med(fb1), 4), "fwd_ms": round(med(f1) * 1e3, 1), "fwd_bwd_ms": round(med(fb1) * 1e3, 1),
                           "runs": len(fb1)},
            "restated_sampler": {"value": round(1.0 / restated, 4), "ms_per_image": round(restated * 1e3, 2),
                                 "what": "same pass with the oracle's gather-based bilinear_sample instead of F.grid_sample, one run"}}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU over RCCL
    (what the reference's `torchrun --nproc_per_node=K train.py` does, train_ResNet.sh:1 / trainer.py:50-55).  On a box
    with fewer than N devices the ranks only start with PD_BENCH_SHARE_GPU=1 (functional check: all ranks on cuda:0,
    gloo for the timing collectives since RCCL refuses two ranks on one device)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    if ndev < args.gpus:
        if not env.get("PD_BENCH_SHARE_GPU"):
            raise SystemExit("bench.py --gpus %d: only %d device(s) visible (set PD_BENCH_SHARE_GPU=1 to run all ranks "
                             "on cuda:0 as a functional check)" % (args.gpus, ndev))
        env.setdefault("PD_BENCH_BACKEND", "gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] spawning %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    from planedepth_amd import parallel
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    # RCCL ("nccl"); used for the timing barrier / max reduction only.  PD_BENCH_BACKEND=gloo + PD_BENCH_SHARE_GPU=1 is
    # a functional check of the multi-process flow on a one-GPU box (all ranks on cuda:0).
    rank, world, local_rank = parallel.init_process_group_from_env(os.environ.get("PD_BENCH_BACKEND", "nccl"))
    if world != args.gpus:
        print("[bench] note: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d"
              % (args.gpus, world, world), file=sys.stderr, flush=True)
    device = torch.device("cuda", 0 if os.environ.get("PD_BENCH_SHARE_GPU") else local_rank)
    torch.cuda.set_device(device)
    import __graft_entry__ as entry
    entry.build()

    from planedepth_amd import _capi as _C0
    build_flags = int(_C0.load().pd_build_flags())
    if build_flags & 2:
        raise SystemExit("bench.py: the library was built with -DPD_DIAGNOSTICS (timing ablations: wrong results by design); "
                         "its timings are not a benchmark")
    c = make_batch(args, device, seed=rank)  # every rank draws its own shard: no data-path collective
    step, _ = build_step(args, c, device)
    eager_step = step

    def capture_step():
        # whole-step capture (the pattern torch documents for graphs with a backward): warm up on a side stream so every
        # lazy initialisation (workspace sizes, kernel attributes) has happened, then record one step; the product's
        # launches go to torch's current stream, which is the capturing one.
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            eager_step()
        return graph.replay

    if args.hip_graph:
        args.launch = "graph"
    # capture FIRST, before any eager step has run on the default stream (autograd's AccumulateGrad nodes remember the
    # stream they were created on; a capture after eager steps breaks for the multi-view configurations)
    launch_probe = None
    graph_step = None
    if args.launch in ("auto", "graph"):
        try:
            graph_step = capture_step()
        except Exception as e:   # a capture that fails must not take the bench line with it
            torch.cuda.synchronize(device)
            launch_probe = {"graph_capture_failed": type(e).__name__}
    # The kernel-timing legs of the roofline block run BEFORE the timed window, on every rank: a process's first ~50
    # steps run ~10 % slower than its steady state (the same 20 steps take 0.355 ms after 5 warm-up steps, 0.316 ms after
    # 200: scripts/gpu_r3_warm.sh, DESIGN.md section 6 — the device's power state, not the host: a HIP-graph replay shows
    # the same ramp), and a training run lives in the steady state.  ~400 launches of the same kernels, ~100 ms;
    # the in-step kernel figures are averaged over the second half of their leg for the same reason.
    leg_iters = max(50, min(args.steps, 100))
    kt = in_step_kernel_times(eager_step, device, iters=2 * leg_iters, skip=leg_iters)   # the figure the roofline uses
    iso = kernel_times(args, c, device, iters=leg_iters)
    pre_timed = {"in_step_kernel_timing_steps": 2 * leg_iters, "isolated_fwd_launches": (leg_iters + 1) if iso else 0,
                 "isolated_bwd_launches": (leg_iters + 1) if iso else 0, "warmup_steps": args.warmup,
                 "hip_graph_capture_steps": 0,
                 "why": "steady state: a fresh process runs its first ~50 steps ~10 % slower (device power state; DESIGN.md "
                        "section 6), and the roofline legs need the kernel times anyway"}
    pre_timed["steps_total"] = pre_timed["in_step_kernel_timing_steps"] + args.warmup + pre_timed["hip_graph_capture_steps"]
    if "copy" not in args.skip_context:
        hbm_copy = measured_copy_rate(device)
    # Host hygiene for the 5 ms windows: an eager step costs the host 0.10-0.12 ms (0.20-0.25 in the bad mode of torch's device thread, NOTEBOOK 11.6) of the 0.26 ms the device needs, so a pause of
    # Python's cyclic collector (a generation-2 pass walks everything the set-up legs left behind) lands in a window twice: as
    # the pause itself, and as the ~50 slower steps after any idle gap long enough for the device to clock down (measured: a
    # collect() right in front of the warm-up steps cost the first window 10 %).  Collected and frozen HERE, with the launch
    # probe's 124 steps still ahead; switched off for the timed windows below; --keep_gc leaves it alone (A/B).
    import gc
    if not args.keep_gc:
        gc.collect()
        gc.freeze()
    if not args.autograd_threads:
        torch.autograd.set_multithreading_enabled(False)   # (back on ahead of the DDP step block)
    # ---- how the timed steps are issued ----
    # every rank must take the same path through the probe's collectives: a capture that failed anywhere means eager everywhere
    if args.launch in ("auto", "graph") and parallel.max_over_ranks(0.0 if graph_step is not None else 1.0, device) > 0.0:
        graph_step = None
    if args.launch == "graph" and graph_step is not None:
        step = graph_step
    elif args.launch == "auto" and graph_step is not None:
        def probe(fn, n=30):
            fn()
            torch.cuda.synchronize(device)
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(device)
            return (time.perf_counter() - t) / n * 1e3
        t_eager, t_graph = probe(eager_step), probe(graph_step)
        t_eager, t_graph = min(t_eager, probe(eager_step)), min(t_graph, probe(graph_step))
        # every rank must take the same decision (the ranks meet in the timing barrier): the slowest rank's view decides
        t_eager, t_graph = parallel.max_over_ranks(t_eager, device), parallel.max_over_ranks(t_graph, device)
        # both forms are always in the line (`value` must not switch methodology silently between boxes or rounds)
        launch_probe = {"eager_ms_per_step": round(t_eager, 4), "graph_ms_per_step": round(t_graph, 4), "steps_each": 2 * 31,
                        "eager_images_per_sec": round(args.batch * world / (t_eager * 1e-3), 1),
                        "graph_images_per_sec": round(args.batch * world / (t_graph * 1e-3), 1)}
        step = graph_step if t_graph < t_eager else eager_step
    used_graph = step is not eager_step
    pre_timed["hip_graph_capture_steps"] = 4 if graph_step is not None else 0
    pre_timed["launch_probe_steps"] = 4 * 31 if (launch_probe and "steps_each" in launch_probe) else 0
    pre_timed["steps_total"] = (pre_timed["in_step_kernel_timing_steps"] + args.warmup + pre_timed["hip_graph_capture_steps"] +
                                pre_timed["launch_probe_steps"])
    # (the collector is off from here to the end of the timed windows: see the collect / freeze above the launch probe)
    if not args.keep_gc:
        gc.disable()
    for _ in range(args.warmup):
        step()
    parallel.barrier(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    parallel.barrier(device)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device)

    value = parallel.throughput(args.batch, args.steps, world, elapsed)   # images (not image-views) per second
    # Spread: `value` above is the contract's window; the same K steps are timed again --windows times, each
    # window bracketed the same way (barrier + synchronize, max over ranks), so the line shows how far one 5 ms window is from
    # the next on THIS box.  The device-to-device copy rate is taken before and after as the box's own yardstick.
    window_rates = []
    for _ in range(max(0, args.windows)):
        parallel.barrier(device)
        tw = time.perf_counter()
        for _ in range(args.steps):
            step()
        parallel.barrier(device)
        window_rates.append(parallel.throughput(args.batch, args.steps, world, parallel.max_over_ranks(time.perf_counter() - tw, device)))
    if not args.keep_gc:
        gc.enable()
        gc.unfreeze()
    hbm_copy_after = measured_copy_rate(device) if "copy" not in args.skip_context else None
    result = {
        "metric": METRIC, "value": round(value, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "launch": "HIP graph replay of one captured step" if used_graph else "eager (one host launch per kernel)",
        "launch_policy": args.launch, "launch_probe": launch_probe,
        "host_gc": "on" if args.keep_gc else "collected, frozen and disabled for the timed windows",
        "host_autograd": "device thread (torch's default)" if args.autograd_threads else "calling thread (torch.autograd.set_multithreading_enabled(False)) for the launch probe, the timed windows and the operator legs after them (fast_rows_option, next_rows); torch's default for the ddp_step block",
        "pre_timed_steps": pre_timed,
        "windows": ({"n": len(window_rates), "steps_each": args.steps, "median": round(sorted(window_rates)[len(window_rates) // 2], 1),
                     "min": round(min(window_rates), 1), "max": round(max(window_rates), 1),
                     "what": "further timed windows of the same length, run right after the window `value` reports (images/sec; "
                             "not part of `value`)"} if window_rates else None),
        "library": dict(entry.BUILD_INFO, build_flags=build_flags),   # "built" here from source, or "reused" (the travelling .so matches this source hash); build_flags 0 = no experiments, no timing ablations compiled in
        "known_deviation": "row kernels' backward (row-stream by default, row-shift under PD_IMPL_ROWS1): the adjoint drops the "
                           "eps-weighted (eps <= 8e-6) term of the neighbouring source row on rows whose y round trip is "
                           "inexact; bounded at 3e-5 of the gradients' range against the general kernels (tests: "
                           "test_rowshift_kernels_vs_general_kernels_and_oracle, "
                           "test_rowshift_adjoint_cross_row_term_is_bounded_at_large_height); forward exact",
        "config": {"workload": "BASELINE configs[1]: %s, %s, %s loss, batch %d/GPU, %dx%d, %d planes, "
                               "grads to logits/sigma/plane disparities + upstream rgb_rec gradient"
                               % (args.warp_type, "target_sides ['r', -1, 1]: stereo + two pose_net frames, 3 sweeps per image" if args.mono_sides else "mono pose (pose_net: rotation only, F8)" if args.mono_pose
                                  else ("colmap pose (rotation + translation)" if args.colmap_pose else
                                        ("stereo target r" + (" as per-row shifts (row-shift kernels)" if args.warp_type == "homography_warp" and not args.general_stereo else ""))), "L1" if args.no_mixture else "Laplacian-mixture", args.batch,
                                  args.height, args.width, args.planes + args.xz_levels + args.yz_levels),
                   "global_batch": args.batch * world, "planes": args.planes + args.xz_levels + args.yz_levels, "height": args.height,
                   "width": args.width, "xz_levels": args.xz_levels, "yz_levels": args.yz_levels, "parallelism": "dp%d (independent shards, no data-path collective)" % world,
                   "padding_mask": "not read (xy planes only: the decoder's mask is all ones by construction)"
                   if (args.no_padding_mask or args.xz_levels + args.yz_levels == 0) else "decoder's dense [B,N,H,W] float mask",
                   # what the timed step contains changed between rounds for the non-default configurations: compare like with like
                   "workload_version": 5,
                   "workload_changes": {"3": "homography_warp: outputs['distance'] is handed over as the decoder's leaf tensor instead of "
                                             "being re-derived from the disparities inside the timed step (was 10-15 % of such a step)",
                                        "4": "--xz_levels: outputs['disp_layered'] is handed over as the decoder's non-leaf dense map; the path "
                                             "returns its gradient as a stride-0 view (no [B,N,H,W] zero fill / clone inside the step)",
                                        "5": "homography_warp with --xz_levels: the decoder's own xz-plane normals / distances / horizon mask "
                                             "(synthetic.decoder_plane_geometry = networks/depth_decoder.py:146-207) instead of frontal normals for "
                                             "every plane: BASELINE configs[3] as SURVEY 8d (4) specifies it; the default workload is unchanged"}},
    }
    if world > 1:
        import torch.distributed as dist
        result["comm"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                          "devices": "shared cuda:0 (PD_BENCH_SHARE_GPU)" if os.environ.get("PD_BENCH_SHARE_GPU")
                          else "one per rank"}
    if rank == 0:
        if "copy" not in args.skip_context:
            result["hbm_copy_measured"] = hbm_copy
            result["hbm_copy_measured_after"] = hbm_copy_after
        if kt:
            fwd_b, bwd_b = algorithmic_bytes(args)
            dom = "bwd" if kt["bwd"] >= kt["fwd"] else "fwd"
            per_launch = (bwd_b if dom == "bwd" else fwd_b) * args.batch
            ach = per_launch / (kt[dom] * 1e-3) / 1e9
            traffic, traffic_src = measured_traffic(args, "pd_plane_sweep_" + dom)
            block = {"bound": "hbm", "kernel": "pd_plane_sweep_" + dom, "achieved": round(ach, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": per_launch,
                     "avg_launch_ms": round(kt[dom], 4),
                     "timing": "HIP events on the launch stream around the C-ABI call (for the default workload the sweep kernel "
                               "alone: ph_mean and g_plane arrive pre-zeroed, PD_PH_MEAN_ZEROED / PD_BWD_PLANE_ZEROED; other "
                               "workloads add their 5 us helper launches), inside the training step, steady state"}
            # headline workload -> "roofline"; the general (homography) kernels report the same block under their own key
            result["roofline" if args.warp_type == "disp_warp" else "roofline_general"] = block
            if args.warp_type != "disp_warp":
                result["roofline"] = dict(block, note="general kernels (homography_warp); the headline disp_warp "
                                                      "kernels are measured by the default invocation")
            # SURVEY.md 8(d)'s own figure: the WHOLE path (images/s x algorithmic bytes per image and view over the spec
            # peak), from the timed window's `value`; `roofline` above is the dominant kernel alone
            n_views = 3 if args.mono_sides else 1
            path_GBs = (fwd_b + bwd_b) * n_views * value / world / 1e9
            result["roofline_path"] = {
                "bound": "hbm", "achieved": round(path_GBs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(path_GBs / HBM_PEAK_GBS, 4), "bytes_per_image": (fwd_b + bwd_b) * n_views,
                "definition": "SURVEY 8d: images/s per GPU x (6N+18) HW 4 bytes per image and view / 8 TB/s (the target "
                              "there: 0.60 = 31.3 k images/s at N = 49, 192x640)"}
            result["kernels"] = {
                "fwd_ms": round(kt["fwd"], 4), "bwd_ms": round(kt["bwd"], 4),
                "fwd_GBs": round(fwd_b * args.batch / (kt["fwd"] * 1e-3) / 1e9, 1),
                "bwd_GBs": round(bwd_b * args.batch / (kt["bwd"] * 1e-3) / 1e9, 1),
                "fwd_frac_of_peak": round(fwd_b * args.batch / (kt["fwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bwd_frac_of_peak": round(bwd_b * args.batch / (kt["bwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if iso:
                result["kernels"].update(isolated_fwd_ms=round(iso["fwd"], 4), isolated_bwd_ms=round(iso["bwd"], 4))
        if world == 1 and args.warp_type == "disp_warp" and not args.no_next_rows:
            # The opt-in approximation next to the reported (exact) number: PD_IMPL_FAST_ROWS drops the second source row of the
            # rows whose y round trip is inexact (weight eps <= 8e-6) in forward and backward — all the traffic the headline
            # kernels move above the algorithmic bytes; rejected as the default by the parity suite.
            # Both legs eager, same process, same tensors: like for like.
            from planedepth_amd import ops as _ops, _capi as _C

            def eager_rate(n=60):
                for _ in range(10):
                    eager_step()
                torch.cuda.synchronize(device)
                t = time.perf_counter()
                for _ in range(n):
                    eager_step()
                torch.cuda.synchronize(device)
                return args.batch * n / (time.perf_counter() - t)
            prev = _ops.SWEEP_IMPL
            try:
                r_exact = eager_rate()
                _ops.SWEEP_IMPL = _C.PD_IMPL_FAST_ROWS
                r_fast = eager_rate()
            finally:
                _ops.SWEEP_IMPL = prev
            result["fast_rows_option"] = {
                "images_per_sec": round(r_fast, 1), "exact_images_per_sec_same_leg": round(r_exact, 1),
                "ratio": round(r_fast / r_exact, 4),
                "what": "PD_SWEEP_IMPL=2 / pd_sweep_desc.impl = PD_IMPL_FAST_ROWS: second source rows whose weight is below 2^-16 dropped in "
                        "forward and backward.  NOT the default and not `value`: the parity suite, run under both row modes, finds it beyond "
                        "BASELINE's 1e-4 on white-noise inputs at the benchmark sizes (worst 3.1e-4, profiles/r05_parity.md) and inside it on "
                        "the golden fixtures, the KATs and band-limited inputs; eager launches, 60 steps each"}
        if world == 1 and not args.no_next_rows:
            result["next_rows"] = next_rows_times(args, device)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            result["speedup_vs_cpu_baseline"] = round(value / result["cpu_baseline"]["value"], 1)
    if not args.autograd_threads:
        torch.autograd.set_multithreading_enabled(True)   # the DDP step block runs under torch's default
    if not args.no_ddp_step and args.warp_type == "disp_warp" and not (args.xz_levels or args.yz_levels):
        # The headline line must not die with the secondary block — neither by an exception nor by a collective that never
        # returns (a rank that failed while the others wait in an all-reduce: RCCL's watchdog would abort the process before
        # anything is printed).  A timer thread prints the line without the block and ends the process if it overruns.
        import threading

        def bail():
            if rank == 0:
                result["ddp_step_timed_out"] = True      # top level: a hang of the secondary block must not hide inside it
                result["ddp_step"] = {"error": "timed out after %d s (PD_DDP_STEP_TIMEOUT_S): the secondary DDP training-step "
                                               "block hung (a collective that never returned?); `value` above was measured "
                                               "before it and is complete" % limit}
                print(json.dumps(result), flush=True)
            os._exit(0)
        limit = int(os.environ.get("PD_DDP_STEP_TIMEOUT_S", "150"))
        timer = threading.Timer(limit, bail)
        timer.daemon = True
        timer.start()
        result["ddp_step_timed_out"] = False
        result["ddp_step_failed"] = False
        try:   # every rank takes part (DDP's collectives); rank 0 reports
            blk = ddp_step_block(args, device, rank, world)
        except Exception as e:
            blk = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            result["ddp_step_failed"] = True
        timer.cancel()
        if rank == 0:
            result["ddp_step"] = blk
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
