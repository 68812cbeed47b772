"""One-off differential fuzz (GPU): the specialised kernels of the `disp_warp` path (segment-stream forward, row-stream /
row-shift backward; PD_IMPL_AUTO) against the general kernels (PD_IMPL_GENERAL: atomic scatter, no row structure) on random
shapes through the public op — odd heights (ragged row groups), widths that are not multiples of the 128-pixel segment, odd
widths (other kernels), one plane, negative shifts, compositing.  Prints every case that differs by more than the suite's bounds.
  python scripts/fuzz_disp_paths.py [--cases 200] [--seed 1]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planedepth_amd import _capi as C, ops  # noqa: E402
from planedepth_amd.synthetic import build_case  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run(c, impl, side, mix, automask, render, H, W):
    ops.SWEEP_IMPL = impl
    lg, sg = c["logits"].clone().requires_grad_(True), c["sigma"].clone().requires_grad_(True)
    dp = c["disp_pp"].clone().requires_grad_(True)
    dists = c["dists"].clone().requires_grad_(True) if render else None
    tgt = c["color_r"] if side == "r" else c["color_l"]
    rgb, ph, mean = ops.plane_sweep_disp(c["color_l"], tgt, lg, sg if mix else None, dp.expand(-1, -1, H, W), None, target_side=side,
                                         use_mixture_loss=mix, automask=automask, render_probability=render, dists=dists,
                                         return_mean=True)
    (mean + (rgb * c["g_rgb_rec"]).sum()).backward()
    out = dict(rgb=rgb, ph=ph, mean=mean.reshape(1), g_l=lg.grad, g_d=dp.grad)
    if mix:
        out["g_s"] = sg.grad
    if render:
        out["g_dists"] = dists.grad
    return {k: v.detach().float().cpu() for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    bad = 0
    for i in range(a.cases):
        B, N = rnd.randint(1, 5), rnd.choice([1, 2, 3, 7, 16, 33, 49, 64, 70])
        H = rnd.choice([2, 3, 4, 5, 7, 23, 47, 48, 95, 97, 130, 192, 211])   # (H, W >= 2: the reference divides by H - 1, W - 1)
        W = rnd.choice([2, 6, 64, 126, 128, 130, 200, 256, 258, 400, 640, 642, 700, 131, 65])
        if B * N * H * W > 40e6:
            H = min(H, 23)
        mix, automask = rnd.random() < 0.7, rnd.random() < 0.5
        render = rnd.random() < 0.25 and N >= 2
        side = rnd.choice(["r", "l"]) if rnd.random() < 0.8 else "r"
        case = build_case(B=B, N=N, H=H, W=W, seed=1000 * a.seed + i, disp_min=0.5, disp_max=max(0.3 * W, 1.0), sigma_interior=True,
                          render_probability=render)
        c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.items()}
        try:
            fast = run(c, C.PD_IMPL_AUTO, side, mix, automask, render, H, W)
            slow = run(c, C.PD_IMPL_GENERAL, side, mix, automask, render, H, W)
        except Exception as e:   # noqa: BLE001
            print("CASE %d (B=%d N=%d H=%d W=%d mix=%d am=%d render=%d side=%s): %s: %s" % (i, B, N, H, W, mix, automask, render, side,
                                                                                       type(e).__name__, e))
            bad += 1
            continue
        for k, v in fast.items():
            tol = 2e-4 if k in ("g_d", "g_dists") else 3e-5
            if N == 1 and k in ("g_l", "g_s"):
                continue   # (one plane: pi = 1 and rgb_rec = c whatever sigma is — the gradients through the softmax and through
                           # rgb_rec are cancellation noise (x 1/sigma^2) in every kernel family; the suite compares them with the oracle)
            if not torch.isfinite(v).all() or (float(slow[k].abs().max()) > 0 and rel(v, slow[k]) > tol):
                print("CASE %d (B=%d N=%d H=%d W=%d mix=%d am=%d render=%d side=%s): %s differs by %.2e" %
                      (i, B, N, H, W, mix, automask, render, side, k, rel(v, slow[k])))
                bad += 1
    ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    print("fuzz: %d cases, %d findings" % (a.cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
