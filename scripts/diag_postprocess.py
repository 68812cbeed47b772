"""Per-call device time of the post-process operators at the benchmark shape (run once per PD_PP_SEG / PD_PP_ROWS setting: the
library reads the switches once per process).  python scripts/diag_postprocess.py [--batch 4 --planes 49]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planedepth_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--planes", type=int, default=49)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--xz", type=int, default=0, help="the last K planes get a disparity that grows with the row (ground planes): a [B,N,H,1] map expanded along x")
    a = ap.parse_args()
    B, N, H, W = a.batch, a.planes, a.height, a.width
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(2 * B, N, H, W, generator=g).to(dev)
    prob = torch.softmax(torch.randn(2 * B, N, H, W, generator=g), 1).to(dev)
    lv = torch.arange(N, dtype=torch.float32)[None, :, None, None] + torch.rand(2 * B, N, 1, 1, generator=g) - 0.5
    dl = (300.0 * (2.0 / 300.0) ** (lv / (N - 1))).to(dev).expand(-1, -1, H, W)
    if a.xz:
        rows = dl[:, :, :, :1].clone()                                    # [2B,N,H,1]
        gain = torch.linspace(0.2, 3.0, H, device=dev).view(1, 1, H, 1)
        rows[:, N - a.xz:] = rows[:, N - a.xz:] * gain
        dl = rows.expand(-1, -1, -1, W)
    disp = (prob * dl).sum(1, True)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / a.iters

    plr = ops.warp_softmax(logits[:B], dl[:B], +1.0)
    o = ops.warp_sum(plr, dl[B:], -1.0)
    rows = [("warp_softmax +1", lambda: ops.warp_softmax(logits[:B], dl[:B], +1.0)),
            ("warp_softmax -1 flip", lambda: ops.warp_softmax(logits[B:], dl[B:], -1.0, flip_src=True)),
            ("warp_sum -1", lambda: ops.warp_sum(plr, dl[B:], -1.0)),
            ("warp_sum +1", lambda: ops.warp_sum(plr, dl[:B], +1.0)),
            ("pp_combine", lambda: ops.pp_combine(disp, o, o)),
            ("post_process_disp", lambda: ops.post_process_disp(logits, prob, disp, dl)),
            ("... stepwise", lambda: ops.post_process_disp_stepwise(logits, prob, disp, dl))]
    hw4 = H * W * 4 * B
    bytes_ = {"warp_softmax +1": 2 * N * hw4, "warp_softmax -1 flip": 2 * N * hw4, "warp_sum -1": (N + 1) * hw4, "warp_sum +1": (N + 1) * hw4,
              "pp_combine": 5 * hw4, "post_process_disp": (7 * N + 3) * hw4, "... stepwise": (7 * N + 3) * hw4}
    print("PD_PP_SEG=%s PD_PP_ROWS=%s PD_PP_CHAIN=%s  %dx%dx%dx%d" % (os.environ.get("PD_PP_SEG", "-"), os.environ.get("PD_PP_ROWS", "-"), os.environ.get("PD_PP_CHAIN", "-"), B, N, H, W))
    for name, fn in rows:
        t = timed(fn)
        print("%-22s %.4f ms  %7.1f GB/s" % (name, t, bytes_[name] / (t * 1e-3) / 1e9))


main()
