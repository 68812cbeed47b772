"""A/B of library builds on the headline kernels, all in ONE process on one box: every library named on the command line
(`name=path`, or a bare variant name -> planedepth_amd/lib/libpd_var_<name>.so; `product` = the product library) is loaded
side by side and runs the same forward -> backward sequence (the caches are then in the state the training step leaves
them in), interleaved round by round; HIP events on the launch stream around each C-ABI call.

    python scripts/diag_kernel_ab.py [--impl 0|2] [--rounds 5] [--iters 40] [--batch 8 ...] product occ4 abl16 product@7 ...

`name@impl` runs that library under another pd_sweep_impl than --impl (e.g. product@7 = PD_IMPL_ROW_SINGLES next to product);
--check compares every arm's outputs with the first arm's bit for bit (max |difference| per tensor).
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from planedepth_amd import _capi as C  # noqa: E402
from planedepth_amd.synthetic import survey_fullsize_case  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in C.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--planes", type=int, default=49)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--automask", action="store_true")
    ap.add_argument("--fwd_only", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    c = survey_fullsize_case(B=args.batch, N=args.planes, H=args.height, W=args.width, seed=1234)
    c = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()}
    B, N, H, W = c["logits"].shape
    plane = c["disp_pp"][:, :, 0, 0].contiguous()
    flags = C.PD_MIXTURE | (C.PD_AUTOMASK if args.automask else 0) | C.PD_PH_MEAN_ZEROED   # (no memset launch in front of the forward)
    d = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, flags, 1.0, args.impl)
    libs, descs, loaded = [], {}, {}
    for spec in args.libs:
        name, _, path = spec.partition("=")
        base, _, impl = name.partition("@")
        if not path:
            path = (os.path.join(ROOT, "planedepth_amd", "lib", "libplanedepth_hip.so") if base == "product"
                    else os.path.join(ROOT, "planedepth_amd", "lib", "libpd_var_%s.so" % base))
        if path not in loaded:
            loaded[path] = load(path)
        libs.append((name, loaded[path]))
        descs[name] = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, flags, 1.0, int(impl) if impl else args.impl)
    lib0 = libs[0][1]
    k = lib0.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
    rgb = torch.empty(B, 3, H, W, device=dev)
    ph = torch.empty(B, 1, H, W, device=dev)
    stash = torch.empty(B, k, H, W, device=dev)
    gl, gs, gp = torch.empty_like(c["logits"]), torch.empty_like(c["sigma"]), torch.empty_like(plane)
    ws = torch.empty(max(l.pd_sweep_bwd_workspace_floats(ctypes.byref(d)) for _, l in libs), device=dev)
    phm = torch.zeros(1, device=dev)
    gphm = torch.ones(1, device=dev)
    st = C.stream_handle(dev)

    def fwd(lib, d=d):
        rc = lib.pd_plane_sweep_fwd(ctypes.byref(d), C.ptr(c["color_l"]), C.ptr(c["color_r"]), C.ptr(c["logits"]),
                                    C.ptr(c["sigma"]), C.ptr(plane), None, None, None, None, C.ptr(rgb), C.ptr(ph), C.ptr(phm),
                                    C.ptr(stash), st)
        assert rc == 0, lib.pd_last_error()

    def bwd(lib, d=d):
        rc = lib.pd_plane_sweep_bwd(ctypes.byref(d), C.ptr(c["color_l"]), C.ptr(c["color_r"]), C.ptr(c["logits"]),
                                    C.ptr(c["sigma"]), C.ptr(plane), None, None, None, None, C.ptr(rgb), C.ptr(stash),
                                    C.ptr(c["g_rgb_rec"]), None, C.ptr(gphm), C.ptr(gl), C.ptr(gs), C.ptr(gp), None, C.ptr(ws), st)
        assert rc == 0, lib.pd_last_error()

    res = {name: {"fwd": [], "bwd": []} for name, _ in libs}
    ref = None
    for name, lib in libs:   # warm up (code objects, power state)
        for _ in range(20):
            fwd(lib, descs[name])
            if not args.fwd_only:
                gp.zero_()
                bwd(lib, descs[name])
        if args.check:
            torch.cuda.synchronize()
            outs = {"rgb_rec": rgb.clone(), "ph_map": ph.clone(), "stash": stash[:, :3].clone(), "g_logits": gl.clone(), "g_sigma": gs.clone(), "g_plane": gp.clone()}
            if ref is None:
                ref = outs
            else:
                print("check %-20s vs %s: " % (name, libs[0][0]) + ", ".join(
                    "%s %.3g (of %.3g)" % (k, float((outs[k] - ref[k]).abs().max()), float(ref[k].abs().max())) for k in outs))
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for name, lib in libs:
            ev = []
            for _ in range(args.iters):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); fwd(lib, descs[name]); e[1].record()
                if not args.fwd_only:
                    bwd(lib, descs[name])
                e[2].record()
                ev.append(e)
            torch.cuda.synchronize()
            skip = args.iters // 4
            res[name]["fwd"].append(sum(e[0].elapsed_time(e[1]) for e in ev[skip:]) / (len(ev) - skip))
            res[name]["bwd"].append(sum(e[1].elapsed_time(e[2]) for e in ev[skip:]) / (len(ev) - skip))
    print("%-22s %9s %9s %9s %9s   (ms; mean / min over %d rounds of %d, impl %d)" % ("library", "fwd", "fwd min", "bwd", "bwd min", args.rounds, args.iters, args.impl))
    summary = {}
    for name, _ in libs:
        f, b = res[name]["fwd"], res[name]["bwd"]
        summary[name] = {"fwd_ms": sum(f) / len(f), "fwd_min_ms": min(f), "bwd_ms": sum(b) / len(b), "bwd_min_ms": min(b)}
        print("%-22s %9.4f %9.4f %9.4f %9.4f" % (name, sum(f) / len(f), min(f), sum(b) / len(b), min(b)))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"args": vars(args), "results": summary}, fh, indent=1)


if __name__ == "__main__":
    main()
