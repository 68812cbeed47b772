"""Row-stream backward vs row-shift backward (PD_IMPL_ROWS1) on one case: where do g_logits / g_sigma differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from gpu_cases import run_product
from planedepth_amd import _capi as C, ops
from planedepth_amd.synthetic import survey_fullsize_case

case = survey_fullsize_case(sigma_interior=True, B=1, N=63, H=192, W=640, n_xz=14)
run, opt_extra = dict(automask=True), dict(yz_levels=0, xz_levels=14)
res = {}
for name, impl in (("stream", C.PD_IMPL_AUTO), ("shift", C.PD_IMPL_ROWS1)):
    ops.SWEEP_IMPL = impl
    res[name] = run_product(case, run, opt_extra=opt_extra)
ops.SWEEP_IMPL = C.PD_IMPL_AUTO
for k in ("g_logits", "g_sigma", "g_disp_pp"):
    a, b = res["stream"][k], res["shift"][k]
    d = (a - b).abs()
    print(k, "max diff", float(d.max()), "max ref", float(b.abs().max()))
    if k == "g_disp_pp":
        continue
    top = torch.topk(d.flatten(), 12)
    for v, idx in zip(top.values, top.indices):
        n_, y_, x_ = int(idx) // (192 * 640) % 63, int(idx) // 640 % 192, int(idx) % 640
        dd = float((case["disp_pp"].expand(-1, -1, 192, 640) * case["row_gain"])[0, n_, y_, 0])
        print("   top", float(v), "n", n_, "y", y_, "x", x_, "disp", dd, "stream", float(a.flatten()[idx]), "shift", float(b.flatten()[idx]))
    bad = (d > 1e-4 * b.abs().max()).nonzero()
    import collections
    print("  x%128 histogram", sorted(collections.Counter(int(i[3]) % 128 for i in bad).items())[:20])
    print("  plane histogram", sorted(collections.Counter(int(i[1]) for i in bad).items()))
    print("  bad elements", len(bad))
    planes_rows = sorted({(int(i[1]), int(i[2])) for i in bad})
    print("  (plane,row) pairs", len(planes_rows), planes_rows[:40])
    disp = (case["disp_pp"].expand(-1, -1, 192, 640) * case["row_gain"])[0]
    for (n, y) in planes_rows[:12]:
        xs = [int(i[3]) for i in bad if int(i[1]) == n and int(i[2]) == y]
        print("   n", n, "y", y, "disp", float(disp[n, y, 0]), "mask", float(case["padding_mask"][0, n, y, 0]), "x", xs[:10], "...", len(xs),
              "stream", [float(a[0, n, y, x]) for x in xs[:3]], "shift", [float(b[0, n, y, x]) for x in xs[:3]])
