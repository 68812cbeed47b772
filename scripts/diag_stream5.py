"""The special-disparity case of the test, per-ROW disparity gradients (PD_DISP_ROWS form) of both backward kernels."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from planedepth_amd import _capi as C, ops
from planedepth_amd.synthetic import build_case

kw = dict(special_disp=[0.0, 0.25, 1.0, 63.0, 64.0, 64.00001, 65.5, 127.99999, 129.0, 200.0], disp_min=0.5, disp_max=9.0)
case = build_case(B=2, N=10, H=7, W=130, seed=4000 + 130, sigma_interior=True, **kw)
c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.items()}
B, N, H, W = c["logits"].shape
SAME = "--same" in sys.argv
for form in ("rows", "plane", "plane_mean"):
    out = {}
    for name, impl in (("stream", C.PD_IMPL_AUTO), ("shift", C.PD_IMPL_ROWS1)):
        ops.SWEEP_IMPL = impl
        lg, sg = c["logits"].clone().requires_grad_(True), c["sigma"].clone().requires_grad_(True)
        if form == "rows":
            dp = c["disp_pp"].expand(B, N, H, 1).contiguous().requires_grad_(True)
            rgb, ph = ops.plane_sweep_disp(c["color_l"], c["color_l" if SAME else "color_r"], lg, sg, dp.expand(B, N, H, W), None, target_side="l", row_uniform=True)
        elif form == "plane":
            dp = c["disp_pp"].clone().requires_grad_(True)
            rgb, ph = ops.plane_sweep_disp(c["color_l"], c["color_l" if SAME else "color_r"], lg, sg, dp.expand(B, N, H, W), None, target_side="l")
        else:
            dp = c["disp_pp"].clone().requires_grad_(True)
            rgb, ph, pm = ops.plane_sweep_disp(c["color_l"], c["color_l" if SAME else "color_r"], lg, sg, dp.expand(B, N, H, W), None, target_side="l", return_mean=True)
        if form == "plane_mean":
            (pm + (rgb * c["g_rgb_rec"]).sum()).backward()
        else:
            (ph.mean() + (rgb * c["g_rgb_rec"]).sum()).backward()
        out[name] = dp.grad.cpu()
    ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    d = (out["stream"] - out["shift"]).abs()
    print(form, "max diff", float(d.max()), "max", float(out["shift"].abs().max()))
    if form == "rows":
        for b in range(B):
            print(" image", b, "plane 0 rows stream", [round(float(v), 5) for v in out["stream"][b, 0, :, 0]])
            print(" image", b, "plane 0 rows shift ", [round(float(v), 5) for v in out["shift"][b, 0, :, 0]])
    else:
        print(" stream", [round(float(v), 5) for v in out["stream"].flatten()[:10]])
        print(" shift ", [round(float(v), 5) for v in out["shift"].flatten()[:10]])
