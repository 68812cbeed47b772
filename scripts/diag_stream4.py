"""Per-PLANE disparities: g_disp of plane 0 (d = 0, side l) as the other planes' disparities vary."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from planedepth_amd import _capi as C, ops

torch.manual_seed(3)
H, W = 7, 130
dev = "cuda"
for B, disps in ((1, [0.0, 7.3]), (1, [0.0, 0.25]), (1, [0.0, 200.0]), (1, [0.0, 129.0]), (1, [0.0, 64.0]), (2, [0.0, 7.3]),
                 (1, [0.0, 0.25, 1.0, 63.0, 64.0, 64.00001, 65.5, 127.99999, 129.0, 200.0])):
    N = len(disps)
    cl, cr = torch.rand(B, 3, H, W, device=dev), torch.rand(B, 3, H, W, device=dev)
    lg0, sg0 = torch.randn(B, N, H, W, device=dev), 0.05 + 0.9 * torch.rand(B, N, H, W, device=dev)
    out = {}
    for name, impl in (("stream", C.PD_IMPL_AUTO), ("shift", C.PD_IMPL_ROWS1)):
        ops.SWEEP_IMPL = impl
        lg, sg = lg0.clone().requires_grad_(True), sg0.clone().requires_grad_(True)
        dp = torch.tensor(disps, device=dev).view(1, N, 1, 1).repeat(B, 1, 1, 1).requires_grad_(True)
        rgb, ph = ops.plane_sweep_disp(cl, cr, lg, sg, dp.expand(B, N, H, W), None, target_side="l")
        (ph.mean() + (rgb * 0.01).sum()).backward()
        out[name] = dp.grad.flatten().cpu()
    ops.SWEEP_IMPL = C.PD_IMPL_AUTO
    print(B, disps[:4], "stream", [round(float(v), 6) for v in out["stream"][:4]], "shift", [round(float(v), 6) for v in out["shift"][:4]])
