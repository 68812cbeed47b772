"""One plane, one shift: the disparity gradient per ROW (dense [B,N,H] form) of the row-stream vs the row-shift backward."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from planedepth_amd import _capi as C, ops

torch.manual_seed(3)
B, N, H, W = 1, 2, 7, 130
dev = "cuda"
cl, cr = torch.rand(B, 3, H, W, device=dev), torch.rand(B, 3, H, W, device=dev)
lg0, sg0 = torch.randn(B, N, H, W, device=dev), 0.05 + 0.9 * torch.rand(B, N, H, W, device=dev)
for side in ("l", "r"):
    for d in (0.0, 1.0, 5.0, 0.25):
        out = {}
        for name, impl in (("stream", C.PD_IMPL_AUTO), ("shift", C.PD_IMPL_ROWS1)):
            ops.SWEEP_IMPL = impl
            lg, sg = lg0.clone().requires_grad_(True), sg0.clone().requires_grad_(True)
            dp = torch.tensor([d, 7.3], device=dev).view(1, N, 1, 1).repeat(B, 1, 1, 1)
            rows = dp.expand(B, N, H, 1).contiguous().requires_grad_(True)       # per-row disparities
            rgb, ph = ops.plane_sweep_disp(cl, cr, lg, sg, rows.expand(B, N, H, W), None, target_side=side, row_uniform=True)
            (ph.mean() + (rgb * 0.01).sum()).backward()
            out[name] = (rows.grad[0, 0, :, 0].cpu(), lg.grad.cpu())
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
        print(side, d, "g_disp rows stream", [round(float(v), 6) for v in out["stream"][0]])
        print(side, d, "g_disp rows shift ", [round(float(v), 6) for v in out["shift"][0]])
        dl = (out["stream"][1] - out["shift"][1]).abs()
        print("   g_logits max diff", float(dl.max()), "at", [int(v) for v in torch.nonzero(dl == dl.max())[0]])
