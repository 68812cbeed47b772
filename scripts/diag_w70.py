"""Error levels of the row-shift and general kernels on the knife-edge disparity case (tests/test_gpu_parity.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from cases import rel_err, run_oracle
from gpu_cases import run_product
from planedepth_amd import ops, _capi as C
from planedepth_amd.synthetic import build_case
disps = [0.0, 1.0, 2.0, 1.9999999, 3.0000002, 7.5, 68.9999, 69.0, 75.0, 1e6]
case = build_case(B=2, N=len(disps), H=11, W=70, seed=370, disp_min=0.5, disp_max=9.0, special_disp=disps, sigma_interior=True)
run = dict(target_side="r", automask=True)
ops.DEBUG_STASH = []
fast = run_product(case, run)
ops.SWEEP_IMPL = C.PD_IMPL_GENERAL
slow = run_product(case, run)
want = run_oracle(case, run)
for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma", "g_disp_pp"):
    print(k, "fast %.3e slow %.3e fast-vs-slow %.3e" % (rel_err(fast[k], want[k]), rel_err(slow[k], want[k]), rel_err(fast[k], slow[k])))
d = (slow["g_logits"] - want["g_logits"]).abs()
i = torch.nonzero(d > 0.5 * d.max())
print("largest general-kernel g_logits deviations (b, n, y, x):", i[:8].tolist(), float(d.max()), float(want["g_logits"].abs().max()))
for k in ("g_logits", "g_sigma"):
    print(k)
    for y in (0, 1, 2):
        print("  y=%d" % y, "slow", slow[k][0, 0, y, 66:70].tolist(), "want", want[k][0, 0, y, 66:70].tolist())
for k in ("g_logits", "g_sigma"):
    d = (slow[k] - want[k]).abs()
    v, idx = torch.topk(d.flatten(), 8)
    shp = d.shape
    print(k, [(tuple(int(j) for j in __import__("numpy").unravel_index(int(i), shp)), "%.2e" % float(x)) for i, x in zip(idx, v)])

sf, ss = ops.DEBUG_STASH[0], ops.DEBUG_STASH[1]
print("stash fast", sf[0, :, 1, 69].tolist())
print("stash slow", ss[0, :, 1, 69].tolist())
print("stash max diff", float((sf[:, :4] - ss[:, :4]).abs().max()))
