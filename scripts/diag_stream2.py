"""Row-stream vs row-shift backward vs the oracle on the special-disparity case of test_rowquad_kernels_equal_rowshift_kernels."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from cases import run_oracle
from gpu_cases import run_product
from planedepth_amd import _capi as C, ops
from planedepth_amd.synthetic import build_case

kw = dict(special_disp=[0.0, 0.25, 1.0, 63.0, 64.0, 64.00001, 65.5, 127.99999, 129.0, 200.0], disp_min=0.5, disp_max=9.0)
case = build_case(B=2, N=10, H=7, W=130, seed=4000 + 130, sigma_interior=True, **kw)
run = dict(target_side="l", use_mixture_loss=True, automask=False)
extra = dict(yz_levels=0, xz_levels=0)
res = {}
import itertools
for name, impl in (("stream", C.PD_IMPL_AUTO), ("stream_b", C.PD_IMPL_AUTO), ("shift", C.PD_IMPL_ROWS1), ("general", C.PD_IMPL_GENERAL)):
    ops.SWEEP_IMPL = impl
    res[name] = run_product(case, run, opt_extra=extra)
ops.SWEEP_IMPL = C.PD_IMPL_AUTO
res["oracle32"] = run_oracle(case, run)
res["oracle64"] = run_oracle(case, run, dtype=torch.float64)
for k in ("g_disp_pp",):
    for name in res:
        print(name, [round(float(v), 5) for v in res[name][k].flatten()[:20]])
for k in ("g_logits", "g_sigma"):
    for name in ("stream", "shift", "general", "oracle32"):
        d = (res[name][k].double() - res["oracle64"][k]).abs().max() / res["oracle64"][k].abs().max()
        print(k, name, "vs oracle64", float(d))
