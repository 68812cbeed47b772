"""stream / shift / general backward against the fp32 and fp64 oracle on one build_case configuration (argv: W H N side d...)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from cases import run_oracle, rel_err
from gpu_cases import run_product
from planedepth_amd import _capi as C, ops
from planedepth_amd.synthetic import build_case

W, H, N, side = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
disps = [float(v) for v in sys.argv[5:]]
kw = dict(special_disp=disps, disp_min=0.5, disp_max=9.0) if disps else dict(disp_min=0.5, disp_max=60.0)
case = build_case(B=2, N=N, H=H, W=W, seed=5000 + W + H, sigma_interior=True, **kw)
run = dict(target_side=side, use_mixture_loss=True, automask=True)
res = {}
for name, impl in (("stream", C.PD_IMPL_AUTO), ("shift", C.PD_IMPL_ROWS1), ("general", C.PD_IMPL_GENERAL)):
    ops.SWEEP_IMPL = impl
    res[name] = run_product(case, run, opt_extra=dict(yz_levels=0, xz_levels=0))
ops.SWEEP_IMPL = C.PD_IMPL_AUTO
res["o32"] = run_oracle(case, run)
res["o64"] = run_oracle(case, run, dtype=torch.float64)
for k in ("g_logits", "g_sigma", "g_disp_pp"):
    print(k, " ".join("%s-vs-%s %.2e" % (a, b, rel_err(res[a][k].double(), res[b][k].double()))
                      for a, b in (("stream", "shift"), ("stream", "general"), ("shift", "general"), ("stream", "o32"), ("shift", "o32"), ("general", "o32"), ("o32", "o64"))))
d = (res["stream"]["g_sigma"] - res["shift"]["g_sigma"]).abs()
idx = torch.nonzero(d == d.max())[0]
print("largest g_sigma diff at", [int(v) for v in idx], "stream", float(res["stream"]["g_sigma"][tuple(idx)]), "shift", float(res["shift"]["g_sigma"][tuple(idx)]),
      "general", float(res["general"]["g_sigma"][tuple(idx)]), "o32", float(res["o32"]["g_sigma"][tuple(idx)]))
