"""Where PD_IMPL_FAST_ROWS differs from PD_IMPL_EXACT_ROWS on the 49 + 14 xz-plane full-size case (g_logits 1.5e-2 in
profiles/r05_parity.md): largest differences per tensor with their (image, plane, row, column)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from gpu_cases import run_product  # noqa: E402
from planedepth_amd import _capi as C, ops  # noqa: E402
from planedepth_amd.synthetic import survey_fullsize_case  # noqa: E402

case = survey_fullsize_case(sigma_interior=True, N=63, n_xz=14)
res = {}
for name, impl in (("exact", C.PD_IMPL_EXACT_ROWS), ("fast", C.PD_IMPL_FAST_ROWS)):
    ops.SWEEP_IMPL = impl
    res[name] = run_product(case, dict(automask=True), opt_extra=dict(yz_levels=0, xz_levels=14))
    ops.SWEEP_IMPL = 0
for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma"):
    a, b = res["fast"][k], res["exact"][k]
    d = (a - b).abs()
    top = torch.topk(d.flatten(), 8)
    print(k, "max|exact| %.3e" % float(b.abs().max()), "max diff %.3e" % float(d.max()))
    for v, i in zip(top.values.tolist(), top.indices.tolist()):
        idx = tuple(int(x) for x in torch.unravel_index(torch.tensor(i), d.shape))
        print("   ", idx, "diff %.3e exact %.3e fast %.3e" % (v, float(b[idx]), float(a[idx])))
