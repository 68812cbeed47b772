"""Diagnostic (GPU box): error of the product path against the oracle in fp32 and fp64, per tensor."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import rel_err, run_oracle  # noqa: E402
from gpu_cases import run_product  # noqa: E402
from planedepth_amd import _capi as C  # noqa: E402
from planedepth_amd import ops  # noqa: E402
from planedepth_amd.synthetic import survey_fullsize_case  # noqa: E402

out = {}
case = survey_fullsize_case(sigma_interior=True)
for tag, run in (("mix", dict()), ("mix_auto", dict(automask=True)), ("l1", dict(use_mixture_loss=False))):
    o32 = run_oracle(case, run)
    o64 = run_oracle(case, run, dtype=torch.float64)
    for impl, name in ((C.PD_IMPL_AUTO, "rowshift"), (C.PD_IMPL_GENERAL, "general")):
        ops.SWEEP_IMPL = impl
        got = run_product(case, run)
        ops.SWEEP_IMPL = C.PD_IMPL_AUTO
        for k in ("rgb_rec", "ph_map", "g_logits", "g_sigma", "g_disp_pp"):
            out["%s/%s/%s" % (tag, name, k)] = dict(vs32=rel_err(got[k], o32[k]), vs64=rel_err(got[k], o64[k].float()),
                                                    o32_vs64=rel_err(o32[k], o64[k].float()))
for k, v in out.items():
    print("%-32s vs32 %.2e  vs64 %.2e  (oracle32 vs 64 %.2e)" % (k, v["vs32"], v["vs64"], v["o32_vs64"]))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag_errors.json"), "w"), indent=1)
