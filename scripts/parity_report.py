"""Measured parity of the product path per configuration and tensor -> markdown (profiles/r03_parity.md).

Columns: normalised max error (max|a-b| / max|b|) of the product against the reference-captured fixture (or, where no
fixture exists at that size, the oracle's fp32 evaluation = the reference's arithmetic), against the oracle's fp64
evaluation of the same formulas, the reference-fp32-vs-fp64 distance for scale, and the share of elements beyond the
element-wise allowance 1e-4*|b| + 1e-4*max|b| (against the fp32 reference).  Needs a GPU; run from the repo root."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from cases import (SMALL, elementwise_report, load_fixture, load_trainer_fixture, rel_err, run_oracle,  # noqa: E402
                   run_oracle_trainer)
from gpu_cases import run_product, run_product_trainer  # noqa: E402
from planedepth_amd.synthetic import survey_fullsize_case  # noqa: E402

KEYS = ("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp", "g_distance", "g_disp_layered", "total_loss",
        "smooth_loss", "g_dists")
rows = []


def add(config, source, got, ref32, exact):
    for k in got:
        if not (k in KEYS or k.startswith("rgb_rec_") or k.startswith("g_Rt")) or k not in ref32:
            continue
        w = ref32[k]
        g = got[k]
        if k == "g_disp_layered":   # per-row hand-over of the row-shift kernels: compare what the decoder's expand sums
            g, w = g.sum(-1), w.sum(-1)
        if float(w.abs().max()) == 0.0:
            continue
        ex = exact.get(k) if exact else None
        if ex is not None and k == "g_disp_layered":
            ex = ex.sum(-1)
        rep = elementwise_report(g, w)
        rows.append((config, source, k, rel_err(g, w), rel_err(g, ex.float()) if ex is not None else None,
                     rel_err(w, ex.float()) if ex is not None else None, rep["frac_beyond"]))


for name in SMALL:
    case, want, run = load_fixture(name)
    got = run_product(case, run)
    homo = run.get("warp_type") == "homography_warp"
    add("fixture %s" % name, "reference-captured", got, want, run_oracle(case, run, dtype=torch.float64) if homo else None)

for tag in ("homo3", "homo_nostereo_l1", "disp_xz"):
    z, meta = load_trainer_fixture(tag)
    for const in ((False, True) if tag == "homo3" else (False,)):
        got = run_product_trainer(z, meta, stereo_constant=const)
        add("trainer_mono %s%s" % (tag, " (stereo pose constant: row-shift view)" if const else ""), "reference-captured",
            got, z, run_oracle_trainer(z, meta, dtype=torch.float64))

# the strict route (PD_TORCH_HOMOGRAPHY: the reference's fp32 torch.inverse chain) against the same reference-captured fixtures
from planedepth_amd import ops  # noqa: E402
ops.TORCH_HOMOGRAPHY = True
try:
    for tag in ("homo3", "homo_nostereo_l1"):
        z, meta = load_trainer_fixture(tag)
        got = run_product_trainer(z, meta, stereo_constant=False)
        add("trainer_mono %s, strict route (PD_TORCH_HOMOGRAPHY=1)" % tag, "reference-captured", got, z,
            run_oracle_trainer(z, meta, dtype=torch.float64))
finally:
    ops.TORCH_HOMOGRAPHY = False

for label, kw, run, extra in (("192x640x49 mixture", {}, {}, None),
                              ("192x640x49 mixture automask", {}, dict(automask=True), None),
                              ("192x640x49 L1", {}, dict(use_mixture_loss=False), None),
                              ("192x640x(49+14) automask", dict(N=63, n_xz=14), dict(automask=True), dict(yz_levels=0, xz_levels=14)),
                              ("192x640x49 homography_warp stereo", {}, dict(warp_type="homography_warp"), None)):
    case = survey_fullsize_case(sigma_interior=True, **kw)
    got = run_product(case, run, opt_extra=extra)
    add("full size %s" % label, "oracle fp32", got, run_oracle(case, run), run_oracle(case, run, dtype=torch.float64))

# one homography per PLANE (a pose with rotation and translation, --use_colmap): general forward + the two-pass gather backward
from planedepth_amd.synthetic import small_pose  # noqa: E402
case = survey_fullsize_case(sigma_interior=True)
case["Rt"] = small_pose(torch.Generator().manual_seed(99), 1)
run = dict(warp_type="homography_warp")
add("full size 192x640x49 homography_warp 6-DoF pose (gather backward)", "oracle fp32", run_product(case, run),
    run_oracle(case, run), run_oracle(case, run, dtype=torch.float64))

fmt = lambda v: "—" if v is None else "%.1e" % v  # noqa: E731
out = ["| configuration | compared with | tensor | vs reference fp32 | vs fp64 | reference fp32 vs fp64 | elements beyond 1e-4·\\|b\\| + 1e-4·max\\|b\\| |",
       "|---|---|---|---|---|---|---|"]
for r in rows:
    out.append("| %s | %s | %s | %s | %s | %s | %.2e |" % (r[0], r[1], r[2], fmt(r[3]), fmt(r[4]), fmt(r[5]), r[6]))
path = os.path.join(ROOT, "gpurun_out", "r3", "r03_parity.md")
os.makedirs(os.path.dirname(path), exist_ok=True)
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
print("... %d rows -> %s" % (len(rows), path))
