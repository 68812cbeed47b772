"""Measured parity of the product path per configuration and tensor -> markdown (profiles/r05_parity.md).

Part 1 — the disp path under the ROW MODES (VERDICT r4 #1a): normalised max error (max|a-b| / max|b|) of every output tensor
and gradient against the reference-captured fixtures (small cases) or the oracle's fp32 evaluation = the reference's
arithmetic (full size), once per mode: PD_IMPL_EXACT_ROWS (every second source row served), the thresholds given with
--eps (a second row below that weight dropped; PD_ROW_EPS is read once per process, so the script re-executes itself per
value), PD_IMPL_FAST_ROWS (2^-16).  The bar is BASELINE.json's 1e-4.
Part 2 — homography_warp (three routes) and the other fixtures with the library's default mode: vs the reference fp32, vs
the fp64 evaluation of the same formulas, the reference-fp32-vs-fp64 distance for scale, and the share of elements beyond
the element-wise allowance 1e-4*|b| + 1e-4*max|b|.  Needs a GPU; run from the repo root:

    python scripts/parity_report.py [--eps 4e-6,2.9e-6] [--out profiles/r05_parity.md]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

KEYS = ("rgb_rec", "ph_map", "ph_loss", "g_logits", "g_sigma", "g_disp_pp", "g_distance", "g_disp_layered", "total_loss",
        "smooth_loss", "g_dists")
XY_ONLY = ["disp_mix_r", "disp_mix_l", "disp_mix_automask", "disp_l1", "disp_l1_automask", "disp_mix_oob",
           "disp_mix_integer_d", "disp_mix_masknovel", "disp_l1_masknovel"]


def band_limited(shape, gen, cutoff=0.12, lo=0.0, hi=1.0):
    x = torch.randn(shape, generator=gen)
    H, W = shape[-2:]
    fy = torch.fft.fftfreq(H).abs()[:, None] * 2.0
    fx = torch.fft.rfftfreq(W).abs()[None, :] * 2.0
    x = torch.fft.irfft2(torch.fft.rfft2(x) * ((fy <= cutoff) & (fx <= cutoff)).to(x.dtype), s=(H, W))
    flat = x.flatten(-2)
    mn, mx = flat.min(-1)[0][..., None, None], flat.max(-1)[0][..., None, None]
    return lo + (hi - lo) * (x - mn) / (mx - mn)


def disp_cases():
    """(label, case, run, opt_extra, reference tensors or None -> oracle fp32)"""
    from cases import SMALL, load_fixture
    from planedepth_amd.synthetic import survey_fullsize_case
    for name in SMALL:
        case, want, run = load_fixture(name)
        if run.get("warp_type", "disp_warp") != "disp_warp":
            continue
        yield "fixture %s (row-shift kernels)" % name, case, run, None, want
        if name in XY_ONLY:
            yield "fixture %s (headline kernels)" % name, case, run, dict(xz_levels=0, yz_levels=0), want
    hk = dict(xz_levels=0, yz_levels=0)
    full = survey_fullsize_case(sigma_interior=True)
    yield "192x640x49 B=1 mixture [configs 0/1]", full, {}, hk, None
    yield "192x640x49 B=1 mixture automask", full, dict(automask=True), hk, None
    yield "192x640x49 B=1 L1", full, dict(use_mixture_loss=False), hk, None
    yield ("192x640x(49+14 xz) automask [configs 3 planes]", survey_fullsize_case(sigma_interior=True, N=63, n_xz=14),
           dict(automask=True), dict(yz_levels=0, xz_levels=14), None)
    yield "192x640x49 B=12 [configs 2]", survey_fullsize_case(sigma_interior=True, B=12), {}, hk, None
    yield "384x1280x49 B=1 [configs 4]", survey_fullsize_case(sigma_interior=True, H=384, W=1280), {}, hk, None
    g = torch.Generator().manual_seed(99)
    bl = survey_fullsize_case(sigma_interior=True)
    B, N, H, W = bl["logits"].shape
    bl["color_l"], bl["color_r"] = band_limited((B, 3, H, W), g), band_limited((B, 3, H, W), g)
    bl["logits"], bl["sigma"] = band_limited((B, N, H, W), g, lo=-3.0, hi=3.0), band_limited((B, N, H, W), g, lo=0.02, hi=0.9)
    yield "192x640x49 B=1 band-limited inputs (SURVEY H2)", bl, {}, hk, None
    yield "192x640x49 B=1 band-limited inputs, automask", bl, dict(automask=True), hk, None


def run_mode(impl):
    """{label: {tensor: rel_err}} of the disp path with ops.SWEEP_IMPL = impl."""
    from cases import rel_err, run_oracle
    from gpu_cases import run_product
    from planedepth_amd import ops
    out = {}
    oracle_cache = {}
    for label, case, run, extra, want in disp_cases():
        ops.SWEEP_IMPL = impl
        try:
            got = run_product(case, run, opt_extra=extra)
        finally:
            ops.SWEEP_IMPL = 0
        if want is None:
            key = (id(case), json.dumps(run, sort_keys=True))
            if key not in oracle_cache:
                oracle_cache[key] = run_oracle(case, run)
            want = oracle_cache[key]
        errs = {}
        for k in KEYS:
            if k in got and k in want and float(want[k].abs().max()) > 0.0:
                errs[k] = rel_err(got[k], want[k])
        out[label] = errs
        print(label, {k: "%.1e" % v for k, v in errs.items()}, flush=True)
    return out


def part2(rows):
    from cases import SMALL, elementwise_report, load_fixture, load_trainer_fixture, rel_err, run_oracle, run_oracle_trainer
    from gpu_cases import run_product, run_product_trainer
    from planedepth_amd import ops
    from planedepth_amd.synthetic import small_pose, survey_fullsize_case

    def add(config, source, got, ref32, exact):
        for k in got:
            if not (k in KEYS or k.startswith("rgb_rec_") or k.startswith("g_Rt")) or k not in ref32:
                continue
            w, g = ref32[k], got[k]
            if k == "g_disp_layered":   # per-row hand-over of the row kernels: compare what the decoder's expand sums
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
        if run.get("warp_type", "disp_warp") == "disp_warp":
            continue
        add("fixture %s" % name, "reference-captured", run_product(case, run), want, run_oracle(case, run, dtype=torch.float64))
    for tag in ("homo3", "homo_nostereo_l1", "disp_xz"):
        z, meta = load_trainer_fixture(tag)
        for const in ((False, True) if tag == "homo3" else (False,)):
            got = run_product_trainer(z, meta, stereo_constant=const)
            add("trainer_mono %s%s" % (tag, " (stereo pose constant: row kernels)" if const else ""), "reference-captured",
                got, z, run_oracle_trainer(z, meta, dtype=torch.float64))
    ops.TORCH_HOMOGRAPHY = True   # the strict route: the reference's fp32 torch.inverse chain
    try:
        for tag in ("homo3", "homo_nostereo_l1"):
            z, meta = load_trainer_fixture(tag)
            add("trainer_mono %s, strict route (PD_TORCH_HOMOGRAPHY=1)" % tag, "reference-captured",
                run_product_trainer(z, meta, stereo_constant=False), z, run_oracle_trainer(z, meta, dtype=torch.float64))
    finally:
        ops.TORCH_HOMOGRAPHY = False
    case = survey_fullsize_case(sigma_interior=True)
    run = dict(warp_type="homography_warp")
    add("full size 192x640x49 homography_warp stereo", "oracle fp32", run_product(case, run), run_oracle(case, run),
        run_oracle(case, run, dtype=torch.float64))
    case = survey_fullsize_case(sigma_interior=True)
    case["Rt"] = small_pose(torch.Generator().manual_seed(99), 1)
    add("full size 192x640x49 homography_warp 6-DoF pose (gather backward)", "oracle fp32", run_product(case, run),
        run_oracle(case, run), run_oracle(case, run, dtype=torch.float64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eps", default="", help="comma list of PD_ROW_EPS thresholds to measure besides exact / fast")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r5", "r05_parity.md"))
    ap.add_argument("--worker", default="", help="internal: mode to run, JSON to stdout's last line")
    ap.add_argument("--no_part2", action="store_true")
    args = ap.parse_args()
    from planedepth_amd import _capi as C
    if args.worker:
        impl = {"exact": C.PD_IMPL_EXACT_ROWS, "fast": C.PD_IMPL_FAST_ROWS}.get(args.worker, C.PD_IMPL_AUTO)
        print("RESULT " + json.dumps(run_mode(impl)))
        return
    modes = [("exact rows", "exact", None)] + [("second rows below %s dropped" % e, "auto", e) for e in args.eps.split(",") if e] + \
            [("fast rows (2^-16)", "fast", None)]
    results = []
    for title, worker, eps in modes:
        env = dict(os.environ)
        if eps:
            env["PD_ROW_EPS"] = eps
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", worker], env=env, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            raise SystemExit("mode %s failed:\n%s" % (title, out.stderr[-3000:]))
        results.append((title, json.loads(line[-1][7:])))
    fmt = lambda v: "—" if v is None else ("**%.1e**" % v if v >= 1e-4 else "%.1e" % v)  # noqa: E731
    out = ["# Parity of the product path, round 5 (`scripts/parity_report.py`, one MI355X)", "",
           "## 1. disp path under the row modes: max|a-b| / max|b| per tensor (bold: beyond BASELINE's 1e-4)", "",
           "| configuration | tensor | " + " | ".join(t for t, _ in results) + " |", "|---|---|" + "---|" * len(results)]
    worst = {t: (0.0, None) for t, _ in results}
    for label in results[0][1]:
        for k in results[0][1][label]:
            vals = [r[label].get(k) for _, r in results]
            out.append("| %s | %s | %s |" % (label, k, " | ".join(fmt(v) for v in vals)))
            for (t, _), v in zip(results, vals):
                if v is not None and v > worst[t][0]:
                    worst[t] = (v, "%s / %s" % (label, k))
    out += ["", "Worst tensor per mode: " + "; ".join("%s: %.2e (%s)" % (t, w[0], w[1]) for t, w in worst.items()), ""]
    if not args.no_part2:
        rows = []
        part2(rows)
        f2 = lambda v: "—" if v is None else "%.1e" % v  # noqa: E731
        out += ["## 2. homography_warp routes and trainer-shaped fixtures (library default)", "",
                "| configuration | compared with | tensor | vs reference fp32 | vs fp64 | reference fp32 vs fp64 | elements beyond 1e-4·\\|b\\| + 1e-4·max\\|b\\| |",
                "|---|---|---|---|---|---|---|"]
        for r in rows:
            out.append("| %s | %s | %s | %s | %s | %s | %.2e |" % (r[0], r[1], r[2], f2(r[3]), f2(r[4]), f2(r[5]), r[6]))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    open(args.out, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:8]))
    print("... -> %s" % args.out)


if __name__ == "__main__":
    main()
