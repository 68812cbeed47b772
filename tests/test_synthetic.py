"""The benchmark's synthetic decoder outputs against what the reference's own DepthDecoder produced (CPU; no GPU needed)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.mark.parametrize("tag", ["homo3", "disp_xz", "homo_nostereo_l1"])
def test_decoder_plane_geometry_matches_the_reference_decoder(tag):
    """synthetic.decoder_plane_geometry (what `bench.py --warp_type homography_warp --mono_sides --automask --xz_levels 14`
    feeds the path: BASELINE configs[3] as SURVEY.md 8d (4) specifies it) restates networks/depth_decoder.py:146-207.
    tests/golden/trainer_mono.npz holds the outputs of the reference's DepthDecoder itself on two random crops
    (off-centre principal points: tilted ground-plane normals, a different horizon row per sample): with the level
    residuals recovered from its `distance`, every other plane quantity must come out the same."""
    from planedepth_amd.synthetic import decoder_plane_geometry
    z = np.load(os.path.join(GOLDEN, "trainer_mono.npz"))
    g = lambda k: torch.from_numpy(z["%s/%s" % (tag, k)])  # noqa: E731
    meta = json.loads(bytes(z["%s/meta" % tag]).decode())
    nl, nx = meta["no_levels"], meta["xz_levels"]
    grid, dist = g("grid"), g("distance")
    H, W = grid.shape[-2:]
    disp_min, disp_max = 0.5, 0.3 * W                      # make_golden.py: trainer_mono_vectors
    lv = torch.log((0.1 * 0.58 * W / dist[:, :nl]) / disp_max) / np.log(disp_min / disp_max) * (nl - 1)
    gyc = (grid[:, 1, -1, 0] + grid[:, 1, 0, 0]) / 2
    t = ((gyc + 1) * H / 2 - H / 2) / (H * 1.92 * (grid[:, 0, 0, -1] - grid[:, 0, 0, 0]) / 2)
    h = dist[:, nl:] * (1 + t ** 2)[:, None] ** 0.5
    res = torch.cat([lv - torch.arange(nl)[None], (h - 0.1852) / (0.3704 - 0.1852) * (nx - 1) - torch.arange(nx)[None]], 1)
    assert float(res.abs().max()) <= 0.5 + 1e-4               # sigmoid(.) - 0.5
    for rows in (False, True):
        out = decoder_plane_geometry(grid, res, no_levels=nl, xz_levels=nx, disp_min=disp_min, disp_max=disp_max, rows=rows)
        for k in ("disp_layered", "padding_mask", "distance", "norm"):
            want = g(k)
            got = out[k].expand_as(want) if rows and want.dim() == 4 else out[k]
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()), (tag, k, rows)
    assert float((1 - g("padding_mask")).sum()) > 0         # the horizon mask is active in the fixture
