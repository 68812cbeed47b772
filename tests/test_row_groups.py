"""The segment-stream forward's row groups (pd_plane_sweep_fwdstream.hip: fwdstream_rows): host-side table only, no GPU needed.
Every target row must be served exactly once whatever the height, and a row that blends two source rows
(trainer.py:540-554 + F.grid_sample's un-normalisation: the vertical round trip is inexact for a quarter of the rows) should sit
in the group of the neighbour it blends in."""
import ctypes

import numpy as np
import pytest

from planedepth_amd import _capi as C


def _groups(H, rows, eps=0.0):
    lib = C.load()
    fn = lib.pd_debug_fwd_row_groups
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                   ctypes.POINTER(ctypes.c_int)]
    G = (H + rows - 1) // rows
    out = np.full(max(G * rows, 1), 0xFFFF, dtype=np.uint16)
    links, cut = ctypes.c_int(0), ctypes.c_int(0)
    n = fn(H, rows, eps, out.ctypes.data, ctypes.byref(links), ctypes.byref(cut))
    return n, out, links.value, cut.value


@pytest.mark.parametrize("rows", [2, 3])
@pytest.mark.parametrize("H", [1, 2, 3, 5, 24, 47, 96, 191, 192, 193, 200, 213])
def test_every_row_exactly_once(H, rows):
    n, out, links, cut = _groups(H, rows)
    assert n in (0, ((H + rows - 1) // rows) * rows)
    served = out[out < H]
    assert sorted(served.tolist()) == list(range(H))
    G = (H + rows - 1) // rows
    per_group = [(out[g * rows:(g + 1) * rows] < H).sum() for g in range(G)]
    assert all(k == rows for k in per_group[:-1]) and per_group[-1] == H - (G - 1) * rows   # same group sizes as consecutive rows
    assert 0 <= cut <= links


def test_kitti_heights_keep_linked_rows_together():
    # H = 192 (BASELINE configs[1]): 48 rows blend two source rows; consecutive groups of three cut 15 of those pairs
    n, out, links, cut = _groups(192, 3)
    assert n == 192 and links == 48 and cut <= 2
    n, out, links, cut = _groups(384, 3)
    assert n == 384 and links > 0 and cut <= links // 8
    # a threshold above every second-row weight (PD_IMPL_FAST_ROWS): nothing is linked, consecutive rows
    n, out, links, cut = _groups(192, 3, 2.0 ** -16)
    assert n == 0 and links == 0 and out[:192].tolist() == list(range(192))


def test_heights_beyond_the_table_keep_consecutive_rows():
    n, out, links, cut = _groups(642, 3)
    assert n == 0

