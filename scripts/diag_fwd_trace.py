"""Per-workgroup timeline of the segment-stream forward (VERDICT r4 #1b): a -DPD_DIAGNOSTICS -DPD_FS_TRACE=1 build stamps s_memtime per wave
at entry / after the staging barrier / after the plane loop / at exit, plus HW_ID and XCC_ID; this script launches the
forward at the headline shape, reads the stamps back and reports
  * the histogram of workgroup durations (exit of the last wave - entry of the first), split by how many of the
    workgroup's rows have two live source rows,
  * the phase split (staging, plane loop, finish + stores) of the median workgroup,
  * per CU: which workgroups ran there, in which order, and the idle gaps between them,
  * when the last workgroup of every CU finished relative to the kernel's span (the tail).

    python scripts/diag_fwd_trace.py [--lib trace] [--impl 0|2] [--out profiles/r05_fwd_trace]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from planedepth_amd import _capi as C  # noqa: E402
from planedepth_amd.synthetic import survey_fullsize_case  # noqa: E402

WORDS = 6


def two_row_rows(H):
    """Rows whose y round trip through normalise / un-normalise is inexact in fp32 (the reference's chain, trainer.py:552 +
    grid_sample): those blend two source rows."""
    y = np.arange(H, dtype=np.float32)
    hm1 = np.float32(H - 1)
    g = ((y / hm1) - np.float32(0.5)) * np.float32(2.0)
    iy = ((g + np.float32(1.0)) * np.float32(0.5)) * hm1
    return iy != y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="trace")
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--planes", type=int, default=49)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    path = os.path.join(ROOT, "planedepth_amd", "lib", "libpd_var_%s.so" % args.lib)
    lib = ctypes.CDLL(path)
    for name, (res, a) in C.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, a
    lib.pd_debug_fs_trace.restype = ctypes.c_int
    lib.pd_debug_fs_trace.argtypes = [ctypes.c_void_p, ctypes.c_long]
    c = survey_fullsize_case(B=args.batch, N=args.planes, H=args.height, W=args.width, seed=1234)
    c = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()}
    B, N, H, W = c["logits"].shape
    plane = c["disp_pp"][:, :, 0, 0].contiguous()
    d = C.SweepDesc(B, N, H, W, C.PD_WARP_DISP, C.PD_MIXTURE | C.PD_PH_MEAN_ZEROED, 1.0, args.impl)
    k = lib.pd_sweep_stash_floats(ctypes.byref(d)) // (H * W)
    rgb = torch.empty(B, 3, H, W, device=dev)
    ph = torch.empty(B, 1, H, W, device=dev)
    stash = torch.empty(B, k, H, W, device=dev)
    phm = torch.zeros(1, device=dev)
    st = C.stream_handle(dev)
    flush = torch.empty(96 * 1024 * 1024, device=dev)   # 384 MB: between launches, so that every launch starts cache-cold like in the step

    def fwd():
        rc = lib.pd_plane_sweep_fwd(ctypes.byref(d), C.ptr(c["color_l"]), C.ptr(c["color_r"]), C.ptr(c["logits"]), C.ptr(c["sigma"]),
                                    C.ptr(plane), None, None, None, None, C.ptr(rgb), C.ptr(ph), C.ptr(phm), C.ptr(stash), st)
        assert rc == 0, lib.pd_last_error()

    for _ in range(10):
        fwd()
    rows = 3
    nwg = ((H + rows - 1) // rows) * B
    nwave = rows * ((W + 127) // 128)
    runs = []
    for rep in range(5):
        flush.fill_(float(rep))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fwd(); e1.record()
        torch.cuda.synchronize()
        buf = np.zeros(4096 * 16 * WORDS, dtype=np.uint64)
        assert lib.pd_debug_fs_trace(buf.ctypes.data, buf.size) == 0
        runs.append((e0.elapsed_time(e1), buf.reshape(4096, 16, WORDS)[:nwg, :nwave].copy()))
    ms, t = runs[-1]
    ts = t[..., :4].astype(np.int64)
    t0 = ts[..., 0].min()
    entry = ts[..., 0].min(axis=1) - t0
    staged = ts[..., 1].max(axis=1) - t0
    loopend = ts[..., 2].max(axis=1) - t0
    exit_ = ts[..., 3].max(axis=1) - t0
    span = int(exit_.max())
    tick_ns = ms * 1e6 / span   # ns per s_memtime tick, if the kernel's span is the event time (upper bound: launch overhead inside)
    hw = (t[:, 0, 4] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (t[:, 0, 4] >> np.uint64(32)).astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cu_id = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    y0 = (t[:, 0, 5] & np.uint64(0xffffffff)).astype(np.int64)
    img = (t[:, 0, 5] >> np.uint64(32)).astype(np.int64)
    inexact = two_row_rows(H)
    heavy = np.array([int(inexact[y:y + rows].sum()) for y in y0]) if args.impl != 2 else np.zeros(nwg, dtype=int)
    dur = exit_ - entry
    out = {"event_ms": ms, "span_ticks": span, "ns_per_tick_upper": tick_ns, "workgroups": int(nwg), "waves_per_workgroup": int(nwave),
           "distinct_cus": int(len(set(cu_id.tolist()))), "impl": args.impl}
    lines = []
    P = lambda s: (lines.append(s), print(s))   # noqa: E731
    P("forward launch %.4f ms; %d workgroups x %d waves on %d distinct CUs; span %d ticks (%.3f ns per tick if span = event time)"
      % (ms, nwg, nwave, out["distinct_cus"], span, tick_ns))
    P("")
    P("| two-source-row rows in the workgroup | workgroups | duration ticks: min / median / max | staging | plane loop | finish + stores |")
    P("|---|---|---|---|---|---|")
    by_heavy = {}
    for hcount in sorted(set(heavy.tolist())):
        m = heavy == hcount
        dd = dur[m]
        st_ = np.median(staged[m] - entry[m]); lp = np.median(loopend[m] - staged[m]); fin = np.median(exit_[m] - loopend[m])
        P("| %d | %d | %d / %d / %d | %d | %d | %d |" % (hcount, m.sum(), dd.min(), np.median(dd), dd.max(), st_, lp, fin))
        by_heavy[int(hcount)] = {"workgroups": int(m.sum()), "min": int(dd.min()), "median": float(np.median(dd)), "max": int(dd.max()),
                                 "staging": float(st_), "loop": float(lp), "finish": float(fin)}
    out["by_two_row_rows"] = by_heavy
    P("")
    hist, edges = np.histogram(dur, bins=12)
    P("duration histogram (ticks): " + ", ".join("%d-%d: %d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))
    out["histogram"] = {"edges": edges.tolist(), "counts": hist.tolist()}
    # per CU
    order = np.argsort(entry)
    percu = {}
    for w in order:
        percu.setdefault(int(cu_id[w]), []).append(int(w))
    counts = np.array([len(v) for v in percu.values()])
    P("")
    P("workgroups per CU: " + ", ".join("%d CUs ran %d" % ((counts == n).sum(), n) for n in sorted(set(counts.tolist()))))
    gaps, finish, first_entry, combos, overlap = [], [], [], {}, 0
    for cid, ws in percu.items():
        finish.append(max(exit_[w] for w in ws))
        first_entry.append(entry[ws[0]])
        for a, b in zip(ws[:-1], ws[1:]):
            gaps.append(int(entry[b] - exit_[a]))
            if entry[b] < exit_[a]:
                overlap += 1
        key = tuple(sorted(int(heavy[w]) for w in ws))
        combos[key] = combos.get(key, 0) + 1
    finish = np.array(finish); gaps = np.array(gaps) if gaps else np.zeros(1)
    P("first workgroup of a CU starts at ticks %d..%d; gap between consecutive workgroups on a CU (entry - previous exit): min %d / median %d / max %d (%d pairs overlapped)"
      % (min(first_entry), max(first_entry), gaps.min(), np.median(gaps), gaps.max(), overlap))
    P("last exit per CU, as a fraction of the span: min %.3f / median %.3f / max 1.000; CUs done before 0.90 of the span: %d of %d"
      % (finish.min() / span, np.median(finish) / span, (finish < 0.9 * span).sum(), len(finish)))
    P("sum of two-row rows over the workgroups of a CU -> CUs, mean last-exit fraction:")
    load = {}
    for cid, ws in percu.items():
        load.setdefault(int(sum(heavy[w] for w in ws)), []).append(max(exit_[w] for w in ws) / span)
    for kk in sorted(load):
        P("  %d two-row rows: %d CUs, last exit %.3f of the span on average" % (kk, len(load[kk]), float(np.mean(load[kk]))))
    out["per_cu"] = {"counts": {int(n): int((counts == n).sum()) for n in set(counts.tolist())},
                     "gap_median": float(np.median(gaps)), "gap_max": int(gaps.max()), "finish_min_frac": float(finish.min() / span),
                     "finish_median_frac": float(np.median(finish) / span),
                     "load": {int(kk): {"cus": len(v), "mean_finish_frac": float(np.mean(v))} for kk, v in load.items()}}
    out["event_ms_all_runs"] = [r[0] for r in runs]
    if args.out:
        with open(args.out + ".json", "w") as fh:
            json.dump(out, fh, indent=1)
        with open(args.out + ".md", "w") as fh:
            fh.write("\n".join(lines) + "\n")
        np.savez_compressed(args.out + "_raw.npz", stamps=t, cu=cu_id, heavy=heavy)


if __name__ == "__main__":
    main()
