"""Run by tests/test_gpu_parity.py::test_step_replayed_from_a_hip_graph_equals_the_eager_step in a process of its own (a capture
that goes wrong takes the process down inside hipStreamEndCapture — the suite must survive that and report it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    from planedepth_amd import ops
    from planedepth_amd.synthetic import build_case
    case = build_case(B=2, N=9, H=24, W=200, seed=654, disp_min=0.5, disp_max=40.0, sigma_interior=True)
    c = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.items()}
    lg, sg = c["logits"].clone().requires_grad_(True), c["sigma"].clone().requires_grad_(True)
    dp = c["disp_pp"].clone().requires_grad_(True)
    one = torch.ones((), device="cuda")
    keep = {}

    def step():
        lg.grad = sg.grad = dp.grad = None
        rgb, ph, ph_mean = ops.plane_sweep_disp(c["color_l"], c["color_r"], lg, sg, dp.expand(-1, -1, 24, 200), None,
                                                automask=True, return_mean=True)
        torch.autograd.backward([ph_mean, rgb], [one, c["g_rgb_rec"]])
        keep.update(rgb=rgb, ph=ph, ph_mean=ph_mean, g_l=lg.grad, g_s=sg.grad, g_d=dp.grad)

    # capture FIRST, before any eager step on the default stream (autograd's AccumulateGrad nodes remember the stream they were
    # created on: a capture after eager steps makes the engine wait on a non-capturing stream and hipStreamEndCapture crashes —
    # bench.py's capture_step has the same order)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):   # (the warm-up torch documents for captures with a backward)
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    got = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        got.append({k: v.detach().clone() for k, v in keep.items()})
    step()   # eager, on the default stream
    torch.cuda.synchronize()
    want = {k: v.detach().clone() for k, v in keep.items()}
    assert float(want["g_d"].abs().max()) > 0 and float(want["ph_mean"]) > 0
    for g_ in got:
        for k, w in want.items():
            tol = 1e-5 if k in ("g_d", "ph_mean") else 0.0   # (sums by float atomics: the order varies)
            assert rel_err(g_[k].cpu(), w.cpu()) <= tol, k
    print("graph replay equals eager: ok")


if __name__ == "__main__":
    main()
