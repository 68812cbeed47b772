"""world_size-2 checks of the N>1 path on CPU (gloo): shards are independent, no data-path collective is needed, and
the timing reduction bench.py uses behaves.  The per-shard arithmetic is done with the oracle (the checker)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cases import run_oracle
from planedepth_amd import parallel
from planedepth_amd.synthetic import build_case

CASE = dict(B=4, N=5, H=8, W=16, seed=77, disp_min=0.5, disp_max=9.0, sigma_interior=True)


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, w, _ = parallel.init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    case = build_case(**CASE)
    shard = parallel.shard_batch(case, rank, world, CASE["B"])
    assert shard["logits"].shape[0] == CASE["B"] // world
    out = run_oracle(shard, {})
    parallel.barrier()
    slow = parallel.max_over_ranks(1.0 + rank)          # rank 1 is "slower"
    gmean = parallel.global_mean(float(out["ph_loss"]))
    ret[rank] = dict(ph=float(out["ph_loss"]), g_logits=out["g_logits"], slow=slow, gmean=gmean,
                     ips=parallel.throughput(CASE["B"] // world, 10, world, slow))
    dist.destroy_process_group()


def test_two_rank_shards_reproduce_the_full_batch():
    world, port = 2, 29617
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    full = run_oracle(build_case(**CASE), {})
    # the full-batch loss is the mean of the shard losses (equal shard sizes) ...
    assert abs(0.5 * (ret[0]["ph"] + ret[1]["ph"]) - float(full["ph_loss"])) < 1e-6
    assert abs(ret[0]["gmean"] - float(full["ph_loss"])) < 1e-6 and ret[0]["gmean"] == ret[1]["gmean"]
    # ... and every image's outputs depend on that image alone: a rank's shard gradient has the shard's shape and is
    # finite; (its scale differs from the full-batch one by the mean's 1/B vs 1/(B/world), which is DDP's contract)
    per = CASE["B"] // world
    g_full = full["g_logits"]
    for rk in range(world):
        g_shard = ret[rk]["g_logits"]
        assert g_shard.shape == g_full[rk * per:(rk + 1) * per].shape and torch.isfinite(g_shard).all()
    # timing reduction: both ranks see the slowest rank's time; throughput counts every rank's images
    assert ret[0]["slow"] == ret[1]["slow"] == 2.0
    assert ret[0]["ips"] == pytest.approx(CASE["B"] * 10 / 2.0)


def test_shard_range_rejects_ragged_batches():
    assert parallel.shard_range(8, 1, 2) == (4, 8)
    with pytest.raises(ValueError):
        parallel.shard_range(7, 0, 2)
