"""The N > 1 path on CPU: two processes over gloo shard pairs round-robin and gather
variable-length matches.  No GPU, no HIP calls."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pats_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_matches(i):
    g = torch.Generator().manual_seed(1000 + i)
    k = (i * 7) % 13                      # includes pairs with zero matches
    return torch.rand((k, 2), generator=g) * 480, torch.rand((k, 2), generator=g) * 640


def _worker(rank, world, port, n_pairs, dst, batch_rows, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.my_pairs(n_pairs, rank, world)
        local = [(i,) + _fake_matches(i) for i in mine]
        allm = shard.gather_matches(local, n_pairs, dst=dst, batch_rows=batch_rows)
        if dst is not None and rank != dst:
            ok = allm is None                                   # only `dst` receives
        else:
            ok = len(allm) == n_pairs
            for i, (ml, mr) in enumerate(allm):
                wl, wr = _fake_matches(i)
                ok = ok and torch.equal(ml, wl) and torch.equal(mr, wr)
        q.put((rank, mine, ok))
    finally:
        dist.destroy_process_group()


# n_pairs = 1: rank 1 owns nothing (the empty-rank case); batch_rows = 5: several bounded rounds
# world 8 (one node of MI355X): 4 001 pairs = configs[3]'s 4 000 + 1 (ranks own 501 / 500: the remainder), and 5 pairs
# (ranks 5..7 own nothing and must still take part in every collective)
@pytest.mark.parametrize("world,n_pairs,dst,batch_rows", [(2, 1, 0, 1 << 20), (2, 7, 0, 5), (2, 8, None, 1 << 20), (2, 7, 1, 3),
                                                         (2, 1, None, 2), (8, 4001, 0, 1 << 20), (8, 5, 0, 4), (8, 13, None, 3)])
def test_shard_and_gather(world, n_pairs, dst, batch_rows):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, dst, batch_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for _, mine, _ in res for i in mine)
    assert owned == list(range(n_pairs))              # every pair exactly once
    assert all(ok for _, _, ok in res)                # the receiver(s) see all matches, in pair order


def test_strong_scaling_split_of_4000_and_4001_pairs():
    """bench.py --total-pairs: every pair belongs to exactly one rank, shares differ by at most one, and the step counts
    cover them (the last step partly filled); a rank without pairs runs no step."""
    for total in (4000, 4001, 5):
        for world in (1, 2, 4, 8):
            shares = [shard.my_pairs(total, r, world) for r in range(world)]
            assert sorted(i for s_ in shares for i in s_) == list(range(total))
            assert max(map(len, shares)) - min(map(len, shares)) <= 1
            for r in range(world):
                st = shard.steps_for(total, r, world, 16)
                assert (st - 1) * 16 < len(shares[r]) <= st * 16 if shares[r] else st == 0
    assert shard.steps_for(4001, 0, 8, 16) == 32 and shard.steps_for(4001, 1, 8, 16) == 32 and shard.steps_for(5, 7, 8, 16) == 0


def test_single_process_gather_needs_no_group():
    local = [(i,) + _fake_matches(i) for i in range(3)]
    out = shard.gather_matches(local, 3)
    assert all(torch.equal(out[i][0], local[i][1]) for i in range(3))
    assert shard.my_pairs(10, 3, 4) == [3, 7]


def test_collective_device_follows_the_backend_not_the_local_tensors(monkeypatch):
    """A rank that owns no pairs has no tensor to take a device from: under nccl (RCCL) it must still
    feed the collective from its HIP device (round-1 bug: it fell back to the CPU there)."""
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert shard.collective_device() == torch.device("cuda", 3)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "cpu:gloo,cuda:nccl")
    assert shard.collective_device() == torch.device("cuda", 3)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")
    assert shard.collective_device() == torch.device("cpu")


def test_bench_plan_only_prints_every_ranks_share_without_a_gpu():
    """`bench.py --gpus 8 --workload yfcc --total-pairs 4000 --plan-only` (BASELINE configs[3], evaluate.py:25-35 sharded over the
    ranks): no GPU is touched; every rank's pairs / steps add up, capacities and the launcher line are printed."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "8", "--workload", "yfcc", "--total-pairs", "4000", "--plan-only"],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["plan_only"] and plan["gpus"] == 8 and plan["scaling"] == "strong" and plan["total_pairs"] == 4000
    assert len(plan["ranks"]) == 8 and sum(x["pairs"] for x in plan["ranks"]) == 4000
    assert all(x["steps"] == (x["pairs"] + plan["pairs_per_step_per_rank"] - 1) // plan["pairs_per_step_per_rank"] for x in plan["ranks"])
    assert plan["coarse_problem"] == "769 x 769" and plan["rows_cap"] > 0 and "--nproc-per-node 8" in plan["launch"]
    assert "--plan-only" not in plan["launch"] and 0 < plan["resident_synthetic_GB_per_rank"] < 288
