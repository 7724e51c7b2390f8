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


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.my_pairs(n_pairs, rank, world)
        local = [(i,) + _fake_matches(i) for i in mine]
        allm = shard.gather_matches(local, n_pairs)
        ok = len(allm) == n_pairs
        for i, (ml, mr) in enumerate(allm):
            wl, wr = _fake_matches(i)
            ok = ok and torch.equal(ml, wl) and torch.equal(mr, wr)
        q.put((rank, mine, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [1, 7, 8])
def test_two_rank_shard_and_gather(n_pairs):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for _, mine, _ in res for i in mine)
    assert owned == list(range(n_pairs))              # every pair exactly once
    assert all(ok for _, _, ok in res)                # every rank sees all matches, in pair order


def test_single_process_gather_needs_no_group():
    local = [(i,) + _fake_matches(i) for i in range(3)]
    out = shard.gather_matches(local, 3)
    assert all(torch.equal(out[i][0], local[i][1]) for i in range(3))
    assert shard.my_pairs(10, 3, 4) == [3, 7]
