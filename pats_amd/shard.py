"""Pair sharding across GPUs.

Image pairs are independent units (evaluate.py:25-35 walks them at batch 1; `scores_refine_iter`
is re-zeroed per pair at models/pats.py:32), so the path shards with NO data-path collective:
rank r of R owns pairs {i : i mod R == r} and runs the whole hot path for them on its own GPU.
The only exchange is the final collection of matches (`matches_l` / `matches_r`, float32 [K_i, 2]
per pair, variable K_i): one all-gather of the counts, one padded all-gather of the payload, over
whatever backend the process group uses ("nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU
tests).  KB..MB per pair against 7 x 153 GB/s links: topology-insensitive, so RCCL's default
algorithm is used.
"""
import torch
import torch.distributed as dist


def my_pairs(n_pairs, rank, world):
    """Indices of the pairs rank `rank` of `world` processes (round-robin)."""
    return list(range(rank, n_pairs, world))


def gather_matches(local, n_pairs, group=None):
    """local: list of (pair_index, matches_l [K,2], matches_r [K,2]) this rank produced.
    Returns on EVERY rank a list of n_pairs entries (matches_l, matches_r) in pair order.
    Works for world_size 1 without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        out = [None] * n_pairs
        for i, ml, mr in local:
            out[i] = (ml, mr)
        return out
    world = dist.get_world_size(group)
    dev = local[0][1].device if local else torch.device("cpu")
    # 1. counts: one row (pair index, K) per local pair, padded to the max pairs per rank
    per_rank = (n_pairs + world - 1) // world
    meta = torch.full((per_rank, 2), -1, dtype=torch.int64, device=dev)
    for j, (i, ml, mr) in enumerate(local):
        meta[j, 0], meta[j, 1] = i, ml.shape[0]
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    kmax = int(max(int(m[:, 1].max().item()) for m in metas))
    kmax = max(kmax, 1)
    # 2. payload: [per_rank, kmax, 4] = (l_row, l_col, r_row, r_col), zero padded
    pay = torch.zeros((per_rank, kmax, 4), dtype=torch.float32, device=dev)
    for j, (i, ml, mr) in enumerate(local):
        k = ml.shape[0]
        pay[j, :k, 0:2] = ml
        pay[j, :k, 2:4] = mr
    pays = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(pays, pay, group=group)
    out = [None] * n_pairs
    for r in range(world):
        for j in range(per_rank):
            i, k = int(metas[r][j, 0].item()), int(metas[r][j, 1].item())
            if i >= 0:
                out[i] = (pays[r][j, :k, 0:2].clone(), pays[r][j, :k, 2:4].clone())
    return out
