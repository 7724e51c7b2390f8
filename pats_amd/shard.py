"""Pair sharding across GPUs.

Image pairs are independent units (evaluate.py:25-35 walks them at batch 1; `scores_refine_iter`
is re-zeroed per pair at models/pats.py:32), so the path shards with NO data-path collective:
rank r of R owns pairs {i : i mod R == r} and runs the whole hot path for them on its own GPU.
The only exchange is the final collection of matches (`matches_l` / `matches_r`, float32 [K_i, 2]
per pair, variable K_i; the format utils/utils.py:189-213 produces):

  1. one all-gather of a small int64 table (pair index, K_i) per rank - every rank learns every
     count with ONE device->host copy;
  2. the payload travels FLAT: each rank concatenates its pairs' rows (l_row, l_col, r_row, r_col)
     into one [K_rank, 4] float32 tensor and it is gathered to `dst` only (rank 0 evaluates), in
     rounds of at most `batch_rows` rows per rank so the receive buffers stay bounded
     (world x batch_rows x 16 B) however many pairs were matched;
  3. `dst` slices the flat rows back into per-pair views using the table of step 1.

"nccl" is RCCL over xGMI on MI355X ("gloo" in the CPU tests).  KB..MB per pair against
7 x 153 GB/s links: topology-insensitive, so RCCL's default algorithm is used.
"""
import torch
import torch.distributed as dist


def my_pairs(n_pairs, rank, world):
    """Indices of the pairs rank `rank` of `world` processes (round-robin)."""
    return list(range(rank, n_pairs, world))


def steps_for(n_pairs, rank, world, pairs_per_step):
    """Strong scaling (BASELINE configs[3]: 4 000 YFCC pairs over 1 / 2 / 4 / 8 ranks): the number of steps of
    `pairs_per_step` pairs rank `rank` needs for its share, the last one partly filled; 0 for a rank that owns no pair."""
    mine = len(my_pairs(n_pairs, rank, world))
    return (mine + pairs_per_step - 1) // pairs_per_step


def collective_device(group=None):
    """The device collectives of `group` must be fed from: the current HIP device under nccl/RCCL
    (also for a rank that owns no pairs and so has no tensor to take a device from), else the CPU."""
    backend = str(dist.get_backend(group)).lower()
    if "nccl" in backend:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_matches(local, n_pairs, group=None, dst=0, batch_rows=1 << 20, device=None):
    """local: list of (pair_index, matches_l [K,2], matches_r [K,2]) this rank produced (may be empty).
    Returns on rank `dst` a list of n_pairs entries (matches_l, matches_r) in pair order (None for a
    pair nobody reported) and None on every other rank; dst=None delivers the list to EVERY rank
    (all-gather instead of gather).  Works for world_size 1 without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        out = [None] * n_pairs
        for i, ml, mr in local:
            out[i] = (ml, mr)
        return out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device) if device is not None else collective_device(group)
    # 1. the table: one row (pair index, K) per local pair, padded with -1 to the max pairs per rank
    per_rank = max(1, (n_pairs + world - 1) // world)
    if len(local) > per_rank:
        raise RuntimeError("gather_matches: %d local pairs but at most %d per rank for %d pairs on %d ranks"
                           % (len(local), per_rank, n_pairs, world))
    rows = [[i, int(ml.shape[0])] for i, ml, _ in local] + [[-1, 0]] * (per_rank - len(local))
    meta = torch.tensor(rows, dtype=torch.int64).to(dev)
    metas = torch.empty((world, per_rank, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas.view(world * per_rank, 2), meta, group=group)
    table = metas.cpu().tolist()                                # the ONE host read
    totals = [sum(k for i, k in table[r] if i >= 0) for r in range(world)]
    # 2. flat payload, gathered in bounded rounds
    parts = [torch.cat([ml.reshape(-1, 2), mr.reshape(-1, 2)], dim=1).to(device=dev, dtype=torch.float32)
             for _, ml, mr in local if ml.shape[0] > 0]
    flat = torch.cat(parts) if parts else torch.empty((0, 4), dtype=torch.float32, device=dev)
    receiver = dst is None or rank == dst
    got = [[] for _ in range(world)] if receiver else None
    top = max(totals)
    for lo in range(0, top, batch_rows):
        n = min(batch_rows, top - lo)
        send = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        mine = flat[lo:lo + n]
        send[:mine.shape[0]] = mine
        if dst is None:
            recv = torch.empty((world, n, 4), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(recv.view(world * n, 4), send, group=group)
            bufs = list(recv.unbind(0))
        else:
            bufs = [torch.empty((n, 4), dtype=torch.float32, device=dev) for _ in range(world)] if receiver else None
            dist.gather(send, gather_list=bufs, dst=dist.get_global_rank(group, dst) if group is not None else dst,
                        group=group)
        if receiver:
            for r in range(world):
                keep = min(n, max(0, totals[r] - lo))
                if keep:
                    got[r].append(bufs[r][:keep])
    if not receiver:
        return None
    # 3. slice the flat rows back into pairs
    out = [None] * n_pairs
    for r in range(world):
        rows_r = torch.cat(got[r]) if got[r] else torch.empty((0, 4), dtype=torch.float32, device=dev)
        off = 0
        for i, k in table[r]:
            if i < 0:
                continue
            out[i] = (rows_r[off:off + k, 0:2], rows_r[off:off + k, 2:4])
            off += k
    return out
