"""The control flow of PATS.forward (models/pats.py:18-85) and of the three layers' forward tails
(first_layer.py:110-157, second_layer.py:100-124, third_layer.py:153-170) on the device-side ops
of pats_amd, with what comes before them abstracted as three callbacks: the backbones (ResNet / FPN - out of scope,
DESIGN.md section 7) followed by the heads, which pats_amd.heads provides (CoarseHeads / FineHeads / ThirdHeads).

What changes against the reference's Python, and why it is equivalent:
  * every OT / expansion / gather / merge / result step is one C-ABI call (pats_amd.ops);
  * the chunk loop reads the host ONCE up front (the cumulative match counts: chunk plan and crop
    counts) and then once per chunk for the third-level problem count P and once for the match count M;
    the reference syncs at every boolean-mask indexing;
  * rows of an L2 chunk that are left with no cell are NOT deleted (pats.py:40-52 compacts patches,
    features and points): they emit nothing in get_result, and the third-level callback receives
    `b_ids` into the un-compacted rows.  Same matches, same order
    (tests/test_gpu_parity.py::test_pipeline_chain compares with the reference's own functions run in
    its own order on the same callbacks).

Callbacks (`nets`), all returning float32 GPU tensors:
  nets.coarse(left, right) -> mdesc0 [1,D,N], mdesc1 [1,D,N], scale [1,1,N], alpha (0-d / float)
      N = (H/32)*(W/32); what FirstLayer computes up to first_layer.py:107.
  nets.fine(num, new_left, new_right, chunk_mask) -> mdesc0 [B,264,145], mdesc1 [B,264,145],
      scale_x [B,1,144], scale_y [B,1,144]; what SecondLayer computes up to second_layer.py:97.
  nets.third(num, mkpts0_c [P,2], mkpts1_c [P,2], b_ids [P]) -> feat0 [P,128,65], feat1 [P,128,65],
      scale [P,1,64]; what ThirdLayer computes up to third_layer.py:152 (mkpts*_c as PATS.forward
      passes them, pats.py:57-58).
"""
import torch

from . import ops


def _round4(x, clamp96):
    """third_layer.py:122 / :126-128: round(x / 4).long() * 4 (targets clamped to [0, 96] first)."""
    if clamp96:
        x = torch.where(x >= 96, torch.tensor(96.0, device=x.device), x)
        x = torch.where(x <= 0, torch.tensor(0.0, device=x.device), x)
    return torch.round(x / 4.0).long() * 4


def forward_path(left, right, nets, if_local=True, if_outdoor=True, merge_new=True, iters=100, batch_chunks=False,
                 device_counts=False, streams=1):
    """left / right: [1,H,W,3] float32 HWC images (what first_layer.py:128-129 permutes to).
    Returns {"matches_l": [M,2], "matches_r": [M,2]} in the reference's (row, col) pixel convention and
    order, plus "chunks": per-chunk (B, P, M) for inspection.

    device_counts=True (with batch_chunks=False): the same chunk-by-chunk walk with the counts the reference reads back inside
    the loop (P per chunk, M per chunk, the boolean-mask sizes) left on the device - see forward_chunks_device below: two host
    reads per pair (the chunk plan up front, the match counts at the end).  `streams` > 1 walks consecutive chunks on
    different HIP streams (the merges stay in chunk order).

    batch_chunks=True runs all chunks of the pair together (the reference walks them one by one to bound
    memory on a 16-40 GB card): ONE fine-level cost+OT+expansion launch over the concatenated chunk rows,
    the merges in chunk order (their only coupling is scores_back, pats.py:32,37), ONE third-level
    launch, ONE get_result with the chunks as its batch dimension; three host reads per pair.  The
    callbacks are then called once with num=None and the list of per-chunk row counts:
        nets.fine(None, new_left_all, new_right_all, masks [C,N], sizes=[B_0, ...])
        nets.third(None, mkpts0_c, mkpts1_c, b_ids (rows of the concatenation), sizes=[B_0, ...])
    Same matches in the same order (tests/test_gpu_parity.py::test_pipeline_chain runs both modes)."""
    if device_counts and not batch_chunks:
        with ops.workspace_cache():
            return forward_chunks_device(left, right, nets, if_local, if_outdoor, merge_new, iters, streams)
    dev = left.device
    H, W = int(left.shape[1]), int(left.shape[2])
    h, w = H // 32, W // 32
    empty = torch.zeros([0, 2], device=dev)
    # ---- first layer tail (first_layer.py:110-146) ------------------------------------------------
    mdesc0, mdesc1, scale, alpha = nets.coarse(left, right)
    scores = ops.cost_ot(mdesc0, mdesc1, 1, alpha, scale, iters)
    scales, cflag = ops.colmass_sqrt(scores, return_flags=True)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_first(scores, scales, (H, W), 32, col_nomatch=cflag)
    sum_cycle = torch.cumsum(torch.logical_not(ifn1).int(), dim=1)
    sc_host = sum_cycle.to("cpu").numpy()                       # host read 1: chunk plan + crop counts
    if int(sc_host[0, -1]) <= 0:                                # pats.py:27-31
        return {"matches_l": empty, "matches_r": empty, "chunks": []}
    cycle_num, second_set, third_set = ops.split_patches(sc_host[0], h, w, 2 * w if if_local else 512)
    K = int(sc_host[0, -1])
    # one gather for the pair: chunk (lo, hi] is rows [lo, min(hi, K)) of it (the cumsum is monotone)
    new_left, new_right, xsn, ysn, avn = ops.Compute_imgs(xs, ys, pts, ifn1, left, right, width=w, height=h,
                                                          known_count=K)
    scores_back = torch.zeros([1, h * w, 16, 9], dtype=torch.float64, device=dev)     # pats.py:32
    merge = ops.merge_patches_new if merge_new else ops.merge_patches_old
    if batch_chunks:
        return _forward_batched(left, nets, if_outdoor, iters, merge, scores_back, ifn1, sum_cycle, second_set,
                                third_set, K, new_left, new_right, xsn, avn, h, w, H, W)
    out_l, out_r, info = [], [], []
    for num in range(cycle_num):
        lo, hi = second_set[num]
        hi_c = min(hi, K)
        B = hi_c - lo
        if B <= 0:
            continue
        mask = torch.logical_or(ifn1, torch.logical_or(sum_cycle <= lo, sum_cycle > hi))    # first_layer.py:137-138
        # ---- second layer tail (second_layer.py:100-124) --------------------------------------------
        f0, f1, sx, sy = nets.fine(num, new_left[lo:hi_c], new_right[lo:hi_c], mask)[:4]
        Z2, cflag2 = ops.cost_ot(f0, f1, 2, 1.0, (sx * sy).contiguous(), iters, bias_k=2.0 if if_outdoor else 3.0,
                                 return_flags=True)
        trust2, pts2, _, _, ifn_L2, _ = ops.est_position_second(Z2, sx, sy, [96, 96], 8, col_nomatch=cflag2)
        ifn_L2, scores_back = merge(B, trust2, (H, W), mask, ifn_L2, scores_back, validate=False)
        tail = int(third_set[num][1])
        if tail != 0:                                           # pats.py:38-39
            ifn_L2[-tail:, :] = True
        # ---- third layer (pats.py:53-58, third_layer.py:121-128,153-170) ------------------------------
        mk0, mk1, b_ids = ops.third_inputs(ifn_L2, pts2)        # host read: P
        P = int(mk0.shape[0])
        if P == 0:                                              # pats.py:42
            info.append((B, 0, 0))
            continue
        feat0, feat1, scale3 = nets.third(num, mk0, mk1, b_ids)
        p_s, p_t = _round4(mk0, False), _round4(mk1, True)
        m0f, m1f, label, ifm = ops.third_level(feat0, feat1, scale3, p_s, p_t, outdoor=if_outdoor, iters=iters)
        # ---- results (pats.py:59-78) ----------------------------------------------------------------------
        ifn16, pts16 = ops.refine_scatter(ifn_L2, pts2, m1f, label)
        sc_rows = xsn.reshape(-1, w * h, 2)[torch.logical_not(mask)]                  # [B,2]   (pats.py:70)
        ml, mr = ops.get_result(1, [mask, ifn16], [avn.flip(dims=[2]) / 32.0, pts16.flip(dims=[2]) / 2.0],
                                [xsn, sc_rows], [[32, h, w], [2, 48, 48]],
                                [torch.ones([1], dtype=torch.bool, device=dev),
                                 torch.ones([B], dtype=torch.bool, device=dev)], validate=False)   # host read: M
        out_l.append(ml)
        out_r.append(mr)
        info.append((B, P, int(ml.shape[0])))
    return {"matches_l": torch.cat(out_l) if out_l else empty, "matches_r": torch.cat(out_r) if out_r else empty,
            "chunks": info}


def _forward_batched(left, nets, if_outdoor, iters, merge, scores_back, ifn1, sum_cycle, second_set, third_set, K,
                     new_left, new_right, xsn, avn, h, w, H, W):
    dev = left.device
    empty = torch.zeros([0, 2], device=dev)
    spans = [(lo, min(hi, K), int(tl[1])) for (lo, hi), tl in zip(second_set, third_set) if min(hi, K) - lo > 0]
    if not spans:
        return {"matches_l": empty, "matches_r": empty, "chunks": []}
    sizes = [hi - lo for lo, hi, _ in spans]
    C, Bt = len(spans), sum(sizes)
    masks = torch.cat([torch.logical_or(ifn1, torch.logical_or(sum_cycle <= lo, sum_cycle > hi))
                       for lo, hi in second_set if min(hi, K) - lo > 0])                  # [C,N]  (first_layer.py:137-138)
    rows = torch.cat([torch.arange(lo, hi, device=dev) for lo, hi, _ in spans])          # overlap rows appear twice
    f0, f1, sx, sy = nets.fine(None, new_left[rows], new_right[rows], masks, sizes=sizes)[:4]
    Z2, cflag2 = ops.cost_ot(f0, f1, 2, 1.0, (sx * sy).contiguous(), iters, bias_k=2.0 if if_outdoor else 3.0,
                             return_flags=True)
    trust2, pts2, _, _, ifn_L2, _ = ops.est_position_second(Z2, sx, sy, [96, 96], 8, col_nomatch=cflag2)
    merged = torch.empty_like(ifn_L2)
    off = 0
    for c, (lo, hi, tail) in enumerate(spans):                 # chunk order: scores_back couples them
        B = hi - lo
        out, scores_back = merge(B, trust2[off:off + B], (H, W), masks[c:c + 1], ifn_L2[off:off + B], scores_back,
                                 validate=False)
        if tail != 0:                                           # pats.py:38-39, the reference's own expression
            out[-tail:, :] = True                               # (identical in both modes also for a negative tail)
        merged[off:off + B] = out
        off += B
    mk0, mk1, b_ids = ops.third_inputs(merged, pts2)            # host read: P (all chunks)
    if mk0.shape[0] == 0:
        return {"matches_l": empty, "matches_r": empty, "chunks": [(b, 0, 0) for b in sizes]}
    feat0, feat1, scale3 = nets.third(None, mk0, mk1, b_ids, sizes=sizes)
    m0f, m1f, label, ifm = ops.third_level(feat0, feat1, scale3, _round4(mk0, False), _round4(mk1, True),
                                           outdoor=if_outdoor, iters=iters)
    ifn16, pts16 = ops.refine_scatter(merged, pts2, m1f, label)
    # get_result with the chunks as level-0 batch: row k of level 1 belongs to the k-th unmasked cell in
    # (chunk, patch) order = the concatenation order
    xs_c, av_c = xsn.expand(C, -1, -1).contiguous(), avn.expand(C, -1, -1).contiguous()
    sc_rows = xs_c[torch.logical_not(masks)]                                            # [Bt,2]
    ml, mr = ops.get_result(C, [masks, ifn16], [av_c.flip(dims=[2]) / 32.0, pts16.flip(dims=[2]) / 2.0],
                            [xs_c, sc_rows], [[32, h, w], [2, 48, 48]],
                            [torch.ones([C], dtype=torch.bool, device=dev), torch.ones([Bt], dtype=torch.bool, device=dev)],
                            validate=False)                       # host read: M
    return {"matches_l": ml, "matches_r": mr, "chunks": [(b, -1, -1) for b in sizes]}


_ONE = {}


def _one(device):
    """The reference's `self.one` (second_layer.py:63): a device-resident 1.0, made once per device."""
    key = str(device)
    if key not in _ONE:
        _ONE[key] = torch.tensor(1.0, device=device)
    return _ONE[key]


_SIDE = {}


def _side_streams(device, n):
    """The chunk walk's side streams, made once per device (a new HIP stream per pair cost up to 90 ms now and then)."""
    pool = _SIDE.setdefault(str(device), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


class _ChunkTable:
    """What ops.get_result_chunks reads of a ChunkRows table, for ONE chunk of one pair (its mask as the level-0 flags)."""
    __slots__ = ("Cmax", "pairs", "h", "w", "rows_cap", "masks")

    def __init__(self, rows, c, B):
        self.Cmax, self.pairs, self.h, self.w, self.rows_cap, self.masks = 1, 1, rows.h, rows.w, int(B), rows.masks[c:c + 1]


def forward_chunks_device(left, right, nets, if_local=True, if_outdoor=True, merge_new=True, iters=100, streams=1):
    """PATS.forward's control flow (pats.py:18-85: the first layer, then chunk by chunk the second layer, the merge, the third
    layer and get_result) with the counts the reference reads back INSIDE the chunk loop left on the device.  Host reads per
    pair: the chunk plan (the chunk sizes B size the backbone's batch, so the host must know them: first_layer.py:130-146) and,
    after the last chunk, the per-chunk match counts.  Per chunk: ~24 launches, no torch kernel, no synchronisation.
      * the chunk masks, the rows' cells and pats.py:38-39's tail rows come from the device-side table (ops.chunk_rows);
      * the merge is ONE launch per chunk on the chunk's own tensors (ops.merge_patches_chunk), scores_back handed on;
      * the third level runs over the chunk's capacity 144 B with the count P on the device (ops.third_inputs(sync=False),
        ops.third_level(count=P)); get_result likewise (ops.get_result_chunks on the chunk's mask).
    Callbacks as in forward_path, except that the third level's tensors are a CAPACITY and the count arrives with them:
        nets.third(num, mkpts0_c [144 B,2], mkpts1_c [144 B,2], b_ids [144 B], count=P_dev [1] int64) ->
            feat0 [144 B,128,65], feat1, scale [144 B,1,64] [, p_s, p_t [144 B,2] int64 - ops.third_descriptors' roundings]
        nets.fine may append scale_x * scale_y as a fifth tensor.
    streams > 1: chunk c runs on side stream c % streams (its kernels are too small to fill the GPU: ~54 rows); the merges
    wait for each other in chunk order (scores_back, pats.py:37).  Same matches, same order."""
    dev = left.device
    H, W = int(left.shape[1]), int(left.shape[2])
    h, w = H // 32, W // 32
    empty = torch.zeros([0, 2], device=dev)
    mdesc0, mdesc1, scale, alpha = nets.coarse(left, right)
    scores = ops.cost_ot(mdesc0, mdesc1, 1, alpha, scale, iters)
    scales, cflag = ops.colmass_sqrt(scores, return_flags=True)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_first(scores, scales, (H, W), 32, col_nomatch=cflag)
    rows = ops.chunk_rows(ifn1, h, w, 2 * w if if_local else 512)
    new_left, new_right, xsn, ysn, avn, _, _, _ = ops.Compute_imgs_ex(xs, ys, pts, ifn1, left, right, width=w, height=h,
                                                                      known_count="device")
    # host read 1: the chunk plan (row offsets of the chunks, their first crops, the table's status)
    plan = torch.cat([rows.chunk_base, rows.second.reshape(-1), rows.status.to(torch.int64)]).cpu().tolist()
    Cmax = rows.Cmax
    base, second, status = plan[:Cmax + 1], plan[Cmax + 1:-1], plan[-1]
    if status:
        raise RuntimeError("pats_amd.pipeline: the chunk table overflowed (status %d)" % status)
    if base[-1] <= 0:                                           # pats.py:27-31
        return {"matches_l": empty, "matches_r": empty, "chunks": []}
    scores_back = torch.empty([1, h * w, 16, 9], dtype=torch.float64, device=dev)       # pats.py:32 (cleared by the first merge)
    cur = torch.cuda.current_stream()
    side = _side_streams(dev, streams) if streams > 1 else None
    if side is not None:
        ready = torch.cuda.Event()
        ready.record(cur)
    one, bias_k = _one(dev), 2.0 if if_outdoor else 3.0
    parts, sizes, prev_merge, first = [], [], None, True
    for c in range(Cmax):
        B = base[c + 1] - base[c]
        if B <= 0:
            continue
        lo = second[2 * c]
        st = side[len(parts) % streams] if side is not None else cur
        with torch.cuda.stream(st):
            if side is not None and len(parts) < streams:
                st.wait_event(ready)
            # ---- second layer tail (second_layer.py:100-118) ----------------------------------------------------------
            fine = nets.fine(c, new_left[lo:lo + B], new_right[lo:lo + B], rows.masks[c])
            f0, f1, sx, sy = fine[:4]
            ns2 = fine[4] if len(fine) > 4 else (sx * sy).contiguous()
            # ---- cost + OT + expansion + merge + third-level inputs: ONE C call (ops.chunk_fine_tail); the merges in chunk order ------
            done = None
            if side is not None:                              # (the call waits for the previous chunk's merge right before its own and
                done = torch.cuda.Event()                     #  re-records `done` right behind it; recorded here once so its handle exists)
                done.record(st)
            merged, pts2, mk0, mk1, b_ids, P = ops.chunk_fine_tail(f0, f1, one, ns2, sx, sy, iters, bias_k, merge_new, rows, c, base[c],
                                                                   (H, W), scores_back, first, wait_before_merge=prev_merge,
                                                                   record_after_merge=done)
            first = False
            prev_merge = done
            # ---- third layer over the chunk's capacity + scatter + get_result: the callback, then ONE C call (ops.chunk_third_tail) ----
            third = nets.third(c, mk0, mk1, b_ids, count=P)
            feat0, feat1, scale3 = third[:3]
            p_s, p_t = third[3:5] if len(third) > 3 else (_round4(mk0, False), _round4(mk1, True))
            ml, mr, M = ops.chunk_third_tail(feat0, feat1, P, scale3, p_s, p_t, iters, if_outdoor, merged, pts2, rows.masks[c], h, w, avn, xsn)
        parts.append((ml, mr, M, P))
        sizes.append(B)
    if side is not None:
        for st in side:
            cur.wait_stream(st)
        for ml, mr, M, P in parts:
            for t in (ml, mr, M, P):
                t.record_stream(cur)
    # host read 2: the counts of every chunk
    counts = torch.cat([torch.cat([M, P]) for _, _, M, P in parts]).cpu().tolist()
    Ms, Ps = counts[0::2], counts[1::2]
    out_l = [ml[:m] for (ml, _, _, _), m in zip(parts, Ms) if m > 0]
    out_r = [mr[:m] for (_, mr, _, _), m in zip(parts, Ms) if m > 0]
    return {"matches_l": torch.cat(out_l) if out_l else empty, "matches_r": torch.cat(out_r) if out_r else empty,
            "chunks": [(b, p_, m) for b, p_, m in zip(sizes, Ps, Ms)]}
