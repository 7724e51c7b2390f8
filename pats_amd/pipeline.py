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


def forward_path(left, right, nets, if_local=True, if_outdoor=True, merge_new=True, iters=100, batch_chunks=False):
    """left / right: [1,H,W,3] float32 HWC images (what first_layer.py:128-129 permutes to).
    Returns {"matches_l": [M,2], "matches_r": [M,2]} in the reference's (row, col) pixel convention and
    order, plus "chunks": per-chunk (B, P, M) for inspection.

    batch_chunks=True runs all chunks of the pair together (the reference walks them one by one to bound
    memory on a 16-40 GB card): ONE fine-level cost+OT+expansion launch over the concatenated chunk rows,
    the merges in chunk order (their only coupling is scores_back, pats.py:32,37), ONE third-level
    launch, ONE get_result with the chunks as its batch dimension; three host reads per pair.  The
    callbacks are then called once with num=None and the list of per-chunk row counts:
        nets.fine(None, new_left_all, new_right_all, masks [C,N], sizes=[B_0, ...])
        nets.third(None, mkpts0_c, mkpts1_c, b_ids (rows of the concatenation), sizes=[B_0, ...])
    Same matches in the same order (tests/test_gpu_parity.py::test_pipeline_chain runs both modes)."""
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
        f0, f1, sx, sy = nets.fine(num, new_left[lo:hi_c], new_right[lo:hi_c], mask)
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
    f0, f1, sx, sy = nets.fine(None, new_left[rows], new_right[rows], masks, sizes=sizes)
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
