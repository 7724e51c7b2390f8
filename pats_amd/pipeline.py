"""The control flow of PATS.forward (models/pats.py:18-85) and of the three layers' forward tails
(first_layer.py:110-157, second_layer.py:100-124, third_layer.py:153-170) on the device-side ops
of pats_amd, with the networks (ResNet / FPN / AttentionalGNN / projection heads - out of scope,
DESIGN.md section 7) abstracted as three callbacks.

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


def forward_path(left, right, nets, if_local=True, if_outdoor=True, merge_new=True, iters=100):
    """left / right: [1,H,W,3] float32 HWC images (what first_layer.py:128-129 permutes to).
    Returns {"matches_l": [M,2], "matches_r": [M,2]} in the reference's (row, col) pixel convention and
    order, plus "chunks": per-chunk (B, P, M) for inspection."""
    dev = left.device
    H, W = int(left.shape[1]), int(left.shape[2])
    h, w = H // 32, W // 32
    empty = torch.zeros([0, 2], device=dev)
    # ---- first layer tail (first_layer.py:110-146) ------------------------------------------------
    mdesc0, mdesc1, scale, alpha = nets.coarse(left, right)
    scores = ops.cost_ot(mdesc0, mdesc1, 1, alpha, scale, iters)
    scales = ops.colmass_sqrt(scores)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_first(scores, scales, (H, W), 32)
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
        Z2 = ops.cost_ot(f0, f1, 2, 1.0, (sx * sy).contiguous(), iters, bias_k=2.0 if if_outdoor else 3.0)
        trust2, pts2, _, _, ifn_L2, _ = ops.est_position_second(Z2, sx, sy, [96, 96], 8)
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
