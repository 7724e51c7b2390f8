"""Throughput mode of pats_amd.pipeline: PATS.forward's hot path (models/pats.py:18-85 and the three layers' forward
tails) for a BATCH of image pairs, without a single host read between the coarse descriptors and the matches.

The reference runs one pair at a time and, inside it, one chunk at a time (evaluate.py:25, pats.py:33,
first_layer.py:131-146) because it targets a 16-40 GB card, and every boolean-mask indexing on the way is a
device->host sync that sizes the next tensor.  Pairs are independent and the chunks of a pair couple only through the
merge's `scores_back` (pats.py:32,37), so with 288 GB of HBM every stage becomes ONE launch over all pairs and chunks:

  coarse   cost + log_optimal_transport + column mass + area expansion for all pairs        (first_layer.py:110-127)
  plan     cumulative match counts, split_patches, chunk masks, the fine level's row table   (first_layer.py:130-146)
           rows ordered (chunk, pair, cell): chunk c of every pair is one contiguous block   (ops.chunk_rows)
  crops    Compute_imgs for all pairs, (image, patch) order, counts on the device            (utils.py:1343-1393)
  fine     cost + log_optimal_transport2 + ln k + area expansion for all rows                (second_layer.py:100-118)
  merge    merge_patches_new / _old: chunk blocks in order, each over all pairs, tail rows   (second_layer.py:119-122,
           masked                                                                             pats.py:38-39)
  third    surviving cells -> points (pats.py:53-58), cost + OT + Compute_result + label    (third_layer.py:153-170)
           over a CAPACITY with the count on the device
  result   scatter onto the 48x48 sub-cell grid (pats.py:59-67), get_result for all chunks   (pats.py:68-78)

Sizes are capacities fixed on the host (`Capacities`); the counts the reference reads back (matched patches K, rows B,
third-level problems P, matches M) stay on the device and come back WITH the results: `status` / `P` / `M` are read by
the caller when it fetches the matches.  Rows of the table that no chunk uses (padding past the device-side total
`rows.chunk_base[-1]`) are SKIPPED by the fine level's launches (cost build, OT, expansion take the count from the device;
a network callback may do the same with ops.fine_descriptors(count=...)) and are "no match" at the merge; rows that are left
without a cell are not compacted (pats.py:40-52) - they emit nothing, exactly as in pipeline.forward_path.

Network callbacks (`nets`, the out-of-scope backbones + heads; GPU float32 tensors, no host read required of them):
  nets.coarse(lefts, rights) -> mdesc0 [pairs,D,N], mdesc1 [pairs,D,N], scale [pairs,1,N], alpha
  nets.fine(rows, new_left, new_right) -> mdesc0 [rows_cap,264,145], mdesc1, scale_x [rows_cap,1,144], scale_y
      [, scale_x * scale_y] (what ops.scale_head hands out; formed here if absent)
      rows: ops.ChunkRows (row r shows crop rows.row_crop[r] of new_left / new_right [pairs*N,96,96,3])
  nets.third(rows, mkpts0_c [P_cap,2], mkpts1_c [P_cap,2], b_ids [P_cap], P_dev [1]) ->
      feat0 [P_cap,128,65], feat1 [P_cap,128,65], scale [P_cap,1,64] [, p_s, p_t [P_cap,2] int64: the points rounded to
      the 4-px lattice as ops.third_descriptors returns them; formed here if absent]      (b_ids = row of the table)
"""
import torch

from . import ops


class Capacities:
    """Host-side sizes of one batch.  p_cap_per_pair bounds the third-level problems of a pair: the merge leaves every
    8-px cell to at most one window per chunk, so 16*h*w cells is the natural size (cells handed over between chunks can
    add a few; the default keeps 25 % headroom; an overflow is reported in the result, never silent)."""

    def __init__(self, pairs, h, w, if_local=True, p_cap_per_pair=None, rows_cap=None):
        self.pairs, self.h, self.w = int(pairs), int(h), int(w)
        self.N = self.h * self.w
        self.chunk_cap = 2 * self.w if if_local else 512                      # first_layer.py:131-135
        self.Cmax = ops.max_chunks(self.h, self.w, self.chunk_cap)
        # worst case: every cell matched, every chunk boundary repeats one grid row.  A caller that knows its data (a dry
        # run of the coarse stage) may pass a tighter rows_cap: the fine level then runs fewer padding rows, and a batch
        # that does not fit is reported through `status` (split_by_pair raises), never truncated silently.
        self.rows_cap = self.pairs * (self.N + (self.Cmax - 1) * self.w) if rows_cap is None else int(rows_cap)
        per_pair = int(1.25 * 16 * self.N) if p_cap_per_pair is None else int(p_cap_per_pair)
        self.P_cap = self.pairs * per_pair


_ONE = {}


def _one(device):
    """The reference's `self.one` (second_layer.py:63): a device-resident 1.0, made once per device."""
    key = str(device)
    if key not in _ONE:
        _ONE[key] = torch.tensor(1.0, device=device)
    return _ONE[key]


def _round4(x, clamp96):
    """third_layer.py:122 / :126-128: round(x / 4).long() * 4 (targets clamped to [0, 96] first)."""
    if clamp96:
        x = torch.clamp(x, 0.0, 96.0)           # == the reference's two torch.where; python scalars: no H2D copy per call
    return torch.round(x / 4.0).long() * 4


def coarse_stage(lefts, rights, nets, cap, iters=100, fine_inputs=True):
    """(fine_inputs="rows_only": stop after the row table - capacity planning.)
    The first layer's tail for all pairs + the chunk plan / row table + the crops (first_layer.py:110-146,
    utils.py:1343-1393) and - fine_inputs=True - the second layer's descriptors for those rows (nets.fine: backbone on the
    crops + the a15 gather).  Independent of every other batch and HBM-bound (crops, gathers): a caller may run it on a
    stream of its own beside the solver stages of the previous batches (bench.py does)."""
    H, W = int(lefts.shape[1]), int(lefts.shape[2])
    h, w = cap.h, cap.w
    assert (H // 32, W // 32) == (h, w) and lefts.shape[0] == cap.pairs
    mdesc0, mdesc1, scale, alpha = nets.coarse(lefts, rights)
    Z = ops.cost_ot(mdesc0, mdesc1, 1, alpha, scale, iters)
    scales, cflag = ops.colmass_sqrt(Z, return_flags=True)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_first(Z, scales, (H, W), 32, col_nomatch=cflag)
    rows = ops.chunk_rows(ifn1, h, w, cap.chunk_cap, Cmax=cap.Cmax, rows_cap=cap.rows_cap)
    if fine_inputs == "rows_only":
        return {"rows": rows, "ifn1": ifn1}
    new_left, new_right, xsn, ysn, avn, bound5, K_img, K_tot = ops.Compute_imgs_ex(
        xs, ys, pts, ifn1, lefts, rights, width=w, height=h, known_count="device")
    co = {"rows": rows, "new_left": new_left, "new_right": new_right, "xsn": xsn, "avn": avn, "K_img": K_img,
          "ifn1": ifn1, "H": H, "W": W}
    if fine_inputs:
        co["fine"] = nets.fine(rows, new_left, new_right)
    return co


def _timed(events, tag):
    if events is None:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    events.setdefault(tag, []).append((e0, e1))
    e0.record()
    return e1


def fine_solve_stage(co, nets, cap, if_outdoor=True, merge_new=True, iters=100, events=None):
    """second_layer.py:100-122 + pats.py:38-39,53-58 for every row of a coarse_stage result: cost + OT + expansion, the
    merges in chunk order, the surviving cells' points (VALU-bound)."""
    rows, H, W = co["rows"], co["H"], co["W"]
    fine = co["fine"] if "fine" in co else nets.fine(rows, co["new_left"], co["new_right"])
    f0, f1, sx, sy = fine[:4]
    ns2 = fine[4] if len(fine) > 4 else (sx * sy).contiguous()
    e = _timed(events, "fine")
    if events is not None:                      # the boundary between the pair's two kernels, recorded inside the C call
        em = torch.cuda.Event(enable_timing=True)
        em.record()
        events.setdefault("fine_mid", []).append(em)
        ops.set_cost_ot_mid_event(em)
    live = rows.chunk_base[-1:]                 # rows in use, on the device: the launches cover rows_cap, padding rows are skipped
    Z2, cflag2 = ops.cost_ot(f0, f1, 2, _one(f0.device), ns2, iters, bias_k=2.0 if if_outdoor else 3.0, return_flags=True,
                             count=live)
    if e is not None:
        e.record()
    trust2, pts2, _, _, ifn_L2, _ = ops.est_position_second(Z2, sx, sy, [96, 96], 8, col_nomatch=cflag2, count=live)
    merged = ops.merge_patches_batch(merge_new, rows, trust2, (H, W), ifn_L2)
    mk0, mk1, b_ids, P = ops.third_inputs(merged, pts2, capacity=cap.P_cap, sync=False)
    return {"co": co, "merged": merged, "pts2": pts2, "P": P, "mk0": mk0, "mk1": mk1, "b_ids": b_ids,
            "stages": {"Z2": Z2, "trust2": trust2, "pts2": pts2, "ifn_L2": ifn_L2, "sx": sx, "sy": sy, "f0": f0,
                       "f1": f1, "ns2": ns2, "mk0": mk0, "mk1": mk1, "b_ids": b_ids}}


def third_gather_stage(fs, nets, cap):
    """The third layer's descriptors for the surviving cells (nets.third: backbone maps + the a16 window gather;
    HBM-bound).  Completes a fine_solve_stage result in place."""
    third = nets.third(fs["co"]["rows"], fs["mk0"], fs["mk1"], fs["b_ids"], fs["P"])
    feat0, feat1, scale3 = third[:3]
    p_s, p_t = third[3:5] if len(third) > 3 else (_round4(fs["mk0"], False), _round4(fs["mk1"], True))
    fs.update(feat0=feat0, feat1=feat1, scale3=scale3, p_s=p_s, p_t=p_t)
    fs["stages"].update(feat0=feat0, feat1=feat1, scale3=scale3, p_s=p_s, p_t=p_t)
    return fs


def fine_stage(co, nets, cap, if_outdoor=True, merge_new=True, iters=100, events=None):
    return third_gather_stage(fine_solve_stage(co, nets, cap, if_outdoor, merge_new, iters, events), nets, cap)


def third_stage(fs, nets, cap, if_outdoor=True, iters=100, events=None):
    """third_layer.py:153-170 over the capacity with the count on the device, then pats.py:59-78: the scatter onto the
    sub-cell grid and get_result for every chunk of every pair."""
    co, rows, P = fs["co"], fs["co"]["rows"], fs["P"]
    e = _timed(events, "third")
    m0f, m1f, label, ifm = ops.third_level(fs["feat0"], fs["feat1"], fs["scale3"], fs["p_s"], fs["p_t"], outdoor=if_outdoor,
                                           iters=iters, count=P)
    if e is not None:
        e.record()
    ifn16, pts16 = ops.refine_scatter(fs["merged"], fs["pts2"], m1f, label)
    ml, mr, mrow, M = ops.get_result_chunks(rows, ifn16, co["avn"], pts16, co["xsn"])
    stages = dict(fs["stages"], m0f=m0f, m1f=m1f, label=label, ifm=ifm, pts16=pts16)
    return {"matches_l": ml, "matches_r": mr, "match_row": mrow, "M": M, "P": P, "status": rows.status, "rows": rows,
            "if_nomatching16": ifn16, "merged": fs["merged"], "K_img": co["K_img"], "crops": (co["new_left"], co["new_right"]),
            "coarse": co, "stages": stages}


def fine_third_stage(co, nets, cap, if_outdoor=True, merge_new=True, iters=100, events=None):
    fs = fine_stage(co, nets, cap, if_outdoor, merge_new, iters, events)
    return third_stage(fs, nets, cap, if_outdoor, iters, events)


def forward_pairs(lefts, rights, nets, cap, if_outdoor=True, merge_new=True, iters=100, events=None):
    """lefts / rights [pairs,H,W,3] float32 HWC.  Returns a dict of DEVICE tensors:
        matches_l, matches_r [M_cap,2]   the first M rows valid, reference order inside every pair (chunk, patch, sub-cell)
        match_row [M_cap] int32          row of the table per match;  rows.row_cell[match_row] // N = pair
        M, P [1] int64, status [1] int32 match count, third-level problem count (P > cap.P_cap = overflow), table status
        rows                             the ops.ChunkRows table
        stages                           the intermediate tensors (parity checks; nothing reads them here)
    No host read happens in here."""
    co = coarse_stage(lefts, rights, nets, cap, iters, fine_inputs=False)
    return fine_third_stage(co, nets, cap, if_outdoor, merge_new, iters, events)


def group_by_pair(out, cap, buffers=None):
    """Device side of the hand-over: the batch's matches regrouped by pair (ops.matches_by_pair), no host read.  Adds
    `by_pair` = (matches_l, matches_r, pair_off) and `summary` (int64 [pairs + 4]: the pairs + 1 offsets, then M, P, table status -
    everything the host reads of a step, in ONE buffer) to the result; a caller in a loop passes `buffers` (out_l, out_r,
    summary-sized pair_off) to reuse the outputs."""
    ml, mr, off, summary = ops.matches_by_pair(out["rows"], out["matches_l"], out["matches_r"], out["match_row"], out["M"], out=buffers,
                                               P=out["P"])
    out["by_pair"], out["summary"] = (ml, mr, off), summary
    return out["by_pair"]


def split_by_pair(out, cap):
    """Host side, AFTER the step: per-pair (matches_l, matches_r) lists from a forward_pairs result, in the reference's
    order.  Reads the counts back (the one synchronisation of a batch) and raises on a capacity overflow."""
    if "summary" not in out:
        group_by_pair(out, cap)
    o = out["summary"].cpu().tolist()                 # the one synchronisation of a batch: offsets, M, P, status in one copy
    M, P, status = o[cap.pairs + 1:]
    if status & 1:
        raise RuntimeError("pats_amd.batch: a pair needed more than Cmax = %d chunks" % cap.Cmax)
    if status & 2:
        raise RuntimeError("pats_amd.batch: the row table overflowed rows_cap = %d" % cap.rows_cap)
    if P > cap.P_cap:
        raise RuntimeError("pats_amd.batch: %d third-level problems exceed P_cap = %d" % (P, cap.P_cap))
    ml, mr, _ = out["by_pair"]
    return [(ml[o[p]:o[p + 1]], mr[o[p]:o[p + 1]]) for p in range(cap.pairs)]
