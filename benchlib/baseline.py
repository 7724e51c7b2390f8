"""bench.py: the CPU baseline (`cpu_baseline`): the oracle on one whole pair of a step - checker code, timed as a baseline only - its
stage-by-stage parity sample, and the torch-CPU transcription of the reference's arithmetic."""
import json  # noqa: F401
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import REPO, synth  # noqa: F401


def torch_cpu_sinkhorn(Z, log_mu, log_nu, iters):
    """What the reference executes on CPU (modules.py:137-143), transcribed: logsumexp row / column sweeps."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def torch_cpu_cost_ot2(d0, d1, ns, iters):
    """second_layer.py:100-104 / third_layer.py:156-158 on CPU tensors: einsum cost build + log_optimal_transport2."""
    scores = 0.1 * (torch.einsum("bdn,bdm->bnm", d0, d1) / d0.shape[1] ** .5)
    b, m, n = scores.shape
    ms = torch.tensor(float(m - 1))
    nssum = ns.sum(dim=2)                                             # [b,1]
    norm = -(ms + nssum).log()
    log_nu = torch.cat([ns.log()[:, 0] + norm, ms.log().expand(b, 1) + norm], dim=1)
    log_mu = torch.cat([norm.expand(b, m - 1), nssum.log() + norm], dim=1)
    return torch_cpu_sinkhorn(scores, log_mu, log_nu, iters) - norm[:, :, None]


ULP4 = 4.0 * 2.0 ** -23     # "a threshold tie": the deciding quantities agree to 4 ulp


def expansion_parity(ops, oracle, dev, sx, sy, gZ2, Z2):
    """Area expansion of pair 0's fine problems (utils.py:1213-1243), HIP against the oracle, every differing row classified.

    (1) SAME INPUT: the oracle expands the plan the GPU produced (exp on the GPU, the identical fp32 array on both
        sides), so the only freedom left is the summation order of a strip.  A row whose rectangle differs is a
        threshold tie if the oracle's own decision margin - the relative distance between the strip sum that decided
        and `lower_bound` / the competing strip (oracle_iterative_expand_margin) - is within 4 ulp; anything else is a
        REAL mismatch, and the bench asserts there is none.
    (2) END TO END: each side expands its OWN plan.  The plans agree to the 1e-4 transport-mass gate, not bit for bit,
        and `lower_bound` = 1e-3 is only 10x that gate, so a strip sum - or a single strip cell, for the per-element test
        of :1225 that charges the opposite dustbin mass to whole_cost - that lands within the measured plan difference of
        the threshold is counted on one side only; such a row carries a different trust score (this is where round 2's
        unexplained max |d trust| = 0.04 came from: one row, one cell) and possibly a different rectangle.  A differing
        row is "explained" if its margin is below what the measured plan difference of its problem can move a strip sum
        by (12 cells x max |dP|, relative to lower_bound); anything else is REAL and asserted zero."""
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    nb = gZ2.shape[0]
    pos, rng_ = ops.Compute_positions_and_ranges(12, 12, dev)
    gP = ops.exp(gZ2)
    gsame = ops.Iterative_expand_matrix(gP, td(sx).reshape(nb, -1, 1), td(sy).reshape(nb, -1, 1), [0, 12, 0, 12], rng_, pos,
                                        lower_bound=1e-3, iter_num=8, width=12, height=12)
    gbound, gtrust = gsame[5].cpu().numpy(), gsame[0].cpu().numpy()
    osame = oracle.iterative_expand(gP.cpu().numpy(), sx, sy, 12, 12, 12, 1e-3, 8, with_margin=True)
    diff_rows = (gbound != osame[5]).any(axis=2)
    tie = diff_rows & (osame[6][..., 0] <= ULP4)
    real_same = diff_rows & ~tie
    elem_tie = osame[6][..., 1] <= ULP4
    dtrust = np.abs(gtrust - osame[0])
    ok_rows = ~diff_rows & ~elem_tie
    trust_same = float(dtrust[ok_rows].max()) if ok_rows.any() else 0.0
    # end to end (each side its own plan)
    own = oracle.iterative_expand(np.exp(Z2), sx, sy, 12, 12, 12, 1e-3, 8, with_margin=True)
    ebound_diff = (gbound != own[5]).any(axis=2)
    dP = np.abs(np.exp(gZ2.cpu().numpy().astype(np.float64)) - np.exp(Z2.astype(np.float64)))[:, :-1, :].max(axis=(1, 2))
    reach = (12.0 * dP / 1e-3 + ULP4)[:, None]
    explained = ebound_diff & (own[6][..., 0] <= reach)
    real_e2e = ebound_diff & ~explained
    dtrust_e2e = np.abs(gtrust - own[0])
    tol_t = 1e-4 + 1e-4 * np.abs(own[0])
    tdiff = (dtrust_e2e > tol_t) & ~ebound_diff
    t_explained = tdiff & (own[6][..., 1] <= reach)
    return {
        "l2_rows": int(diff_rows.size),
        "l2_bound_mismatch_same_input": int(diff_rows.sum()), "l2_bound_threshold_ties": int(tie.sum()),
        "l2_bound_real_mismatch": int(real_same.sum() + real_e2e.sum()),
        "l2_trust_max_abs_diff_same_input_same_rectangle": trust_same,
        "l2_bound_mismatch_end_to_end": int(ebound_diff.sum()),
        "l2_bound_mismatch_end_to_end_explained_by_plan_difference": int(explained.sum()),
        "l2_plan_max_abs_diff": float(dP.max()),
        "l2_trust_max_abs_diff": float(dtrust_e2e.max()), "l2_trust_max_abs": float(np.abs(own[0]).max()),
        "l2_trust_rows_differing_with_equal_rectangles": int(tdiff.sum()),
        "l2_trust_rows_explained_by_element_threshold": int(t_explained.sum()),
        "l2_trust_real_mismatch": int((tdiff & ~t_explained).sum()),
    }


def cpu_baseline(ops, batch, dev, nets, cap, wl, out, torch_leg=True):
    """The CPU oracle ("port") on the host cores over ONE WHOLE PAIR (pair 0 of a step: L1 in full, every fine problem,
    every third-level problem the merge left, the merges, the scatter and get_result) - measured, not extrapolated.
    Each stage is fed what the GPU handed its own next stage, so the same run is a stage-by-stage parity check on the
    bench's own data (`parity_sample`; index outputs are ASSERTED).  Beside it the torch-CPU transcription of
    modules.py:137-182 + the einsum cost builds, on samples.  Checker code, timed as a baseline only."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import pats_oracle as oracle
    cores = oracle.num_threads()
    h, w, N = cap.h, cap.w, cap.N
    H, W = 32 * h, 32 * w
    st, rows, co = out["stages"], out["rows"], out["coarse"]
    cell = rows.row_cell.cpu().numpy()
    total = int(rows.chunk_base[-1].item())
    rows0 = np.nonzero((cell[:total] >= 0) & (cell[:total] // N == 0))[0]
    B0 = len(rows0)
    base = rows.chunk_base.cpu().numpy()
    r0t = torch.from_numpy(rows0).to(dev)
    cpu = lambda t: t.detach().cpu().numpy()   # noqa: E731
    times = {}

    # ---- L1 (first_layer.py:110-127) -------------------------------------------------------------------------------
    d0, d1, ns = cpu(nets.d0[0:1]), cpu(nets.d1[0:1]), cpu(nets.ns[0:1])
    t0 = time.perf_counter()
    S = oracle.cost(d0, d1)
    Z = oracle.log_optimal_transport(S, float(nets.alpha.item()), ns, ITERS)
    sc = oracle.colmass_sqrt(Z)
    r1, c1 = oracle.argmax(Z)
    oracle.iterative_expand(np.exp(Z), sc, sc, w, h, w, 1e-5, 15)
    times["L1"] = time.perf_counter() - t0
    ifn1_o = (r1[:, :-1] == N)
    parity = {"pair": 0, "l1_if_nomatching_mismatch": int((ifn1_o[0] != cpu(co["ifn1"][0])).sum()),
              "l1_matched_patches": int((~ifn1_o).sum())}

    # ---- L2 (second_layer.py:100-118) on the descriptors the GPU's gather produced ---------------------------------
    f0, f1 = cpu(st["f0"][r0t]), cpu(st["f1"][r0t])
    sx, sy = cpu(st["sx"][r0t]), cpu(st["sy"][r0t])
    t0 = time.perf_counter()
    S2 = oracle.cost(f0, f1)
    Z2 = oracle.dustbin_bias(oracle.log_optimal_transport2(S2, 1.0, sx * sy, ITERS), wl["bias_k"])
    r2, c2 = oracle.argmax(Z2)
    oracle.iterative_expand(np.exp(Z2), sx, sy, 12, 12, 12, 1e-3, 8)
    times["L2"] = time.perf_counter() - t0
    gZ2 = st["Z2"][r0t].contiguous()
    gr2, gc2 = ops.argmax(gZ2)
    e2, e2r = np.exp(cpu(gZ2).astype(np.float64)), np.exp(Z2.astype(np.float64))
    pre = ops.est_position_second(gZ2, st["sx"][r0t].contiguous(), st["sy"][r0t].contiguous(), [96, 96], 8)   # before the merge
    parity.update({"l2_problems": B0, "l2_row_argmax_mismatch": int((cpu(gr2) != r2).sum()),
                   "l2_col_argmax_mismatch": int((cpu(gc2) != c2).sum()),
                   "l2_mass_max_abs_diff": float(np.abs(e2[:, :-1, :-1] - e2r[:, :-1, :-1]).max()),
                   "l2_flag_mismatch": int((cpu(pre[4]) != (r2[:, :-1] == 144)).sum() + (cpu(pre[5]) != (c2[:, :-1] == 144)).sum())})
    parity.update(expansion_parity(ops, oracle, dev, sx, sy, gZ2, Z2))

    # ---- merge (second_layer.py:119-122, pats.py:38-39): the oracle on the GPU's trust scores, chunk after chunk -----
    trust_g, ifn_g = cpu(pre[0]), cpu(pre[4])
    masks0 = cpu(rows.masks[:, 0, :])
    third_set = cpu(rows.third[0])
    nchunks = int(rows.cycle_num[0].item())
    merged_o = np.ones((B0, 144), bool)
    scores_back = np.zeros((1, N, 16, 9), np.float64)
    t0 = time.perf_counter()
    o = 0
    for c in range(min(nchunks, cap.Cmax)):
        n = int(((cell[int(base[c]):int(base[c + 1])] // N) == 0).sum())
        if n == 0:
            continue
        res, _, _, sb = oracle.merge_patches(wl["merge_new"], trust_g[o:o + n], (H, W), masks0[c:c + 1], ifn_g[o:o + n], scores_back)
        scores_back = sb if wl["merge_new"] else np.zeros_like(sb)
        tail = int(third_set[c, 1])
        if tail != 0:
            res[-tail:, :] = True
        merged_o[o:o + n] = res
        o += n
    times["merge"] = time.perf_counter() - t0
    merged_g = cpu(out["merged"][r0t])
    parity["merge_if_nomatching_mismatch"] = int((merged_g != merged_o).sum())

    # ---- L3 (pats.py:53-58, third_layer.py:153-170) on the descriptors the GPU's window gather produced --------------
    P = int(out["P"].item())
    b_ids = cpu(st["b_ids"][:P])
    idx3 = np.nonzero(np.isin(b_ids, rows0))[0]
    i3t = torch.from_numpy(idx3).to(dev)
    P0 = len(idx3)
    t3a, t3b, sc3 = cpu(st["feat0"][i3t]), cpu(st["feat1"][i3t]), cpu(st["scale3"][i3t])
    ps3, pt3 = cpu(st["p_s"][i3t]), cpu(st["p_t"][i3t])
    sq = np.sqrt(sc3 + np.float32(1e-8)).astype(np.float32)
    pts2_0 = cpu(st["pts2"][r0t])
    t0 = time.perf_counter()
    mk0_o, mk1_o, bid_o = oracle.third_inputs(merged_o, pts2_0)
    S3 = oracle.cost(t3a, t3b)
    Z3 = oracle.log_optimal_transport2(S3, 1.0, sc3, ITERS)
    q0, q1, _, qlabel, qifm = oracle.compute_result(np.exp(Z3), sq, sq, ps3, pt3, wl["outdoor"])
    times["L3"] = time.perf_counter() - t0
    g1 = cpu(st["m1f"][i3t])
    glabel = cpu(st["label"].reshape(-1, 16, 2)[i3t])
    parity.update({"l3_problems": P0,
                   "l3_points_mismatch": int((mk0_o != cpu(st["mk0"][i3t])).sum() + (mk1_o != cpu(st["mk1"][i3t])).sum()) if len(mk0_o) == P0 else -1,
                   "l3_label_mismatch": int((glabel.reshape(-1, 2) != qlabel).sum()),
                   "l3_if_matching_mismatch": int((cpu(st["ifm"][i3t]).astype(bool) != qifm.astype(bool)).sum()),
                   "l3_mkpts0_mismatch": int((cpu(st["m0f"][i3t]) != q0).sum()),
                   "l3_mkpts1_max_abs_diff_px": float(np.abs(g1 - q1).max()) if P0 else 0.0})

    # ---- results (pats.py:59-78): the oracle's scatter + get_result on the GPU's third-level output -----------------
    t0 = time.perf_counter()
    ifn16_o, pts16_o = oracle.refine_scatter(merged_o, pts2_0, g1, glabel[:, :, 0].reshape(-1))
    C = masks0.shape[0]
    xs0, av0 = cpu(co["xsn"][0:1]), cpu(co["avn"][0:1])
    xs_c, av_c = np.repeat(xs0, C, axis=0), np.repeat(av0, C, axis=0)
    sc_rows = xs_c[~masks0]
    ml_o, mr_o = oracle.get_result(C, [masks0, ifn16_o], [np.ascontiguousarray(av_c[:, :, ::-1]) / np.float32(32.0),
                                                          np.ascontiguousarray(pts16_o[:, :, ::-1]) / np.float32(2.0)],
                                   [xs_c, np.repeat(sc_rows.reshape(-1, 1, 2), 2304, 1)], [[32, h, w], [2, 48, 48]],
                                   [np.ones(C, bool), np.ones(B0, bool)])
    times["result"] = time.perf_counter() - t0
    ml_g, mr_g = [cpu(t) for t in batch.split_by_pair(out, cap)[0]]
    same_count = ml_g.shape == ml_o.shape
    parity.update({"matches_pair0": int(ml_g.shape[0]), "matches_count_equal": bool(same_count),
                   "matches_l_mismatch": int((ml_g != ml_o).sum()) if same_count else -1,
                   "matches_r_mismatch": int((mr_g != mr_o).sum()) if same_count else -1})
    if same_count and parity["matches_l_mismatch"]:
        sys.stderr.write("matches_l gpu %s\noracle %s\nmatches_r gpu %s\noracle %s\n" % (ml_g[:4], ml_o[:4], mr_g[:4], mr_o[:4]))
    assert parity["l1_if_nomatching_mismatch"] == 0 and parity["l2_flag_mismatch"] == 0, parity
    assert parity["l2_row_argmax_mismatch"] == 0 and parity["l2_col_argmax_mismatch"] == 0, parity
    assert parity["l2_bound_real_mismatch"] == 0 and parity["l2_trust_real_mismatch"] == 0, parity
    assert parity["merge_if_nomatching_mismatch"] == 0 and parity["l3_points_mismatch"] == 0, parity
    assert parity["l3_label_mismatch"] == 0 and parity["l3_if_matching_mismatch"] == 0 and parity["l3_mkpts0_mismatch"] == 0, parity
    assert parity["l3_mkpts1_max_abs_diff_px"] <= 3e-4 * 8 and parity["l2_mass_max_abs_diff"] <= 1e-4, parity
    assert same_count and parity["matches_l_mismatch"] == 0 and parity["matches_r_mismatch"] == 0, parity
    per_pair = sum(times.values())
    if not torch_leg:                                    # the secondary workloads: the oracle's pair + its parity only
        return {"value": 1.0 / per_pair, "unit": "pairs/s", "cores": cores, "kind": "port", "seconds_per_pair": per_pair,
                "sample": "oracle/pats_oracle.c on ONE WHOLE PAIR (pair 0 of a step): L1 %dx%d, %d fine, %d third-level problems"
                          % (N + 1, N + 1, B0, P0), "parity_sample": parity}

    # ---- torch-CPU transcription of what the reference executes (einsum cost + logsumexp sweeps), on samples ----------
    torch.set_num_threads(cores)
    tns = torch.from_numpy(ns)
    t0 = time.perf_counter()
    sco = 0.1 * (torch.einsum("bdn,bdm->bnm", torch.from_numpy(d0), torch.from_numpy(d1)) / 448 ** .5)
    b, m, n = sco.shape
    alpha = torch.tensor(float(nets.alpha.item()))
    coup = torch.cat([torch.cat([sco, alpha.expand(b, m, 1)], -1), alpha.expand(b, 1, n + 1)], 1)
    msn = torch.tensor(float(m))
    norm = -(msn + tns.sum(dim=2)).log()
    log_nu = torch.cat([tns.log()[:, 0] + norm, msn.log().expand(b, 1) + norm], dim=1)
    log_mu = torch.cat([norm.expand(b, m), tns.sum(dim=2).log() + norm], dim=1)
    torch_cpu_sinkhorn(coup, log_mu, log_nu, ITERS)
    tt1 = time.perf_counter() - t0
    # the WHOLE pair, measured (no sampling): every fine problem, every third-level problem the merge left, in the batch
    # sizes the reference issues them in (one chunk of <= 2w rows at a time; the third level chunk by chunk: ~300 problems)
    t0 = time.perf_counter()
    for o in range(0, B0, 40):
        torch_cpu_cost_ot2(torch.from_numpy(f0[o:o + 40]), torch.from_numpy(f1[o:o + 40]), torch.from_numpy((sx * sy)[o:o + 40]), ITERS)
    tt2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    step3 = max(1, -(-P0 // max(nchunks, 1)))
    for o in range(0, P0, step3):
        torch_cpu_cost_ot2(torch.from_numpy(t3a[o:o + step3]), torch.from_numpy(t3b[o:o + step3]), torch.from_numpy(sc3[o:o + step3]), ITERS)
    tt3 = time.perf_counter() - t0
    torch_pair = tt1 + tt2 + tt3
    return {"value": 1.0 / per_pair, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle/pats_oracle.c (OpenMP over problems) on ONE WHOLE PAIR, measured: pair 0 of a step - L1 %dx%d "
                      "(%.3fs), its %d fine problems (%.3fs), the merges of its %d chunks (%.3fs), its %d third-level problems "
                      "(%.3fs), scatter + get_result (%.3fs)" % (N + 1, N + 1, times["L1"], B0, times["L2"], nchunks, times["merge"],
                                                                  P0, times["L3"], times["result"]),
            "seconds_per_pair": per_pair,
            "torch_cpu": {"value": 1.0 / torch_pair, "unit": "pairs/s", "cores": cores,
                          "sample": "measured on ONE WHOLE PAIR (no sampling): torch transcription of the reference's CPU arithmetic (einsum "
                                    "cost builds + modules.py:137-182 logsumexp sweeps; no expansion / merge), %d torch threads: L1 "
                                    "(%.3fs), all %d fine problems in chunks of 40 (%.3fs), all %d third-level problems in %d chunks (%.3fs)"
                                    % (cores, tt1, B0, tt2, P0, max(nchunks, 1), tt3),
                          "seconds_per_pair": torch_pair},
            "parity_sample": parity}
