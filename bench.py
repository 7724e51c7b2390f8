#!/usr/bin/env python3
"""bench.py - image-pairs/sec of the PATS OT hot path on MI355X (BASELINE.json configs[1]).

One STEP = one pass of the hot path over a batch of `--pairs` synthetic 640x480 pairs at the
reference's shapes, inputs resident in HBM before the timed region:
  L1  [1,448,300]^2 cost build (MFMA) -> log_optimal_transport 301x301, 100 sweeps -> column mass
      -> argmax + 15-step area expansion -> split_patches (cap 2w = 40 as `if_local`) -> Compute_imgs
      (bounds, left crops, right crop + bilinear resize = the native tensor_resize)
  L2  [B,264,145]^2 cost -> log_optimal_transport2 145x145, 100 sweeps, +ln2 dustbin
      -> argmax + 8-step expansion
  L3  [60*B,128,65]^2 cost -> log_optimal_transport2 65x65, 100 sweeps -> Compute_result + label
  out refine scatter (pats.py:59-67) + get_result (utils.py:189-213): matches_l / matches_r
The step makes NO host read: the chunk plan is computed on the device (pats_split_patches_device), the
crop gathers and get_result run over their capacity with device-side counts (the reference syncs at
every boolean mask).  Descriptors are synthetic (no weights/datasets exist for the reference here);
P = 60*B is the SURVEY's chosen fill.  `value` = pairs/sec over all ranks (pairs shard across ranks,
no data-path collective; "weak" scaling); after the timed region every rank's matches of its last
step are gathered to rank 0 over RCCL (shard.gather_matches), timed separately as `gather_ms`.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself
(torch.distributed.run on 127.0.0.1); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.
One JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from pats_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA
ITERS = 100
ONE = [None]            # device-resident 1.0 (the reference's `self.one`, second_layer.py:63)


# name -> (grid h, grid w, if_local, outdoor, label); BASELINE.json configs[1..3], shapes from SURVEY.md section 8d
WORKLOADS = {"megadepth": (15, 20, True, True, "configs[1]: MegaDepth 640x480 shapes, outdoor (if_local chunks of 2w, +ln2, label from the dustbin)"),
             "scannet": (15, 20, False, False, "configs[2]: ScanNet 640x480 shapes, indoor (one L2 chunk, cap 512; +ln3; fixed-cell label)"),
             "yfcc": (24, 32, True, True, "configs[3]: YFCC 768x1024 shapes (24x32 grid, 769x769 coarse problem), outdoor")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=48,
                    help="image pairs per step per rank (48 = about 100 GB of synthetic descriptors resident in the 288 GB of HBM)")
    ap.add_argument("--fill", type=int, default=60, help="third-level problems per fine problem (P = fill*B)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="megadepth",
                    help="megadepth = BASELINE configs[1] (the bench line); scannet = configs[2] shapes (indoor rules, one L2 chunk); "
                         "yfcc = configs[3] shapes (768x1024 pairs, 769x769 coarse problem) - secondary measurements")
    ap.add_argument("--per-chunk", action="store_true",
                    help="run Compute_imgs once per coarse chunk like the reference's loop (one host read per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary roofline measurements")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the coarse stage and the fine / third stage of consecutive batches one after the other "
                         "(default: on two HIP streams, the coarse stage of batch i + 1 beside the rest of batch i)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time")
    return ap.parse_args()


def desc_pair(shape, dev, gen, drop=0.0):
    base = torch.randn(shape, device=dev, generator=gen)
    d0 = 3.0 * (base + 0.3 * torch.randn(shape, device=dev, generator=gen))
    d1 = 3.0 * (base + 0.3 * torch.randn(shape, device=dev, generator=gen))
    if drop > 0:
        gone = torch.rand((shape[0], 1, shape[2]), device=dev, generator=gen) < drop
        d0 = torch.where(gone, 3.12 * torch.randn(shape, device=dev, generator=gen), d0)
    return d0.contiguous(), d1.contiguous()


def scale_head(shape, dev, gen):
    x = 0.3 * torch.randn(shape, device=dev, generator=gen)
    return torch.exp(torch.sigmoid(x) * synth.LN256 - synth.LN256 / 2)


class Workload:
    """Device-resident synthetic inputs of `pairs` 640x480 pairs.  Stages are batched ACROSS pairs
    and across the coarse chunks: the reference walks pairs and chunks in Python loops
    (evaluate.py:25, pats.py:33) only because it targets one 16-40 GB card; every coarse / fine /
    third-level problem is independent, so with 288 GB each stage is one launch."""

    def __init__(self, ops, dev, gen, pairs, fill, per_chunk=False, workload="megadepth"):
        h, w, self.if_local, self.outdoor, self.label = WORKLOADS[workload]
        c = synth.coarse_inputs(h=h, w=w) if workload != "megadepth" else synth.coarse_inputs()
        self.H, self.W, self.cap = 32 * h, 32 * w, (2 * w if self.if_local else 512)
        self.bias_k = 2.0 if self.outdoor else 3.0
        self.pairs, self.h, self.w, self.fill = pairs, c["h"], c["w"], fill
        self.per_chunk_imgs = per_chunk
        rep = lambda a: torch.from_numpy(a).to(dev).repeat(pairs, *([1] * (a.ndim - 1))).contiguous()  # noqa: E731
        self.d0, self.d1, self.ns = rep(c["d0"]), rep(c["d1"]), rep(c["ns"])
        self.alpha = torch.tensor(float(c["alpha"]), device=dev)
        left, right = synth.image_pair(H=self.H, W=self.W)
        self.left, self.right = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
        self.lefts = self.left.expand(pairs, -1, -1, -1).contiguous()       # every pair: the same synthetic image
        self.rights = self.right.expand(pairs, -1, -1, -1).contiguous()
        # dry run of the coarse stage (with a host read) to learn the deterministic chunk plan
        self.plan, self.counts = coarse_plan_host(ops, self)
        self.C = len(self.plan)
        B1 = sum(self.plan)
        B = B1 * pairs
        f0, f1 = desc_pair((B, 264, 145), dev, gen, drop=0.12)
        f0[:, :, -1] *= 0.5
        f1[:, :, -1] *= 0.5
        sx, sy = scale_head((B, 1, 144), dev, gen), scale_head((B, 1, 144), dev, gen)
        P = fill * B
        t0, t1 = desc_pair((P, 128, 65), dev, gen, drop=0.12)
        t0[:, :, -1] *= 0.5
        t1[:, :, -1] *= 0.5
        sc = scale_head((P, 1, 64), dev, gen)
        p_s = (torch.randint(1, 23, (P, 2), device=dev, generator=gen) * 4)
        p_t = (torch.randint(0, 25, (P, 2), device=dev, generator=gen) * 4)
        # the merge's output stand-in (merge_patches_* is out of the bench: its chunks couple through
        # scores_back, pats.py:32,37): exactly `fill` surviving L2 cells per row, fixed pattern, so that
        # third-level problem p belongs to the p-th surviving cell as pats.py:53-58 orders them
        keep = torch.zeros((B, 144), dtype=torch.bool, device=dev)
        order = torch.argsort(torch.rand((B, 144), device=dev, generator=gen), dim=1)[:, :fill]
        keep.scatter_(1, order, True)
        self.chunk = dict(B=B, P=P, f0=f0, f1=f1, sx=sx, sy=sy, ns2=(sx * sy).contiguous(), t0=t0, t1=t1, sc=sc,
                          p_s=p_s, p_t=p_t, ifn_L2=torch.logical_not(keep).contiguous())
        self.ones_c = torch.ones((pairs * self.C,), dtype=torch.bool, device=dev)
        self.ones_b = torch.ones((B,), dtype=torch.bool, device=dev)
        self.B, self.P = B1, fill * B1


def coarse_ops(ops, wl):
    """first_layer.py:110-135 for all pairs: one batched cost+OT launch, column mass, argmax + expansion."""
    Z = ops.cost_ot(wl.d0, wl.d1, 1, wl.alpha, wl.ns, ITERS)
    scales, cflag = ops.colmass_sqrt(Z, return_flags=True)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_first(Z, scales, (wl.H, wl.W), 32, col_nomatch=cflag)
    sum_cycle = torch.cumsum(torch.logical_not(ifn1).int(), dim=1, dtype=torch.int32)
    return pts, xs, ys, ifn1, sum_cycle


def coarse_plan_host(ops, wl):
    """The chunk plan of pair 0 on the HOST (set-up / --per-chunk only): per-chunk row counts, crops per pair."""
    pts, xs, ys, ifn1, sum_cycle = coarse_ops(ops, wl)
    sc_host = sum_cycle.to("cpu").numpy()
    plans = []
    for i in range(wl.pairs):
        n, second, third = ops.split_patches(sc_host[i], wl.h, wl.w, wl.cap)
        K = int(sc_host[i, -1])
        plans.append([min(hi, K) - lo for lo, hi in second])
    assert all(p == plans[0] for p in plans)
    return plans[0], [int(sc_host[i, -1]) for i in range(wl.pairs)]


def coarse_stage(ops, wl):
    """first_layer.py:110-146 for all pairs, no host read: OT + expansion, the chunk plan on the device, and
    ONE subdivision gather for the whole step (Compute_imgs takes a batch of images, crops ordered
    (image, patch); chunk c of pair i is a contiguous run of it because the cumsum is monotone -
    tests/test_gpu_parity.py::test_chunk_crops_are_slices, ::test_compute_imgs_batch_of_images)."""
    pts, xs, ys, ifn1, sum_cycle = coarse_ops(ops, wl)
    if wl.per_chunk_imgs:
        # the reference's loop (first_layer.py:136-146): host plan, one Compute_imgs per pair and chunk mask
        sc_host = sum_cycle.to("cpu").numpy()
        for i in range(wl.pairs):
            n, second, third = ops.split_patches(sc_host[i], wl.h, wl.w, wl.cap)
            for lo, hi in second:
                mask = torch.logical_or(ifn1[i:i + 1], torch.logical_or(sum_cycle[i:i + 1] <= lo,
                                                                        sum_cycle[i:i + 1] > hi))
                ops.Compute_imgs(xs[i:i + 1], ys[i:i + 1], pts[i:i + 1], mask, wl.left, wl.right, width=wl.w,
                                 height=wl.h)
        num, second, third = ops.split_patches_device(sum_cycle, wl.h, wl.w, wl.cap)
        nl, nr, xsn, ysn, avn = ops.Compute_imgs(xs, ys, pts, ifn1, wl.lefts, wl.rights, width=wl.w, height=wl.h,
                                                 known_count=wl.counts)
        return dict(ifn1=ifn1, sum_cycle=sum_cycle, second=second, num=num, xsn=xsn, avn=avn)
    num, second, third = ops.split_patches_device(sum_cycle, wl.h, wl.w, wl.cap)
    nl, nr, xsn, ysn, avn, bound5, K_img, K_tot = ops.Compute_imgs_ex(xs, ys, pts, ifn1, wl.lefts, wl.rights,
                                                                      width=wl.w, height=wl.h, known_count="device")
    return dict(ifn1=ifn1, sum_cycle=sum_cycle, second=second, num=num, xsn=xsn, avn=avn, K_img=K_img)


def fine_and_third(ops, wl, co, ev):
    ch = wl.chunk
    if ev is not None:
        f0_, f1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0_.record()
    Z2, cflag2 = ops.cost_ot(ch["f0"], ch["f1"], 2, ONE[0], ch["ns2"], ITERS, bias_k=wl.bias_k, return_flags=True)
    if ev is not None:
        f1_.record()
        ev["fine"].append((f0_, f1_))
    trust2, pts2, _, _, ifn_L2, _ = ops.est_position_second(Z2, ch["sx"], ch["sy"], [96, 96], 8, col_nomatch=cflag2)
    # third level: cost build + OT + exp + Compute_result + label in ONE launch (the dominant kernel)
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    m0f, m1f, label, ifm = ops.third_level(ch["t0"], ch["t1"], ch["sc"], ch["p_s"], ch["p_t"], outdoor=wl.outdoor, iters=ITERS)
    if ev is not None:
        e1.record()
        ev["third"].append((e0, e1, ch["P"]))
    # results (pats.py:59-78): third-level matches scattered onto the 48x48 sub-cell grid, then get_result
    # with the (pair, chunk) masks of first_layer.py:137-138 as its level-0 batch - all on the device
    ifn16, pts16 = ops.refine_scatter(ch["ifn_L2"], pts2, m1f, label)
    C, N = wl.C, wl.h * wl.w
    lo, hi = co["second"][:, :C, 0:1], co["second"][:, :C, 1:2]                    # [pairs,C,1]
    sc3 = co["sum_cycle"][:, None, :]
    masks = torch.logical_or(co["ifn1"][:, None, :], torch.logical_or(sc3 <= lo, sc3 > hi)).reshape(-1, N)
    xs_c = co["xsn"][:, None].expand(-1, C, -1, -1).reshape(-1, N, 2)
    av_c = co["avn"][:, None].expand(-1, C, -1, -1).reshape(-1, N, 2)
    # pats.py:70 `x_scale_new[~mask]` without its host sync: a stable sort lists the unmasked cells in order
    cells = torch.argsort(masks.reshape(-1).to(torch.uint8), stable=True)[:ch["B"]]
    sc_rows = xs_c.reshape(-1, 2)[cells]
    ml, mr, M = ops.get_result(wl.pairs * C, [masks, ifn16], [av_c.flip(dims=[2]) / 32.0, pts16.flip(dims=[2]) / 2.0],
                               [xs_c.contiguous(), sc_rows], [[32, wl.h, wl.w], [2, 48, 48]], [wl.ones_c, wl.ones_b],
                               validate=False, sync=False)
    return dict(ml=ml, mr=mr, M=M, masks=masks, ifn16=ifn16, label=label, ifm=ifm, m1f=m1f)


def step(ops, wl, ev):
    co = coarse_stage(ops, wl)
    return co, fine_and_third(ops, wl, co, ev)


def run_steps(ops, wl, ev, n, streams):
    """n complete steps (batches).  streams = None: one after the other.  streams = (sA, sB): the steps of consecutive
    batches are independent (pairs are), so the coarse stage of batch i + 1 (one-CU Sinkhorn kernels on 48 of 256 CUs,
    HBM-bound crop gathers) runs on sA beside the fine + third stage of batch i (VALU-bound) on sB.  Every batch still
    goes through every kernel; nothing leaves the function unfinished (the caller's stream waits for both)."""
    if streams is None or n <= 0:
        co = out = None
        for _ in range(n):
            co, out = step(ops, wl, ev)
        return co, out
    sA, sB = streams
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur)
    sB.wait_stream(cur)

    def coarse():
        with torch.cuda.stream(sA):
            c = coarse_stage(ops, wl)
            e = torch.cuda.Event()
            e.record(sA)
        return c, e
    co, done = coarse()
    out = None
    for i in range(n):
        nxt = coarse() if i + 1 < n else None
        with torch.cuda.stream(sB):
            sB.wait_event(done)
            for v in co.values():                       # allocated on sA, read on sB
                if isinstance(v, torch.Tensor):
                    v.record_stream(sB)
            out = fine_and_third(ops, wl, co, ev)
        last_co = co
        if nxt is not None:
            co, done = nxt
    cur.wait_stream(sA)
    cur.wait_stream(sB)
    return last_co, out


def local_matches(wl, co, out, rank, world):
    """Per-pair (matches_l, matches_r) of this rank's last step, outside the clock: get_result emits rows in
    (pair, chunk, patch, sub-cell) order, so pair i owns a contiguous run whose length is the number of
    surviving sub-cells of its rows."""
    C = wl.C
    rows_per_mask = torch.logical_not(out["masks"]).sum(dim=1).reshape(wl.pairs, C).sum(dim=1).cpu().tolist()
    per_row = torch.logical_not(out["ifn16"]).sum(dim=1).cpu().numpy()
    M = int(out["M"].item())
    res, r0, m0 = [], 0, 0
    for i in range(wl.pairs):
        k = int(per_row[r0:r0 + rows_per_mask[i]].sum())
        res.append((rank + i * world, out["ml"][m0:m0 + k], out["mr"][m0:m0 + k]))
        r0 += rows_per_mask[i]
        m0 += k
    assert m0 == M, "get_result count %d != per-pair total %d" % (M, m0)
    return res


def torch_cpu_sinkhorn(Z, log_mu, log_nu, iters):
    """What the reference executes on CPU (modules.py:137-143), transcribed: logsumexp row / column sweeps."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def torch_cpu_ot2(scores, ns, iters):
    b, m, n = scores.shape
    ms = torch.tensor(float(m - 1))
    nssum = ns.sum(dim=2)                                             # [b,1]
    norm = -(ms + nssum).log()
    log_nu = torch.cat([ns.log()[:, 0] + norm, ms.log().expand(b, 1) + norm], dim=1)
    log_mu = torch.cat([norm.expand(b, m - 1), nssum.log() + norm], dim=1)
    return torch_cpu_sinkhorn(scores, log_mu, log_nu, iters) - norm[:, :, None]


ULP4 = 4.0 * 2.0 ** -23     # "a threshold tie": the deciding quantities agree to 4 ulp


def expansion_parity(ops, oracle, dev, f, gZ2, g2, Z2, ex2, r2, c2):
    """Area expansion of the L2 sample (utils.py:1213-1243), HIP against the oracle, every differing row classified.

    (1) SAME INPUT: the oracle expands the plan the GPU produced (exp on the GPU, the identical fp32 array on both
        sides), so the only freedom left is the summation order of a strip.  A row whose rectangle differs is a
        threshold tie if the oracle's own decision margin - the relative distance between the strip sum that decided
        and `lower_bound` / the competing strip (oracle_iterative_expand_margin) - is within 4 ulp; anything else is a
        REAL mismatch, and the bench asserts there is none.
    (2) END TO END: each side expands its OWN plan.  The plans agree to the 1e-4 transport-mass gate, not bit for bit,
        and `lower_bound` = 1e-3 is only 10x that gate, so a strip sum that lands within the measured plan difference
        of the threshold grows on one side and not on the other; such a row then carries a different rectangle AND a
        different trust score (whole_cost) - this is where round 2's unexplained max |d trust| = 0.04 came from.
        A differing row is "explained" if its margin is below what the measured plan difference of its problem can
        move a strip sum by (12 cells x max |dP|, relative to lower_bound); anything else is REAL and asserted zero."""
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    nb = gZ2.shape[0]
    pos, rng_ = ops.Compute_positions_and_ranges(12, 12, dev)
    gP = ops.exp(gZ2)
    gsame = ops.Iterative_expand_matrix(gP, td(f["scale_x"]).reshape(nb, -1, 1), td(f["scale_y"]).reshape(nb, -1, 1),
                                        [0, 12, 0, 12], rng_, pos, lower_bound=1e-3, iter_num=8, width=12, height=12)
    gbound = gsame[5].cpu().numpy()
    gtrust = gsame[0].cpu().numpy()
    osame = oracle.iterative_expand(gP.cpu().numpy(), f["scale_x"], f["scale_y"], 12, 12, 12, 1e-3, 8, with_margin=True)
    diff_rows = (gbound != osame[5]).any(axis=2)
    margin = osame[6][..., 0]
    tie = diff_rows & (margin <= ULP4)
    real_same = diff_rows & ~tie
    # trust of rows whose rectangles agree: same strips, same sums up to order
    same_rows = ~diff_rows
    # the per-element `> lower_bound` test of :1225 only feeds whole_cost: a row with an element within 4 ulp of the
    # threshold may differ in trust although its rectangle agrees
    elem_tie = osame[6][..., 1] <= ULP4
    dtrust = np.abs(gtrust - osame[0])
    trust_same = float(dtrust[same_rows & ~elem_tie].max()) if (same_rows & ~elem_tie).any() else 0.0
    # end to end (each side its own plan)
    ebound_diff = (gbound != ex2[5]).any(axis=2)
    dP = np.abs(np.exp(gZ2.cpu().numpy().astype(np.float64)) - np.exp(Z2.astype(np.float64)))[:, :-1, :].max(axis=(1, 2))
    own = oracle.iterative_expand(np.exp(Z2), f["scale_x"], f["scale_y"], 12, 12, 12, 1e-3, 8, with_margin=True)
    reach = (12.0 * dP / 1e-3 + ULP4)[:, None]
    explained = ebound_diff & (own[6][..., 0] <= reach)
    real_e2e = ebound_diff & ~explained
    dtrust_e2e = np.abs(g2[0].cpu().numpy() - ex2[0])
    agree = ~ebound_diff & (own[6][..., 1] > reach)
    # a row whose trust differs although its rectangle agrees: the per-element test `expand_sum > lower_bound` of :1225
    # decides whether the opposite dustbin mass of a strip cell is charged to the row (whole_cost = ... + nomatching / 4),
    # and a cell whose mass is within the plan difference of lower_bound is charged on one side only
    tol_t = 1e-4 + 1e-4 * np.abs(ex2[0])
    tdiff = (dtrust_e2e > tol_t) & ~ebound_diff
    t_explained = tdiff & (own[6][..., 1] <= reach)
    ifn1 = g2[4].cpu().numpy()
    ifn2 = g2[5].cpu().numpy()
    flag_mismatch = int((ifn1 != (r2[:, :-1] == 144)).sum() + (ifn2 != (c2[:, :-1] == 144)).sum())
    return {
        "l2_rows": int(diff_rows.size),
        "l2_bound_mismatch_same_input": int(diff_rows.sum()), "l2_bound_threshold_ties": int(tie.sum()),
        "l2_bound_real_mismatch": int(real_same.sum() + real_e2e.sum()),
        "l2_trust_max_abs_diff_same_input_same_rectangle": trust_same,
        "l2_bound_mismatch_end_to_end": int(ebound_diff.sum()),
        "l2_bound_mismatch_end_to_end_explained_by_plan_difference": int(explained.sum()),
        "l2_plan_max_abs_diff": float(dP.max()),
        "l2_smallest_margin_of_a_differing_row": float(own[6][..., 0][ebound_diff].min()) if ebound_diff.any() else None,
        "l2_trust_max_abs_diff": float(dtrust_e2e.max()), "l2_trust_max_abs": float(np.abs(ex2[0]).max()),
        "l2_trust_max_abs_diff_where_rectangles_agree": float(dtrust_e2e[agree].max()) if agree.any() else 0.0,
        "l2_trust_rows_differing_with_equal_rectangles": int(tdiff.sum()),
        "l2_trust_rows_explained_by_element_threshold": int(t_explained.sum()),
        "l2_trust_real_mismatch": int((tdiff & ~t_explained).sum()),
        "l2_flag_mismatch": flag_mismatch,
    }


def cpu_baseline(ops, dev, pairs_B, pairs_P, seconds):
    """The CPU oracle ("port") on the host cores: L1 in full, bounded samples of L2/L3 scaled to one
    pair, plus the torch-CPU transcription of the Sinkhorn loop on the same samples.  Checker code timed
    as a baseline only - never part of the measured GPU path.  The HIP path is also run on the same L2 /
    L3 samples and compared with the oracle's answers (`parity_sample`): index outputs and expansion rectangles are
    ASSERTED - every differing row must classify as a threshold tie (expansion_parity)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import pats_oracle as oracle
    cores = oracle.num_threads()
    c = synth.coarse_inputs()
    t0 = time.perf_counter()
    S = oracle.cost(c["d0"], c["d1"])
    Z = oracle.log_optimal_transport(S, c["alpha"], c["ns"], ITERS)
    sc = oracle.colmass_sqrt(Z)
    oracle.argmax(Z)
    oracle.iterative_expand(np.exp(Z), sc, sc, 20, 15, 20, 1e-5, 15)
    t_l1 = time.perf_counter() - t0
    # L2 sample
    nb = max(cores, 8)
    f = synth.fine_inputs(seed=77, B=nb)
    t0 = time.perf_counter()
    S2 = oracle.cost(f["d0"], f["d1"])
    Z2 = oracle.dustbin_bias(oracle.log_optimal_transport2(S2, 1.0, f["scale_x"] * f["scale_y"], ITERS), 2.0)
    r2, c2 = oracle.argmax(Z2)
    ex2 = oracle.iterative_expand(np.exp(Z2), f["scale_x"], f["scale_y"], 12, 12, 12, 1e-3, 8)
    t_l2 = (time.perf_counter() - t0) / nb
    # L3 sample sized to the remaining budget
    probe = synth.third_inputs(seed=78, P=4 * cores)
    t0 = time.perf_counter()
    S3 = oracle.cost(probe["d0"], probe["d1"])
    oracle.log_optimal_transport2(S3, 1.0, probe["scale"], ITERS)
    per = (time.perf_counter() - t0) / (4 * cores)
    np3 = int(max(4 * cores, min(4096, (seconds - t_l1 - t_l2 * nb) / max(per, 1e-6))))
    t3in = synth.third_inputs(seed=79, P=np3)
    sq = np.sqrt(t3in["scale"] + np.float32(1e-8)).astype(np.float32)
    t0 = time.perf_counter()
    S3 = oracle.cost(t3in["d0"], t3in["d1"])
    Z3 = oracle.log_optimal_transport2(S3, 1.0, t3in["scale"], ITERS)
    r0, r1, rwl, rlabel, rifm = oracle.compute_result(np.exp(Z3), sq, sq, t3in["p_s"], t3in["p_t"], True)
    t_l3 = (time.perf_counter() - t0) / np3
    per_pair = t_l1 + t_l2 * pairs_B + t_l3 * pairs_P

    # the same samples through the HIP path: indices must be identical, transport mass within 1e-4
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    g0, g1, glabel, gifm = ops.third_level(td(t3in["d0"]), td(t3in["d1"]), td(t3in["scale"]), td(t3in["p_s"]),
                                           td(t3in["p_t"]), outdoor=True, iters=ITERS)
    gZ2 = ops.cost_ot(td(f["d0"]), td(f["d1"]), 2, 1.0, td(f["scale_x"] * f["scale_y"]), ITERS, bias_k=2.0)
    g2 = ops.est_position_second(gZ2, td(f["scale_x"]), td(f["scale_y"]), [96, 96], 8)
    gr2, gc2 = ops.argmax(gZ2)
    e2, e2r = np.exp(gZ2.cpu().numpy().astype(np.float64)), np.exp(Z2.astype(np.float64))
    parity = {
        "l3_problems": np3, "l3_label_mismatch": int((glabel.cpu().numpy() != rlabel).sum()),
        "l3_if_matching_mismatch": int((gifm.cpu().numpy().astype(bool) != rifm.astype(bool)).sum()),
        "l3_mkpts0_mismatch": int((g0.cpu().numpy() != r0).sum()),
        "l3_mkpts1_max_abs_diff_px": float(np.abs(g1.cpu().numpy() - r1).max()),
        "l2_problems": nb, "l2_row_argmax_mismatch": int((gr2.cpu().numpy() != r2).sum()),
        "l2_col_argmax_mismatch": int((gc2.cpu().numpy() != c2).sum()),
        "l2_mass_max_abs_diff": float(np.abs(e2[:, :-1, :-1] - e2r[:, :-1, :-1]).max()),
    }
    parity.update(expansion_parity(ops, oracle, dev, f, gZ2, g2, Z2, ex2, r2, c2))
    assert parity["l2_bound_real_mismatch"] == 0 and parity["l2_flag_mismatch"] == 0, parity
    assert parity["l2_trust_real_mismatch"] == 0, parity
    assert parity["l3_label_mismatch"] == 0 and parity["l3_if_matching_mismatch"] == 0 and parity["l3_mkpts0_mismatch"] == 0, parity

    # torch-CPU transcription of modules.py:137-182 on the same L1 problem and (smaller) L2 / L3 samples
    torch.set_num_threads(cores)
    tS = torch.from_numpy(S)
    tns = torch.from_numpy(c["ns"])
    t0 = time.perf_counter()
    b, m, n = tS.shape
    alpha = torch.tensor(float(c["alpha"]))
    coup = torch.cat([torch.cat([tS, alpha.expand(b, m, 1)], -1), alpha.expand(b, 1, n + 1)], 1)
    msn = torch.tensor(float(m))
    norm = -(msn + tns.sum(dim=2)).log()
    log_nu = torch.cat([tns.log()[:, 0] + norm, msn.log().expand(b, 1) + norm], dim=1)
    log_mu = torch.cat([norm.expand(b, m), tns.sum(dim=2).log() + norm], dim=1)
    torch_cpu_sinkhorn(coup, log_mu, log_nu, ITERS)
    tt1 = time.perf_counter() - t0
    n2 = min(nb, 64)
    t0 = time.perf_counter()
    torch_cpu_ot2(torch.from_numpy(S2[:n2]), torch.from_numpy((f["scale_x"] * f["scale_y"])[:n2]), ITERS)
    tt2 = (time.perf_counter() - t0) / n2
    n3 = min(np3, 1024)
    t0 = time.perf_counter()
    torch_cpu_ot2(torch.from_numpy(S3[:n3]), torch.from_numpy(t3in["scale"][:n3]), ITERS)
    tt3 = (time.perf_counter() - t0) / n3
    torch_pair = tt1 + tt2 * pairs_B + tt3 * pairs_P
    return {"value": 1.0 / per_pair, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle/pats_oracle.c (OpenMP over problems), EXTRAPOLATED from samples: L1 301x301 in full "
                      "(%.3fs), %d L2 problems (%.4fs each), %d L3 problems (%.5fs each), scaled to B=%d, P=%d per pair"
                      % (t_l1, nb, t_l2, np3, t_l3, pairs_B, pairs_P),
            "torch_cpu": {"value": 1.0 / torch_pair, "unit": "pairs/s", "cores": cores,
                          "sample": "torch.logsumexp transcription of modules.py:137-182 (Sinkhorn only, no cost build / "
                                    "expansion), %d torch threads, EXTRAPOLATED: L1 301x301 (%.3fs), %d L2 (%.4fs each), "
                                    "%d L3 (%.5fs each) scaled to B=%d, P=%d" % (cores, tt1, n2, tt2, n3, tt3, pairs_B, pairs_P)},
            "parity_sample": parity}


def timed(fn, reps=5, warm=2):
    """Mean duration of one call: one HIP event pair around `reps` back-to-back calls on the stream the kernels run on.
    For a kernel of tens of microseconds the calls must not allocate (pass out=) and reps must be large enough for
    the queue to stay ahead of the GPU - an event pair per call adds ~35 us of marker latency to each."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def secondary_rooflines(ops, dev, wl, fine_ms):
    """Other kernels of the path against their nearer roofline (live HIP-event timings; rocprof counterparts
    under profiles/r02_*).  Config 5 = BASELINE.json configs[4]."""
    out = []
    r = synth.roofline_inputs()
    d0, d1, ns = [torch.from_numpy(r[k]).to(dev) for k in ("d0", "d1", "ns")]
    N, D = d0.shape[2], d0.shape[1]
    S = ops.cost(d0, d1)
    ms = timed(lambda: ops.cost(d0, d1, out=S), reps=200, warm=20)
    tf = 2.0 * D * N * N / (ms * 1e-3) / 1e12
    split_note = ("fp32 operands as fp16 hi + lo pairs, three exact-product passes of v_mfma_f32_32x32x16_f16 (fp32 accumulation; "
                  "closer to float64 than the fp32 fma chain, tools/cost_ab.py): `achieved` counts the 2*D*M*N algorithmic flops "
                  "against the fp32 matrix peak the reference arithmetic would be priced at; the fp16 pipe executes three times "
                  "as many (f16_pipe_*).  PATS_COST_F32=1 = the fp32-MFMA path (profiles/r02_cost_ab.txt)")
    out.append({"kernel": "cost_mfma_kernel, config 5 (4096^2 x %d)" % D, "bound": "mfma", "achieved": tf,
                "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F32_PEAK_TFLOPS, "ms": ms,
                "f16_pipe_tflops": 3.0 * tf, "f16_pipe_frac": 3.0 * tf / F16_PEAK_TFLOPS, "note": split_note,
                "profile": "profiles/r02_config5_kernel_stats.md"})
    S = ops.cost(d0, d1)
    alpha = torch.tensor(float(r["alpha"]), device=dev)
    iters5 = 200
    ms = timed(lambda: ops.log_optimal_transport(S, alpha, ns, iters5), reps=3, warm=1)
    M = N + 1
    gbs = 8.0 * M * M * iters5 / (ms * 1e-3) / 1e9
    out.append({"kernel": "stream_sweep_kernel, config 5 (4097^2, %d sweeps)" % iters5, "bound": "hbm", "achieved": gbs,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "ms": ms,
                "sweeps_per_s": iters5 / (ms * 1e-3),
                "note": "algorithmic 8*M*N bytes per sweep; the 67 MB matrix is Infinity-Cache (256 MiB) resident, so "
                        "this can exceed the DRAM roofline - labelled, not a DRAM claim",
                "profile": "profiles/r02_config5_kernel_stats.md"})
    del S, d0, d1
    ch = wl.chunk
    S2 = ops.cost(ch["f0"], ch["f1"])
    ms = timed(lambda: ops.cost(ch["f0"], ch["f1"], out=S2), reps=8, warm=2)
    del S2
    tf = 2.0 * 264 * 145 * 145 * ch["B"] / (ms * 1e-3) / 1e12
    by = (2.0 * 264 * 145 * 4 + 145 * 145 * 4) * ch["B"]
    out.append({"kernel": "cost_mfma_kernel, fine level (%d x [264,145]^2)" % ch["B"], "bound": "hbm",
                "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ms": ms, "tflops": tf, "note": "one workgroup per problem reads 306 KB of descriptors and writes 84 KB of scores: "
                "with the fp16-split contraction the matrix pipe needs 0.4 ms of this, the rest is the descriptor stream"})
    if fine_ms is not None:
        by = (2.0 * 264 * 145 * 4 + 145 * 145 * 4) * ch["B"]
        gbs = by / (fine_ms * 1e-3) / 1e9
        out.append({"kernel": "fine-level cost + Sinkhorn (%d x 145x145, descriptors in, log-plan out)" % ch["B"],
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "ms": fine_ms, "sweep_elements_per_s": 2.0 * ITERS * 145 * 145 * ch["B"] / (fine_ms * 1e-3),
                    "valu_frac": 2.0 * 2.0 * ITERS * 145 * 145 * ch["B"] / (fine_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS})
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU over RCCL
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # plumbing-test knobs (a 1-GPU box cannot host two RCCL ranks): PATS_BENCH_SHARE_DEVICE=1 maps every
    # rank onto the visible devices modulo their count, PATS_BENCH_BACKEND=gloo swaps the backend.
    # Neither is set by the driver; numbers from such a run are not bench lines.
    if os.environ.get("PATS_BENCH_SHARE_DEVICE"):
        local_rank %= torch.cuda.device_count()
    backend = os.environ.get("PATS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    from pats_amd import ops, shard

    gen = torch.Generator(device=dev)
    gen.manual_seed(synth.SEED + rank)
    ONE[0] = torch.tensor(1.0, device=dev)
    wl = Workload(ops, dev, gen, args.pairs, args.fill, args.per_chunk, args.workload)
    B, P = wl.B, wl.P

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    streams = None if (args.no_overlap or args.per_chunk) else (torch.cuda.Stream(), torch.cuda.Stream())
    run_steps(ops, wl, None, args.warmup, streams)
    ev = {"third": [], "fine": []}
    barrier()
    ops.sinkhorn_fallbacks(reset=True)
    t0 = time.perf_counter()
    co, out = run_steps(ops, wl, ev, args.steps, streams)
    barrier()
    dt = time.perf_counter() - t0
    fallbacks = ops.sinkhorn_fallbacks(reset=True)       # after the timed region (it synchronises)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_gpus = dist.get_world_size() if dist is not None else 1
    total_pairs = args.pairs * args.steps * n_gpus
    value = total_pairs / dt

    # the path's only exchange: the last step's matches of every rank -> rank 0 (RCCL), outside the clock
    local = local_matches(wl, co, out, rank, n_gpus)
    barrier()
    t0 = time.perf_counter()
    gathered = shard.gather_matches(local, args.pairs * n_gpus)
    barrier()
    gather_ms = 1e3 * (time.perf_counter() - t0)
    matches_per_pair = None
    if rank == 0:
        assert all(g is not None for g in gathered), "gather_matches lost a pair"
        matches_per_pair = float(np.mean([g[0].shape[0] for g in gathered]))
        gather_bytes = sum(g[0].shape[0] for g in gathered) * 16

    # dominant kernel: the 65x65 third-level launch (HIP events on the launch stream)
    ms = np.array([a.elapsed_time(b) for a, b, _ in ev["third"]])
    probs = np.array([p for _, _, p in ev["third"]], dtype=np.float64)
    fine_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fine"]]))
    # algorithmic HBM bytes per problem of the fused third-level kernel: both descriptor blocks in
    # (2 x 128 x 65 fp32), areas + coarse points in, 16 matches + labels + flags out; the 65x65 plan
    # stays on chip (SURVEY 8d "cost build: 4*D*(M+N) in, 0 out if fused")
    BYTES_PER_PROBLEM = 2 * 128 * 65 * 4 + 64 * 4 + 2 * 2 * 8 + 2 * 16 * 2 * 4 + 16 * 2 * 4 + 16
    alg_bytes = float(BYTES_PER_PROBLEM) * probs
    achieved = float((alg_bytes / (ms * 1e-3)).mean() / 1e9)
    exp_rate = float((2.0 * ITERS * 65 * 65 * probs / (ms * 1e-3)).mean())
    # fp32 VALU work of the kernel as flops: 2 half-sweeps x 65 x 65 FMAs (2 flops) per sweep per problem
    valu_tflops = float((2.0 * 2.0 * ITERS * 65 * 65 * probs / (ms * 1e-3)).mean() / 1e12)
    sweeps_per_pair = ITERS * (1 + B + P)        # one sweep = row + column normalisation of one problem

    # HBM traffic of the dominant kernel from rocprofv3 PMC passes (collected separately, see the file)
    traffic, traffic_src = None, None
    for name in ("r02_pmc_third.json", "r01_pmc_third.json"):
        pmc_path = os.path.join(REPO, "profiles", name)
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            traffic = float(pmc["hbm_bytes_per_problem"]) * float(probs.mean())
            traffic_src = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH calibrated " \
                          "x%.2f on cost65_kernel's known byte count; per problem x problems per launch" \
                          % (name, pmc["fetch_calibration"]["factor"])
            break

    res = {
        "metric": "image-pairs/sec (coarse+fine OT) on 640x480 MegaDepth; OT iters/sec per pair",
        "value": value, "unit": "pairs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl.label + ": coarse+fine+third OT + cost volume + expansion + subdivision gather + get_result",
                   "pairs_per_step_per_rank": args.pairs,
                   "batching": "each stage is one launch over all pairs of the step; no host read inside a step"
                               + ("" if streams is None else "; the coarse stage of batch i + 1 runs on a second HIP stream "
                                  "beside the fine / third stage of batch i (--no-overlap: one after the other)"),
                   "L1": "1x[448,%d]^2 -> %dx%d" % (wl.h * wl.w, wl.h * wl.w + 1, wl.h * wl.w + 1),
                   "L2": "%d x [264,145]^2 -> 145x145 (%d coarse chunks, batched into one launch)" % (B, len(wl.plan)),
                   "L3": "%d x [128,65]^2 -> 65x65 (fill %d)" % (P, args.fill), "sinkhorn_iters": ITERS,
                   "parallelism": "pairs sharded over %d rank(s), no data-path collective; matches gathered to rank 0 "
                                  "after the timed region (%s)" % (n_gpus, backend if dist is not None else "single process")},
        "ot_iters_per_sec": value * sweeps_per_pair,
        "guard_fallbacks_per_step": fallbacks / max(1, args.steps),
        "gather_ms": gather_ms, "matches_per_pair": matches_per_pair,
        "roofline": {"bound": "hbm", "kernel": "third_fused_kernel (fused third level, %d problems per launch)" % wl.chunk["P"],
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": float(alg_bytes.mean()),
                     "avg_launch_ms": float(ms.mean()), "launches": int(len(ms)),
                     "algorithmic_bytes_per_problem": BYTES_PER_PROBLEM,
                     "sweep_elements_per_s": exp_rate,
                     "valu_frac": valu_tflops / F32_PEAK_TFLOPS, "valu_tflops": valu_tflops,
                     "mfma_flops_per_s": float((2.0 * 128 * 64 * 64 * probs / (ms * 1e-3)).mean()),
                     "note": "fused cost build (fp16-split operands, three exact-product f16 MFMA passes, fp32 accumulation: error vs "
                             "float64 below the fp32 fma chain's; PATS_THIRD_VARIANT=300 = fp32 MFMA) + 100 linear-domain Sinkhorn sweeps + Compute_result per 65x65 problem, "
                             "one wave each, the 65x65 block held in registers; descriptors are read once, the plan never "
                             "reaches HBM.  HBM is the nearest of the two allowed rooflines but not the limiter: the sweeps "
                             "are fp32 VALU work (valu_frac = sweep FMA flops / 157.3 TF/s vector peak)"},
    }
    if rank == 0:
        if n_gpus > 1:
            res["gather_bytes"] = gather_bytes
        if not args.no_secondary and n_gpus == 1:
            res["roofline_secondary"] = secondary_rooflines(ops, dev, wl, fine_ms)
        if not args.no_cpu_baseline and n_gpus == 1:
            res["cpu_baseline"] = cpu_baseline(ops, dev, B, P, args.cpu_seconds)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
