"""bench.py: the secondary legs of the default line - rooflines of the other kernels and of config 5, the GNN layers, gather layouts,
step determinism, guard-trip sweep, and BASELINE configs[2] / [3] in the same run."""
import json  # noqa: F401
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import REPO, synth  # noqa: F401
from .baseline import cpu_baseline
from .nets import BenchNets, StepWatch, run_steps


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


def secondary_rooflines(ops, dev):
    """Kernels of BASELINE.json configs[4] (config 5 of SURVEY 8d) against their nearer roofline (live HIP-event timings;
    rocprof counterparts under profiles/)."""
    res = []
    r = synth.roofline_inputs()
    d0, d1, ns = [torch.from_numpy(r[k]).to(dev) for k in ("d0", "d1", "ns")]
    N, D = d0.shape[2], d0.shape[1]
    S = ops.cost(d0, d1)
    ms = timed(lambda: ops.cost(d0, d1, out=S), reps=200, warm=20)
    tf = 2.0 * D * N * N / (ms * 1e-3) / 1e12
    res.append({"kernel": "cost_mfma_kernel, config 5 (4096^2 x %d)" % D, "bound": "mfma", "achieved": 3.0 * tf,
                "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 3.0 * tf / F16_PEAK_TFLOPS, "ms": ms,
                "algorithmic_tflops": tf, "fp32_equivalent_frac": tf / F32_PEAK_TFLOPS,
                "note": "priced on the pipe the kernel uses: fp32 operands as fp16 hi + lo pairs, THREE exact-product passes of "
                        "v_mfma_f32_32x32x16_f16 per tile (fp32 accumulation) = 3 x the 2*D*M*N algorithmic flops against the dense fp16 "
                        "matrix peak; the limiter is the descriptor stream, the LDS staging and the VALU split, not the matrix pipe.  "
                        "fp32_equivalent_frac = algorithmic flops against the 157.3 TF/s fp32 matrix peak the reference arithmetic "
                        "would be priced at (a note, not the claim)"})
    alpha = torch.tensor(float(r["alpha"]), device=dev)
    iters5 = 200
    ms = timed(lambda: ops.log_optimal_transport(S, alpha, ns, iters5), reps=3, warm=1)
    M = N + 1
    gbs = 8.0 * M * M * iters5 / (ms * 1e-3) / 1e9
    # match indices against the REFERENCE's own 4097 x 4097, 200-sweep run (tests/golden/roofline_4097.npz holds both argmax vectors):
    # an index may differ only where the two candidates' log-plan values agree to 4 ulp (flat N(0, 0.01) scores: exact-noise ties)
    ties = None
    gpath = os.path.join(REPO, "tests", "golden", "roofline_4097.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        Z = ops.log_optimal_transport(S, alpha, ns, int(g["iters"]))
        rr, cc = ops.argmax(Z)
        Zc = Z[0].cpu().numpy()

        def flips(Zn, got, want):
            bad = np.nonzero(got != want)[0]
            real = sum(1 for i in bad if abs(float(Zn[i, got[i]]) - float(Zn[i, want[i]])) >
                       4 * np.spacing(np.float32(max(abs(Zn[i, got[i]]), abs(Zn[i, want[i]])))))
            return int(len(bad)), int(real)
        (nr, real_r), (nc, real_c) = flips(Zc, rr[0].cpu().numpy(), g["max0"]), flips(Zc.T, cc[0].cpu().numpy(), g["max1"])
        ties = {"rows_differing": nr, "cols_differing": nc, "not_a_4ulp_tie": real_r + real_c, "of": 2 * (M - 1),
                "against": "the reference's own run (tests/golden/roofline_4097.npz)"}
        assert real_r + real_c == 0, "config 5: a match index differs from the reference's beyond a 4-ulp tie"
        del Z, Zc
    nblk5, np5 = (M + 16) // 17, (M + 3) & ~3
    phys = (2.0 * nblk5 * np5 * 4 + nblk5 * 8.0 * M + 8.0 * M) * iters5 / (ms * 1e-3) / 1e9
    res.append({"kernel": "stream_resident_kernel, config 5 (4097^2, %d sweeps in one launch, K register-resident)" % iters5, "bound": "hbm", "achieved": phys,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": phys / HBM_PEAK_GBS, "ms": ms, "sweeps_per_s": iters5 / (ms * 1e-3),
                "streaming_model_GBps": gbs, "resident_model_GBps": 8.0 * M * M / (ms * 1e-3) / 1e9, "physical_GBps": phys,
                "argmax_vs_reference": ties,
                "note": "achieved / frac = the PHYSICAL traffic of the solve against 8 TB/s.  SURVEY 8d prices a sweep at 8*M*N bytes when it "
                        "streams (streaming_model_GBps: what a two-pass streaming solve would have to move at this sweep rate - more than HBM "
                        "can deliver) and the whole problem at 8*M*N when it is on-chip resident (resident_model_GBps); since round 5 the "
                        "solve IS resident: stream_resident_kernel (csrc/sinkhorn_stream.hip) keeps every workgroup's 17 x 4097 piece of K "
                        "in registers for all 200 sweeps, so a sweep moves no K at all - PHYSICAL traffic per sweep = 241 rows of column "
                        "partials written and read (2 x 3.95 MB), the 33 KB of {b_j, sweep} granules every workgroup polls, nothing else; "
                        "the memory system is a seventh busy.  What bounds a sweep now is two grid-wide hand-overs through memory that is not coherent across XCDs "
                        "(timeline of the diagnostic build, us per sweep: the barrier behind the partials 5.8 - write-through of the stores, "
                        "arrival, poll - the wait for the granules of the new b 5.8, row dots 1.7, reduce 0.9): 14.2 us = 70 400 sweeps/s "
                        "against 17.1 us = 58 700 for round 4's two launches a sweep (13.1 us of it the 67 MB read of K; hipGraph replay "
                        "of those 400 launches: 59 200 - the gaps are GPU-side, tools/config5_graph_probe.py).  Spins are bounded: a grid "
                        "that is not fully resident gives up and the problem is re-solved by the log-domain kernel"})
    return res


def gnn_secondary(ops, dev, pairs, rows_step, P_step, ms_per_step, outdoor):
    """SURVEY 8f rank 4 beside the headline, NOT in it: the AttentionalGNN stacks that sit between each level's gather and its
    cost build (first_layer.py:102, second_layer.py:89, third_layer.py:148), random weights, timed at the step's own problem
    counts - one AttentionalPropagation per level (both descriptor sides), scaled by the reference's layer counts (18 / 18 / 10).
    Third level: the fused kernel of csrc/gnn_fused.hip (BatchNorm as PATS.eval() leaves it: running statistics outdoors, batch
    statistics indoors, pats.py:112-118); fine level: the tile + attention kernels of csrc/gnn_fine.hip, run as a stack (round 5); coarse
    level: five packed-weights convolutions (csrc/conv_pk.hip) around the general attention kernel.  The MEASURED counterpart - whole
    steps with every head inside - is with_gnn_leg / `bench.py --with-gnn`."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242)

    def layer_ms(C, b, n, train, chunk):
        P = ops.PropagationParams(synth.gnn_params(seed=9, C=C))
        bb = min(b, chunk)
        x = torch.randn((bb, C, n), device=dev, generator=gen)
        s_ = torch.randn((bb, C, n), device=dev, generator=gen)
        ms = timed(lambda: ops.attentional_propagation(x, s_, P, bn_train=train, residual=x), reps=3, warm=1)
        del x, s_
        torch.cuda.empty_cache()
        return ms * b / float(bb), bb
    def fine_stack_ms(b, chunk, layers=4):
        """one layer of the fine level's stack as the stack runs it (round 5, csrc/gnn_fine.hip): both descriptor sets in one launch,
        descriptors kept in the kernel's own form between the layers - timed as a `layers`-deep stack, conversions included"""
        Ps = [ops.PropagationParams(synth.gnn_params(seed=9 + i, C=264)) for i in range(layers)]
        names = (["self", "cross"] * layers)[:layers]
        bb = min(b, chunk)
        x = torch.randn((bb, 264, 145), device=dev, generator=gen)
        s_ = torch.randn((bb, 264, 145), device=dev, generator=gen)
        o = (torch.empty_like(x), torch.empty_like(s_))
        ms = timed(lambda: ops.attentional_gnn(x, s_, Ps, names, out=o), reps=3, warm=1)
        del x, s_, o
        torch.cuda.empty_cache()
        return ms / layers / 2.0 * b / float(bb), bb          # per layer and descriptor set, like layer_ms
    t3, b3 = layer_ms(128, P_step, 65, not outdoor, 131072)
    t2, b2 = fine_stack_ms(rows_step, 4096)
    t1, b1 = layer_ms(448, pairs, 300, False, 64)
    per_step = {"coarse": 2 * 18 * t1, "fine": 2 * 18 * t2, "third": 2 * 10 * t3}
    total = sum(per_step.values())
    flops3 = 2.0 * 65 * (4 * 128 * 128 + 256 * 256 + 256 * 128) + 4 * 2 * (2.0 * 65 * 65 * 32)
    by3 = 3.0 * 128 * 65 * 4 + 128 * 65 * 4
    roof = {"kernel": "gnn_layer_fused_kernel (AttentionalPropagation at [128,65], %d problems per launch%s)"
                      % (b3, "" if outdoor else "; batch statistics: up to the hidden tensor, + statistics passes + last convolution"),
            "bound": "mfma", "achieved": 3.0 * flops3 * b3 / (t3 * b3 / P_step * 1e-3) / 1e12, "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "ms_per_launch": t3 * b3 / P_step, "algorithmic_tflops": flops3 * b3 / (t3 * b3 / P_step * 1e-3) / 1e12,
            "hbm_frac": by3 * b3 / (t3 * b3 / P_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "the whole layer in one kernel, activations LDS-resident in MFMA fragment order; priced like the cost build: fp32 "
                    "operands as fp16 hi + lo, three exact-product passes = 3 x the algorithmic flops against the dense fp16 matrix peak "
                    "(token padding 80 / 65 not counted); hbm_frac = x + source + residual in, out (4 x 33 KB per problem) against 8 TB/s"}
    roof["frac"] = roof["achieved"] / F16_PEAK_TFLOPS
    flops2 = 2.0 * 145 * (4 * 264 * 264 + 528 * 528 + 528 * 264) + 4 * 2 * (2.0 * 145 * 145 * 66)
    by2 = 4 * 153120.0 + 2 * 475680.0      # per problem and layer, all of it past the L2: x and attention images in, attention and output images
                                           # out (4 x 153 120 B), the block of projections (q, k, v^T as fragments: 475 680 B) written and read
    fine = {"kernel": "gnn_fine_tile_kernel + gnn_fine_attn_kernel (AttentionalPropagation at [264,145], two launches a layer, both descriptor sets = %d problems per launch)" % (2 * b2),
            "bound": "mfma", "achieved": 3.0 * flops2 * b2 / (t2 * b2 / rows_step * 1e-3) / 1e12, "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "ms_per_launch": 2.0 * t2 * b2 / rows_step, "ms_per_4096_problems": t2 * 4096.0 / rows_step,
            "algorithmic_tflops": flops2 * b2 / (t2 * b2 / rows_step * 1e-3) / 1e12,
            "hbm_frac": by2 * b2 / (t2 * b2 / rows_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "per-token products (mlp of layer l + q / k / v of layer l + 1) on 64-column tiles of the flattened (problem, token tile) "
                    "list, operands by LDS DMA, outputs in the accumulators, hidden tensor never off the CU; attention core per problem in wave "
                    "roles; same 3 x pricing as the third level's fused layer against the NOMINAL dense fp16 peak - the tile kernel clocks to "
                    "the power budget (1.5-2.0 GHz by box; the same instruction stream on all-zero operands runs 21 % faster: "
                    "profiles/r05_gnn_fine_power_zeros_ab.txt), matrix pipe 49-59 % busy at the clock it gets; hbm_frac = 1.56 MB per problem "
                    "and layer (four descriptor images + the projections written and read) against 8 TB/s; timed as a 4-layer stack, "
                    "conversions at its ends and the first layer's own projection launch included"}
    fine["frac"] = fine["achieved"] / F16_PEAK_TFLOPS
    # the matrix pipe's own rate on random operands (tools/mfma_rate_probe.hip, profiles/r05_mfma_rate_probe.txt): 1 720 TFLOP/s at the
    # 1.74 GHz the part sustains on toggling data - what a split-fp16 product can at most reach here
    fine["frac_of_measured_random_operand_ceiling_1720_TFLOPs"] = fine["achieved"] / 1720.0
    roof["fine_level_layer"] = fine
    return {"ms_per_step": per_step, "layers": {"coarse": 18, "fine": 18, "third": 10},
            "sample": {"third": "%d of %d problems" % (b3, P_step), "fine": "%d of %d rows" % (b2, rows_step), "coarse": "%d of %d pairs" % (b1, pairs)},
            "pairs_per_s_with_gnn": pairs / ((ms_per_step + total) * 1e-3),
            "note": "headline step + the three GNN stacks on random weights, added as sequential stream time (every kernel fills the "
                    "GPU on its own); backbones, KeypointEncoder, final_proj and scale heads not included"}, roof


def gather_layout_ab(ops, dev, cap, P_step, rows=2048):
    """The two descriptor gathers on the SAME logical maps in both memory orders (a sample of `rows` fine rows and the
    matching share of third-level points, times scaled to the step's launch sizes): outputs compared bit for bit."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    R = min(rows, cap.rows_cap)
    P = max(64, int(P_step * R / float(cap.rows_cap)))

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    maps = [torch.randn(sh, device=dev, generator=gen) for sh in ((2 * R, 64, 48, 48), (2 * R, 64, 24, 24), (2 * R, 128, 12, 12))]
    title, rub = torch.randn((R, 8), device=dev, generator=gen), torch.randn((R, 264), device=dev, generator=gen)
    out = torch.empty((2, R, 264, 145), dtype=torch.float32, device=dev)
    f_nchw = timed(lambda: ops.fine_descriptors(maps, title, rub, out=out))
    ref = out.clone()
    maps = [cl(m) for m in maps]
    f_nhwc = timed(lambda: ops.fine_descriptors(maps, title, rub, out=out))
    same = torch.equal(ref, out)
    del maps, out, ref
    ff0, ff1 = (torch.randn((R, 128, 52, 52), device=dev, generator=gen) for _ in range(2))
    mk0 = (torch.randint(1, 11, (P, 2), device=dev, generator=gen) * 8 + 4).float()
    mk1 = torch.rand((P, 2), device=dev, generator=gen) * 96
    b_ids = torch.sort(torch.randint(0, R, (P,), device=dev, generator=gen))[0]
    kenc, rub3 = torch.randn((128, 64), device=dev, generator=gen), torch.randn((R, 128, 144), device=dev, generator=gen)
    o = (torch.empty((P, 128, 65), device=dev), torch.empty((P, 128, 65), device=dev))
    t_nchw = timed(lambda: ops.third_descriptors(ff0, ff1, mk0, mk1, b_ids, kenc, rub3, out=o))
    r0, r1 = o[0].clone(), o[1].clone()
    ff0, ff1 = cl(ff0), cl(ff1)
    t_nhwc = timed(lambda: ops.third_descriptors(ff0, ff1, mk0, mk1, b_ids, kenc, rub3, out=o))
    same = same and torch.equal(r0, o[0]) and torch.equal(r1, o[1])
    assert same, "the channels-last gathers differ from the NCHW gathers"
    kf, kt = cap.rows_cap / float(R), P_step / float(P)
    return {"sample": "%d fine rows, %d third-level points; ms scaled to %d rows / %d points" % (R, P, cap.rows_cap, P_step),
            "fine_desc_ms": {"nchw": f_nchw * kf, "channels_last": f_nhwc * kf},
            "third_desc_ms": {"nchw": t_nchw * kt, "channels_last": t_nhwc * kt}, "outputs_bit_identical": bool(same)}


def step_determinism(batch, nets, cap, wl, n=4):
    """The bench's steps all run on the same resident inputs: n more of them, every stage's output compared bit for bit with
    the first one's (the fine-level log-plans of the rows in use, the third-level points, the matches).  Before the round-3
    barrier fix (now wg_barrier() in csrc/common.hpp) the fine level differed in ~10 of 20 224 problems in every step."""
    kw = dict(if_outdoor=wl["outdoor"], merge_new=wl["merge_new"], iters=ITERS)
    ref, rep = None, {"steps": n, "fine_log_plan_problems_differing": [], "third_level_points_differing": [],
                      "matches_differing": [], "match_count_equal": True}
    for k in range(n):
        out = batch.forward_pairs(nets.lefts, nets.rights, nets, cap, **kw)
        M = int(out["M"].item())
        live = int(out["rows"].chunk_base[-1].item())         # rows in use: padding rows past it are skipped by the launches
        cur = {"Z2": out["stages"]["Z2"][:live].clone(), "m1f": out["stages"]["m1f"].clone(), "ml": out["matches_l"][:M].clone(),
               "mr": out["matches_r"][:M].clone(), "M": M, "P": int(out["P"].item())}
        if ref is None:
            ref = cur
            continue
        rep["fine_log_plan_problems_differing"].append(int((cur["Z2"] != ref["Z2"]).flatten(1).any(1).sum().item()))
        P = min(cur["P"], ref["P"])
        rep["third_level_points_differing"].append(int((cur["m1f"][:P] != ref["m1f"][:P]).flatten(1).any(1).sum().item()))
        same = cur["M"] == ref["M"]
        rep["match_count_equal"] = rep["match_count_equal"] and same
        rep["matches_differing"].append(int(((cur["ml"] != ref["ml"]) | (cur["mr"] != ref["mr"])).any(1).sum().item()) if same else -1)
        del cur
    rep["identical"] = rep["match_count_equal"] and not any(rep["fine_log_plan_problems_differing"] + rep["third_level_points_differing"]
                                                              + rep["matches_differing"])
    return rep


def guard_trip_sweep(ops, batch, nets, cap, wl, fracs=(0.01, 0.10)):
    """pairs/s when a fraction of the fine / third-level problems leaves the linear-domain solver's guard band and is
    re-solved in the log domain: the rows' backbone maps are scaled by 32 (both sides: scores x 1024, far outside the band),
    three steps are timed, the maps restored (a power of two: exactly)."""
    res = []
    R = cap.rows_cap
    g = torch.Generator(device=nets.m0.device)
    g.manual_seed(12345)
    kw = dict(if_outdoor=wl["outdoor"], merge_new=wl["merge_new"], iters=ITERS)
    for frac in fracs:
        pick = torch.nonzero(torch.rand((R,), device=nets.m0.device, generator=g) < frac).flatten()
        both = torch.cat([pick, pick + R])
        for t in (nets.m0, nets.m1, nets.m2):
            t[both] *= 32.0
        nets.ff0[pick] *= 32.0
        nets.ff1[pick] *= 32.0
        torch.cuda.synchronize()
        ops.sinkhorn_fallbacks(reset=True)
        t0 = time.perf_counter()
        for _ in range(3):
            batch.forward_pairs(nets.lefts, nets.rights, nets, cap, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        trips = ops.sinkhorn_fallbacks(reset=True)
        for t in (nets.m0, nets.m1, nets.m2):
            t[both] /= 32.0
        nets.ff0[pick] /= 32.0
        nets.ff1[pick] /= 32.0
        res.append({"wild_row_fraction": frac, "pairs_per_s": 3 * cap.pairs / dt, "guard_fallbacks_per_step": trips / 3.0,
                    "note": "no stream overlap in this leg"})
    return res


def secondary_workloads(ops, batch, dev, rank, names=("scannet", "yfcc"), steps=5, warm=2, maps="nchw"):
    """BASELINE.json configs[2] and configs[3] in the SAME run as the headline (round-5 verdict item 4): the same step on the
    ScanNet shapes (indoor: one fine chunk of up to 300 rows, +ln3, fixed-cell label, merge_old) and on the YFCC shapes (24x32 grid,
    769x769 coarse problem, 16 pairs a step - the 8-GPU sharding of configs[3] is rank-local work of exactly this kind).  Per
    workload: pairs/s over `steps` steps, the step's kernels timed inside the steps by HIP events with the dominant one's
    fraction of the HBM roofline (algorithmic bytes / time / 8 TB/s), and pair 0 of a step checked against the CPU oracle
    stage by stage (index outputs asserted)."""
    out = []
    for name in names:
        h, w, if_local, outdoor, pairs, label = WORKLOADS[name]
        wl = {"outdoor": outdoor, "merge_new": outdoor, "bias_k": 2.0 if outdoor else 3.0}
        gen = torch.Generator(device=dev)
        gen.manual_seed(synth.SEED + rank)
        cap = batch.Capacities(pairs, h, w, if_local=if_local)
        t0 = time.perf_counter()
        nets = BenchNets(ops, dev, gen, cap, h, w, batch=batch, channels_last=maps == "nhwc")
        torch.cuda.synchronize()
        setup_s = time.perf_counter() - t0
        run_steps(batch, nets, cap, wl, None, warm, None)
        ev = {}
        nets.ev = ev
        watch = StepWatch(cap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = run_steps(batch, nets, cap, wl, ev, steps, None, watch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nets.ev = None
        P_step, rows_step = int(o["P"].item()), int(o["rows"].chunk_base[-1].item())
        ms = lambda tag: float(np.mean([a.elapsed_time(b_) for a, b_ in ev[tag]]))       # noqa: E731
        kernels = [("third_fused3_kernel (third-level cost + OT + Compute_result, %d problems)" % P_step, ms("third"), THIRD_BYTES_PER_PROBLEM * P_step),
                   ("cost_mfma_kernel + sinkhorn_blk145w2_kernel (fine-level launch pair, %d rows)" % rows_step, ms("fine"), FINE_BYTES_PER_ROW * rows_step),
                   ("fine_desc_kernel (a15, %d stacked crops)" % (2 * rows_step), ms("fine_desc"), FD_BYTES_PER_IMAGE * 2.0 * rows_step),
                   ("third_desc_kernel (a16, %d points)" % P_step, ms("third_desc"), TD_BYTES_PER_POINT * float(P_step))]
        roofs = sorted(({"kernel": k, "avg_launch_ms": t, "algorithmic_bytes_per_launch": float(by), "bound": "hbm",
                         "achieved": by / (t * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
                        for k, t, by in kernels), key=lambda r: -r["avg_launch_ms"])
        rep = {"workload": label, "value": pairs * steps / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warm,
               "pairs_per_step": pairs, "rows_in_use_per_step": rows_step, "third_problems_per_step": P_step, "setup_s": setup_s,
               "roofline": roofs[0], "other_kernels": roofs[1:], "map_layout": maps}
        try:
            o2 = batch.forward_pairs(nets.lefts, nets.rights, nets, cap, if_outdoor=wl["outdoor"], merge_new=wl["merge_new"], iters=ITERS)
            cb = cpu_baseline(ops, batch, dev, nets, cap, wl, o2, torch_leg=False)
            rep["parity_sample"] = cb["parity_sample"]
            rep["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample")}
            del o2
        except AssertionError as e:
            rep["parity_sample"] = {"FAILED": repr(e)[:400]}
        out.append(rep)
        del nets, o, ev, watch
        torch.cuda.empty_cache()
    return out
