#!/usr/bin/env python3
"""bench.py - image-pairs/sec of the PATS OT hot path on MI355X (BASELINE.json configs[1]).

One STEP = one pass of the hot path over a batch of `--pairs` synthetic 640x480 pairs at the
reference's shapes, inputs resident in HBM before the timed region:
  L1  [1,448,300]^2 cost build (MFMA) -> log_optimal_transport 301x301, 100 sweeps -> column mass
      -> argmax + 15-step area expansion -> split_patches (host, one D->H copy, cap 2w = 40 as
      `if_local`) -> per chunk Compute_imgs (bounds, left crops, right crop+bilinear resize)
  L2  per chunk [B,264,145]^2 cost -> log_optimal_transport2 145x145, 100 sweeps, +ln2 dustbin
      -> argmax + 8-step expansion
  L3  per chunk [60*B,128,65]^2 cost -> log_optimal_transport2 65x65, 100 sweeps -> Compute_result
Descriptors are synthetic (no weights/datasets exist for the reference here); P = 60*B is the
SURVEY's chosen fill.  `value` = pairs/sec over all ranks (pairs shard across ranks, no data-path
collective; "weak" scaling).  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from pats_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ITERS = 100
ONE = [None]            # device-resident 1.0 (the reference's `self.one`, second_layer.py:63)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=16, help="image pairs per step per rank")
    ap.add_argument("--fill", type=int, default=60, help="third-level problems per fine problem (P = fill*B)")
    ap.add_argument("--per-chunk", action="store_true",
                    help="run the fine/third stages once per coarse chunk like the reference's loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    (and across the coarse chunks, unless --per-chunk): the reference walks pairs and chunks in
    Python loops (evaluate.py:25, pats.py:33) only because it targets one 16-40 GB card; every
    coarse / fine / third-level problem is independent, so with 288 GB each stage is one launch."""

    def __init__(self, ops, dev, gen, pairs, fill, per_chunk=False):
        c = synth.coarse_inputs()
        self.pairs, self.h, self.w = pairs, c["h"], c["w"]
        self.per_chunk_imgs = per_chunk
        rep = lambda a: torch.from_numpy(a).to(dev).repeat(pairs, *([1] * (a.ndim - 1))).contiguous()  # noqa: E731
        self.d0, self.d1, self.ns = rep(c["d0"]), rep(c["d1"]), rep(c["ns"])
        self.alpha = torch.tensor(float(c["alpha"]), device=dev)
        left, right = synth.image_pair()
        self.left, self.right = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
        self.lefts = self.left.expand(pairs, -1, -1, -1).contiguous()       # every pair: the same synthetic image
        self.rights = self.right.expand(pairs, -1, -1, -1).contiguous()
        # dry run of the coarse stage to learn the (deterministic) chunk plan
        self.plan = coarse_stage(ops, self)[0]
        B1 = sum(self.plan)
        groups = [b * pairs for b in self.plan] if per_chunk else [B1 * pairs]
        self.chunks = []
        for B in groups:
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
            self.chunks.append(dict(B=B, P=P, f0=f0, f1=f1, sx=sx, sy=sy, ns2=(sx * sy).contiguous(), t0=t0,
                                    t1=t1, sc=sc, p_s=p_s, p_t=p_t))
        self.B = B1
        self.P = fill * B1


def coarse_stage(ops, wl):
    """first_layer.py:110-146 for all pairs: one batched cost+OT launch, one batched expansion; then
    per pair the chunk plan (host) and per chunk the subdivision gather.  Returns per-pair plans."""
    Z = ops.cost_ot(wl.d0, wl.d1, 1, wl.alpha, wl.ns, ITERS)
    scales = ops.colmass_sqrt(Z)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_first(Z, scales, (480, 640), 32)
    sum_cycle = torch.cumsum(torch.logical_not(ifn1).int(), dim=1)
    sc_host = sum_cycle.to("cpu").numpy()          # the step's ONE host read: chunk plans and crop counts of all pairs
    plans, seconds = [], []
    for i in range(wl.pairs):
        n, second, third = ops.split_patches(sc_host[i], wl.h, wl.w, 2 * wl.w)
        seconds.append(second)
    if wl.per_chunk_imgs:
        # the reference's loop (first_layer.py:136-146): one Compute_imgs per pair and chunk mask
        for i in range(wl.pairs):
            plan = []
            for lo, hi in seconds[i]:
                mask = torch.logical_or(ifn1[i:i + 1], torch.logical_or(sum_cycle[i:i + 1] <= lo,
                                                                        sum_cycle[i:i + 1] > hi))
                nl, nr, xsn, ysn, avn = ops.Compute_imgs(xs[i:i + 1], ys[i:i + 1], pts[i:i + 1], mask, wl.left,
                                                         wl.right, width=wl.w, height=wl.h)
                plan.append(int(nr.shape[0]))
            plans.append(plan)
        return plans
    # One gather for the whole step: Compute_imgs takes a batch of images (crops ordered image, patch),
    # and chunk c of pair i is rows [off_i + lo, off_i + min(hi, K_i)) of it - the cumsum is monotone, so
    # a chunk mask selects a contiguous run of matched patches
    # (tests/test_gpu_parity.py::test_chunk_crops_are_slices, ::test_compute_imgs_batch_of_images).
    counts = [int(sc_host[i, -1]) for i in range(wl.pairs)]
    nl, nr, xsn, ysn, avn = ops.Compute_imgs(xs, ys, pts, ifn1, wl.lefts, wl.rights, width=wl.w, height=wl.h,
                                             known_count=counts)
    off = 0
    for i in range(wl.pairs):
        K = counts[i]
        views = [(nl[off + lo:off + min(hi, K)], nr[off + lo:off + min(hi, K)]) for lo, hi in seconds[i]]
        plans.append([v[1].shape[0] for v in views])
        off += K
    return plans


def fine_and_third(ops, ch, ev):
    Z2 = ops.cost_ot(ch["f0"], ch["f1"], 2, ONE[0], ch["ns2"], ITERS, bias_k=2.0)
    out2 = ops.est_position_second(Z2, ch["sx"], ch["sy"], [96, 96], 8)
    # third level: cost build + OT + exp + Compute_result + label in ONE launch (the dominant kernel)
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    res = ops.third_level(ch["t0"], ch["t1"], ch["sc"], ch["p_s"], ch["p_t"], outdoor=True, iters=ITERS)
    if ev is not None:
        e1.record()
        ev.append((e0, e1, ch["P"]))
    return out2, res


def step(ops, wl, ev):
    coarse_stage(ops, wl)
    for ch in wl.chunks:
        fine_and_third(ops, ch, ev)


def cpu_baseline(pairs_B, pairs_P, seconds):
    """The CPU oracle ("port") on the host cores: L1 in full, bounded samples of L2/L3 scaled to one
    pair.  Checker code timed as a baseline only - never part of the measured GPU path."""
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
    oracle.argmax(Z2)
    oracle.iterative_expand(np.exp(Z2), f["scale_x"], f["scale_y"], 12, 12, 12, 1e-3, 8)
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
    oracle.compute_result(np.exp(Z3), sq, sq, t3in["p_s"], t3in["p_t"], True)
    t_l3 = (time.perf_counter() - t0) / np3
    per_pair = t_l1 + t_l2 * pairs_B + t_l3 * pairs_P
    return {"value": 1.0 / per_pair, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "oracle/pats_oracle.c (OpenMP over problems): L1 301x301 in full (%.3fs), %d L2 "
                      "problems (%.4fs each), %d L3 problems (%.5fs each), scaled to B=%d, P=%d per pair"
                      % (t_l1, nb, t_l2, np3, t_l3, pairs_B, pairs_P)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    from pats_amd import ops

    gen = torch.Generator(device=dev)
    gen.manual_seed(synth.SEED + rank)
    ONE[0] = torch.tensor(1.0, device=dev)
    wl = Workload(ops, dev, gen, args.pairs, args.fill, args.per_chunk)
    B, P = wl.B, wl.P

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(ops, wl, None)
    ev = []
    barrier()
    ops.sinkhorn_fallbacks(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(ops, wl, ev)
    barrier()
    dt = time.perf_counter() - t0
    fallbacks = ops.sinkhorn_fallbacks(reset=True)       # after the timed region (it synchronises)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_pairs = args.pairs * args.steps * world
    value = total_pairs / dt

    # dominant kernel: the 65x65 third-level Sinkhorn launch (HIP events on the launch stream)
    ms = np.array([a.elapsed_time(b) for a, b, _ in ev])
    probs = np.array([p for _, _, p in ev], dtype=np.float64)
    # algorithmic HBM bytes per problem of the fused third-level kernel: both descriptor blocks in
    # (2 x 128 x 65 fp32), areas + coarse points in, 16 matches + labels + flags out; the 65x65 plan
    # stays on chip (SURVEY 8d "cost build: 4*D*(M+N) in, 0 out if fused")
    BYTES_PER_PROBLEM = 2 * 128 * 65 * 4 + 64 * 4 + 2 * 2 * 8 + 2 * 16 * 2 * 4 + 16 * 2 * 4 + 16
    alg_bytes = float(BYTES_PER_PROBLEM) * probs
    achieved = float((alg_bytes / (ms * 1e-3)).mean() / 1e9)
    exp_rate = float((2.0 * ITERS * 65 * 65 * probs / (ms * 1e-3)).mean())
    sweeps_per_pair = ITERS * (1 + B + P)        # one sweep = row + column normalisation of one problem

    # HBM traffic of the dominant kernel from rocprofv3 PMC passes (collected separately, see the file)
    traffic, traffic_src = None, None
    pmc_path = os.path.join(REPO, "profiles", "r01_pmc_third.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        traffic = float(pmc["hbm_bytes_per_problem"]) * float(probs.mean())
        traffic_src = "profiles/r01_pmc_third.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), " \
                      "FETCH calibrated x%.2f on cost65_kernel's known byte count; per problem x problems per launch" \
                      % pmc["fetch_calibration"]["factor"]

    out = {
        "metric": "image-pairs/sec (coarse+fine OT) on 640x480 MegaDepth; OT iters/sec per pair",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: MegaDepth 640x480 shapes, outdoor coarse+fine+third OT + cost volume "
                               "+ expansion + subdivision gather",
                   "pairs_per_step_per_rank": args.pairs, "batching": "each stage is one launch over all pairs of the step", "L1": "1x[448,300]^2 -> 301x301",
                   "L2": "%d x [264,145]^2 -> 145x145 (%d coarse chunks, %s)"
                         % (B, len(wl.plan), "one launch per chunk" if args.per_chunk else "batched into one launch"),
                   "L3": "%d x [128,65]^2 -> 65x65 (fill %d)" % (P, args.fill), "sinkhorn_iters": ITERS,
                   "parallelism": "pairs sharded over %d rank(s), no data-path collective" % world},
        "ot_iters_per_sec": value * sweeps_per_pair,
        "guard_fallbacks_per_step": fallbacks / max(1, args.steps),
        "roofline": {"bound": "hbm", "kernel": "third_fused_kernel (fused third level, %d problems per launch)" % wl.chunks[0]["P"],
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": float(alg_bytes.mean()),
                     "avg_launch_ms": float(ms.mean()), "launches": int(len(ms)),
                     "algorithmic_bytes_per_problem": BYTES_PER_PROBLEM,
                     "sweep_elements_per_s": exp_rate,
                     "mfma_flops_per_s": float((2.0 * 128 * 64 * 64 * probs / (ms * 1e-3)).mean()),
                     "note": "fused cost build (fp32 MFMA) + 100 linear-domain Sinkhorn sweeps + Compute_result per 65x65 problem, "
                             "one wave each, the 65x65 block held in registers; descriptors are read once, the plan never "
                             "reaches HBM.  HBM is the nearest of the two allowed rooflines but not the limiter: SQ counters "
                             "(profiles/r01_pmc_third.json) show the sweeps VALU-issue bound at 92 % SIMD issue utilisation"},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B, P, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
