#!/usr/bin/env python3
"""bench.py - image-pairs/sec of the PATS OT hot path on MI355X (BASELINE.json configs[1]).

One STEP = one pass of the hot path over a batch of `--pairs` synthetic 640x480 pairs at the reference's shapes
(SURVEY.md 8d config 2), through pats_amd.batch (every stage ONE launch over all pairs and chunks, NO host read):
  L1    [1,448,300]^2 cost build (MFMA) -> log_optimal_transport 301x301, 100 sweeps -> column mass -> 15-step area
        expansion -> cumulative match counts, split_patches (cap 2w = 40 as `if_local`), chunk masks, the fine level's
        row table -> Compute_imgs for all pairs (bounds, left crops, right crop + bilinear resize = the native tensor_resize)
  L2    descriptor sampling (second_layer.py:71-86) from synthetic backbone maps -> [B,264,145]^2 cost ->
        log_optimal_transport2 145x145, 100 sweeps, +ln2 dustbin -> 8-step expansion -> merge_patches_new for every chunk
        of every pair in the reference's order (scores_back hand-over) -> pats.py:38-39 tail rows
  L3    surviving cells -> points (pats.py:53-58) -> 8x8 window gather (third_layer.py:121-146) from synthetic maps ->
        [P,128,65]^2 cost -> log_optimal_transport2 65x65, 100 sweeps -> Compute_result + label, P decided by the MERGE
  out   refine scatter (pats.py:59-67) + get_result for all chunks (utils.py:189-213): matches_l / matches_r
What is synthetic: the networks' outputs (ResNet / FPN maps, GNN + final_proj as the identity, scale heads) - there are no
weights or datasets here; every pair of a step has its own descriptors, maps and images.  The number of third-level
problems is whatever the merge leaves (every 8-px cell belongs to at most one window per chunk: <= 16 h w per pair), not
a chosen fill - rounds 1-2 fixed P = 60 B with a stand-in keep mask and no merge.
`value` = pairs/sec over all ranks (pairs shard across ranks, no data-path collective; "weak" scaling); after the timed
region every rank's matches of its last step are gathered to rank 0 over RCCL (shard.gather_matches), timed separately.

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

from pats_amd import synth  # noqa: E402,F401
from benchlib.common import *  # noqa: E402,F401,F403  (peaks, ITERS, DTYPE, WORKLOADS, algorithmic bytes per unit)
from benchlib.nets import BenchNets, GnnNets, StepWatch, run_steps, with_gnn_leg  # noqa: E402,F401
from benchlib.baseline import cpu_baseline  # noqa: E402
from benchlib.secondary import (gather_layout_ab, gnn_secondary, guard_trip_sweep, secondary_rooflines, secondary_workloads,  # noqa: E402
                                step_determinism)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=None,
                    help="image pairs per step per rank (default 48 at 640x480: about 130 GB of synthetic backbone maps "
                         "resident in the 288 GB of HBM; 16 at YFCC size)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="megadepth",
                    help="megadepth = BASELINE configs[1] (the bench line); scannet = configs[2] shapes (indoor rules, one L2 chunk); "
                         "yfcc = configs[3] shapes (768x1024 pairs, 769x769 coarse problem) - secondary measurements")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="strong-scaling mode (configs[3]: 4000 YFCC pairs): this many pairs in all, split over the ranks by "
                         "shard.my_pairs; every rank walks its share in steps of --pairs (the last one partly filled)")
    ap.add_argument("--maps", choices=["nhwc", "nchw"], default="nchw",
                    help="memory order of the synthetic backbone maps the two descriptor gathers read in the HEADLINE steps: "
                         "nchw (default) = NCHW-contiguous, what the unchanged reference's backbones emit (second_layer.py:66-69, "
                         "third_layer.py:113-117); nhwc = torch.channels_last, what they emit after ops.prepare_backbones(model).  Same "
                         "logical tensors, same outputs bit for bit.  The other layout is timed in the same run as a secondary "
                         "(value_nchw / value_nhwc) unless --no-secondary")
    ap.add_argument("--rows-cap", choices=["worst", "dry-run"], default="worst",
                    help="row capacity of the fine level's table: worst = pairs * (N + (Cmax - 1) w), what a caller that knows nothing about "
                         "its data allocates (default); dry-run = a dry run of the coarse stage on the step's own pairs + 1 %% (round 3)")
    ap.add_argument("--soak", type=int, default=0, metavar="N",
                    help="no timing: run N + 1 whole steps on the same inputs and compare every stage's output with the first step's "
                         "bit for bit (prints the step_determinism object and exits)")
    ap.add_argument("--wild", type=float, default=0.0, metavar="FRAC",
                    help="diagnostic: scale the backbone maps of this fraction of the fine rows by 32 (scores x 1024) before the timed steps - "
                         "the guard-trip leg as the whole run, for kernel traces; not a bench line")
    ap.add_argument("--with-gnn", action="store_true",
                    help="time whole steps WITH the layers' heads inside (GnnNets: KeypointEncoder, the 18 / 18 / 10-layer GNN stacks, "
                         "final_proj, scale heads on random weights) and print that report instead of the headline line")
    ap.add_argument("--plan-only", action="store_true",
                    help="no GPU: print what a --gpus N run would do - every rank's pair share, steps, row / problem capacities, resident "
                         "bytes and expected set-up time - as one JSON line and exit (the first real 8-GPU run cannot fail on plumbing)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not re-run three steps under rocprofv3 --pmc for roofline.traffic (default: done when rocprofv3 is on PATH, one "
                         "rank, not --no-secondary; the committed profiles/r*_pmc_step_<layout>.json is the fallback, with its age in the line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-pair-at-a-time latency legs (tools/latency.py)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary roofline / guard-trip measurements")
    ap.add_argument("--overlap", type=int, default=0, metavar="K",
                    help="K > 0: two HIP streams with disjoint compute-unit masks - the HBM-bound stages (crops, descriptor gathers) of "
                         "one batch on K of every 8 CUs of each shader engine, the VALU-bound solver stages of its neighbour on the other "
                         "8 - K (ops.masked_stream).  0 = one stream, stage after stage.  (Plain, unmasked streams were measured SLOWER than "
                         "one stream: 30.3 against 26.7 ms per 48-pair step - every kernel of the path fills all CUs' registers on its own, "
                         "so they only time-slice)")
    ap.add_argument("--no-overlap", action="store_true", help="(default since round 3; kept so that older command lines still parse)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def live_pmc(args):
    """roofline.traffic measured in THIS run (round-4 verdict item 8): bench.py re-invokes itself for three steps under
    `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes, no trace domains beside them - the
    micro-architecture guide's recipe) and tools/pmc_step.py turns the two counter dumps into bytes per launch.  None if
    rocprofv3 is missing, a pass fails or times out - the caller then falls back to the committed file and says how old it is."""
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import pmc_step
    tmp = tempfile.mkdtemp(prefix="pats_pmc_", dir="/tmp")
    cmd = [sys.executable, os.path.abspath(__file__), "--maps", args.maps, "--steps", "3", "--warmup", "1", "--no-secondary",
           "--no-cpu-baseline", "--no-pmc", "--no-latency"] + (["--pairs", str(args.pairs)] if args.pairs else [])
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        line = os.path.join(tmp, "bench.json")
        for counter, sub in (("FETCH_SIZE", "F"), ("WRITE_SIZE", "W")):
            with open(line if sub == "F" else os.devnull, "w") as fo:
                r = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, sub), "--"] + cmd, cwd="/tmp",
                                   env=env, stdout=fo, stderr=subprocess.DEVNULL, timeout=300)
            if r.returncode != 0:
                return None
        out = pmc_step.summarise(os.path.join(tmp, "F"), os.path.join(tmp, "W"), line)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(REPO, "gpurun_out", "pmc_step_%s_live.json" % args.maps), "w"), indent=1)
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def plan_only(args):
    """What `bench.py --gpus N ...` would do, computed on the host alone (the C-ABI library loads without a GPU: the chunk
    planner is host code).  Resident bytes: BenchNets' tensors by their shapes; set-up time from the rate rank 0 measured
    (148 GB of synthetic maps in 2.5 s)."""
    from pats_amd import batch, shard
    h, w, if_local, outdoor, default_pairs, label = WORKLOADS[args.workload]
    pairs = args.pairs if args.pairs else default_pairs
    cap = batch.Capacities(pairs, h, w, if_local=if_local)
    N, R, Pc = h * w, cap.rows_cap, cap.P_cap
    f = 4
    resident = (2 * pairs * 448 * N + pairs * N) * f + 2 * pairs * 32 * h * 32 * w * 3 * f           # coarse descriptors, ns, images
    resident += 2 * R * (64 * 48 * 48 + 64 * 24 * 24 + 128 * 12 * 12) * f + R * (8 + 264 + 3 * 144) * f   # fine maps, title, rubbish, scales
    resident += 2 * 2 * R * 264 * 145 * f                                                              # a15 outputs, double-buffered
    resident += 2 * R * 128 * 52 * 52 * f + R * 128 * 144 * f + Pc * 64 * f + 4 * Pc * 128 * 65 * f    # third-level maps, rubbish, scale, a16 outputs
    world = max(1, args.gpus)
    ranks = []
    for r in range(world):
        if args.total_pairs > 0:
            mine = len(shard.my_pairs(args.total_pairs, r, world))
            steps = shard.steps_for(args.total_pairs, r, world, pairs)
        else:
            mine, steps = pairs * args.steps, args.steps
        ranks.append({"rank": r, "device": "cuda:%d" % r, "pairs": mine, "steps": steps,
                      "slots_idle_in_last_step": (steps * pairs - mine) if args.total_pairs > 0 else 0})
    plan = {"plan_only": True, "workload": label, "gpus": world, "scaling": "strong" if args.total_pairs > 0 else "weak",
            "pairs_per_step_per_rank": pairs, "grid": [h, w], "coarse_problem": "%d x %d" % (N + 1, N + 1),
            "rows_cap": R, "chunks_max": cap.Cmax, "third_problem_cap": Pc,
            "resident_synthetic_GB_per_rank": resident / 1e9, "expected_setup_s_per_rank": resident / 59.2e9,
            "total_pairs": args.total_pairs if args.total_pairs > 0 else pairs * args.steps * world,
            "launch": "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py %s"
                      % (world, " ".join(a for a in sys.argv[1:] if a != "--plan-only")),
            "collectives": ["barrier x 2 around the timed region", "all_reduce(MAX) of the elapsed time", "all_gather of (ms_per_step, setup_s)",
                            "shard.gather_matches after the clock: all_gather of the (pair, K) table + flat [K,4] payload to rank 0"],
            "ranks": ranks}
    assert sum(r_["pairs"] for r_ in ranks) == plan["total_pairs"]
    print(json.dumps(plan))


def main():
    args = parse()
    if args.plan_only:
        return plan_only(args)
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
    backend = os.environ.get("PATS_BENCH_BACKEND", "nccl")
    if os.environ.get("PATS_BENCH_SHARE_DEVICE"):
        local_rank %= torch.cuda.device_count()
    elif torch.cuda.device_count() < max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but only %d device(s) are visible (one rank per GPU over RCCL)"
                         % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # PATS_BENCH_FORCE_DIST=1 (a test knob): initialise the process group under a launcher even with ONE rank, so that a 1-GPU box
    # runs every collective of the multi-rank path (barrier, all_reduce, all_gather, shard.gather_matches) over RCCL itself
    if world > 1 or (os.environ.get("PATS_BENCH_FORCE_DIST") and "WORLD_SIZE" in os.environ):
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    from pats_amd import batch, ops, shard

    # roofline.traffic of THIS run: two rocprofv3 --pmc passes over three steps of this same script, in child processes, at the END
    # of this run, once its resident set is freed (two of them do not fit 288 GB side by side; until round 5 the passes ran first -
    # and the CU-masked two-stream leg of the parent then ran at a third of its rate, every time: 548 against 1 930 pairs/s
    # without the passes); until then the committed file stands in, and it stays if a pass fails - with its age
    pj_live = None
    want_live_pmc = (world == 1 and not args.no_pmc and not args.no_secondary and not args.with_gnn and args.soak == 0
                     and args.workload == "megadepth" and args.total_pairs == 0 and args.wild == 0.0)

    h, w, if_local, outdoor, default_pairs, label = WORKLOADS[args.workload]
    pairs = args.pairs if args.pairs else default_pairs
    wl = {"outdoor": outdoor, "merge_new": outdoor, "bias_k": 2.0 if outdoor else 3.0}
    gen = torch.Generator(device=dev)
    gen.manual_seed(synth.SEED + rank)
    cap = batch.Capacities(pairs, h, w, if_local=if_local)
    t_setup = time.perf_counter()
    nets = BenchNets(ops, dev, gen, cap, h, w, batch=batch, channels_last=args.maps == "nhwc", rows_cap_policy=args.rows_cap)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
    n_gpus = dist.get_world_size() if dist is not None else 1
    if args.with_gnn:
        print(json.dumps(with_gnn_leg(ops, batch, dev, nets, cap, wl, h, w, max(1, args.steps), warm=max(1, args.warmup))))
        return
    if args.soak > 0:
        rep = step_determinism(batch, nets, cap, wl, n=args.soak + 1)
        rep["all_zero"] = rep.pop("identical")
        for k in ("fine_log_plan_problems_differing", "third_level_points_differing", "matches_differing"):
            rep[k] = {"steps_compared": len(rep[k]), "steps_with_a_difference": int(sum(1 for v in rep[k] if v != 0)), "worst": int(max(rep[k]))}
        print(json.dumps(rep))
        return

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # strong-scaling mode: --total-pairs split over the ranks; a rank with k pairs runs ceil(k / pairs) steps (each step
    # walks `pairs` slots; the slots past its share in the last step are real work on synthetic pairs and are not counted)
    steps = args.steps
    if args.total_pairs > 0:
        steps = shard.steps_for(args.total_pairs, rank, n_gpus, pairs)

    streams = None
    if args.overlap > 0:
        # mask bit c <-> shader engine c % 32, CU c / 32 of that engine (measured, tools/cu_mask_probe*.py: a VALU-bound kernel
        # slows down by the engine with the fewest enabled CUs; a mask that empties an engine is ignored by the runtime)
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        k = min(max(args.overlap, 1), 7)
        streams = (ops.masked_stream([c for c in range(n_cu) if c // 32 < k]), ops.masked_stream([c for c in range(n_cu) if c // 32 >= k]))
    if args.wild > 0.0:
        gw = torch.Generator(device=dev)
        gw.manual_seed(12345)
        pick = torch.nonzero(torch.rand((cap.rows_cap,), device=dev, generator=gw) < args.wild).flatten()
        both = torch.cat([pick, pick + cap.rows_cap])
        for t_ in (nets.m0, nets.m1, nets.m2):
            t_[both] *= 32.0
        nets.ff0[pick] *= 32.0
        nets.ff1[pick] *= 32.0
    run_steps(batch, nets, cap, wl, None, args.warmup, streams)
    ev = {}
    nets.ev = ev
    watch = StepWatch(cap) if streams is None else None
    barrier()
    ops.sinkhorn_fallbacks(reset=True)
    ops.profile_marker(1)                                # kernel traces are cut to the steps between the two markers
    t0 = time.perf_counter()
    out = run_steps(batch, nets, cap, wl, ev, steps, streams, watch)
    barrier()
    dt = time.perf_counter() - t0
    ops.profile_marker(2)
    nets.ev = None
    rank_ms_per_step = 1e3 * dt / max(steps, 1)
    fallbacks = ops.sinkhorn_fallbacks(reset=True)       # after the timed region (it synchronises)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_pairs = args.total_pairs if args.total_pairs > 0 else pairs * steps * n_gpus
    value = total_pairs / dt

    # the path's only exchange: the last step's matches of every rank -> rank 0 (RCCL), outside the clock
    if out is not None:
        per_pair = batch.split_by_pair(out, cap)
        local = [(rank + i * n_gpus, ml, mr) for i, (ml, mr) in enumerate(per_pair)]
    else:
        local = []                                       # a rank that owns no pair (strong scaling with few pairs)
    barrier()
    t0 = time.perf_counter()
    gathered = shard.gather_matches(local, pairs * n_gpus)
    barrier()
    gather_ms = 1e3 * (time.perf_counter() - t0)
    matches_per_pair = gather_bytes = None
    if rank == 0:
        got = [g_ for g_ in gathered if g_ is not None]
        assert len(got) == len(gathered) or args.total_pairs > 0, "gather_matches lost a pair"
        matches_per_pair = float(np.mean([g_[0].shape[0] for g_ in got])) if got else 0.0
        gather_bytes = sum(g_[0].shape[0] for g_ in got) * 16
    rank_ms, rank_setup = [rank_ms_per_step], [setup_s]
    if dist is not None:
        tl = [torch.zeros(2, device=dev, dtype=torch.float64) for _ in range(n_gpus)]
        dist.all_gather(tl, torch.tensor([rank_ms_per_step, setup_s], device=dev, dtype=torch.float64))
        rank_ms, rank_setup = [float(x[0].item()) for x in tl], [float(x[1].item()) for x in tl]

    res, other = None, []
    if out is not None:
        P_step = int(out["P"].item())
        rows_step = int(out["rows"].chunk_base[-1].item())
        third_ms = np.array([a.elapsed_time(b) for a, b in ev["third"]])
        fine_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fine"]]))
        # algorithmic HBM bytes per problem of the fused third-level kernel: both descriptor blocks in
        # (2 x 128 x 65 fp32), areas + coarse points in, 16 matches + labels + flags out; the 65x65 plan
        # stays on chip (SURVEY 8d "cost build: 4*D*(M+N) in, 0 out if fused")
        BYTES_PER_PROBLEM = 2 * 128 * 65 * 4 + 64 * 4 + 2 * 2 * 8 + 2 * 16 * 2 * 4 + 16 * 2 * 4 + 16
        t_ach = float(BYTES_PER_PROBLEM * P_step / (third_ms.mean() * 1e-3) / 1e9)
        t_valu = float(2.0 * 2.0 * ITERS * 65 * 65 * P_step / (third_ms.mean() * 1e-3) / 1e12)
        # HBM traffic per launch from rocprofv3 PMC passes over this same step (tools/pmc_step.sh -> profiles/r03_pmc_step.json:
        # FETCH_SIZE and WRITE_SIZE in separate runs, calibrated on the cost build's known byte count in the same run)
        pmc, pmc_src, pmc_age = {}, None, None
        pj, pmc_name = None, None
        if pj_live is not None:                          # three steps under rocprofv3 --pmc (two passes), taken before this process
            pj, pmc_name = pj_live, "live: bench.py ran itself under rocprofv3 at the start of this run"     # allocated its own 143 GB
        if pj is None:
            import glob
            cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_step_%s.json" % args.maps)))
            if cands:
                pj, pmc_name = json.load(open(cands[-1])), "profiles/" + os.path.basename(cands[-1])
        if pj is not None and int(pj.get("rows_cap", -1)) == cap.rows_cap and args.workload == "megadepth":
            pmc = pj["kernels"]
            pmc_src = "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes over bench.py --steps 3), factors " \
                      "from the cost build's known read count in the same run (x%.2f reads; writes x%.2f: tools/write_patterns.hip; the third-level " \
                      "kernel's 8-byte lane loads x1.38 as calibrated in round 2)" \
                      % (pmc_name, pj["calibration"]["fetch_factor"], pj["calibration"]["write_factor"])
            # how old is the figure?  kernel sources whose content differs from what the PMC run measured
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import pmc_step
            now, then = pmc_step.csrc_sha16(), pj.get("csrc_sha16")
            pmc_age = {"source": pmc_name, "kernel_sources_changed_since": sorted(k for k in now if then.get(k) != now[k]) if then else "unknown (no hashes in the file)"}

        def traffic_of(prefix):
            # (kernel names in the PMC file carry their template arguments - `fine_desc_kernel<0> grid=..` -: match with and without them)
            import re
            hit = [v for k, v in pmc.items() if k.startswith(prefix) or re.sub(r"<[^<>]*>", "", k).startswith(prefix)]
            return (float(max(hit, key=lambda v: v["hbm_bytes"])["hbm_bytes"]), pmc_src) if hit else (None, None)
        # ("_pmc": the kernel-name prefixes a roofline's traffic is summed over - the live passes at the end of the run re-fill it)
        P_COST, P_OT145 = "pats::cost_mfma_kernel grid=%d" % (cap.rows_cap * 256), "pats::sinkhorn_blk145"
        traffic, traffic_src = traffic_of("pats::third_fused3_kernel")
        third_roof = {"bound": "hbm", "kernel": "third_fused3_kernel (fused third level, %d problems per launch over a capacity of %d)"
                      % (P_step, cap.P_cap), "achieved": t_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": t_ach / HBM_PEAK_GBS,
                      "traffic": traffic, "_pmc": ["pats::third_fused3_kernel"], "traffic_unit": "bytes per launch", "traffic_source": traffic_src, "traffic_age": pmc_age,
                      "algorithmic_bytes_per_launch": float(BYTES_PER_PROBLEM * P_step), "avg_launch_ms": float(third_ms.mean()),
                      "launches": int(len(third_ms)), "algorithmic_bytes_per_problem": BYTES_PER_PROBLEM,
                      "valu_frac": t_valu / F32_PEAK_TFLOPS, "valu_tflops": t_valu,
                      "note": "fused cost build + 100 linear-domain Sinkhorn sweeps + Compute_result per 65x65 problem, one wave each, the "
                              "block held in registers; descriptors are read once, the plan never reaches HBM.  HBM is the nearer of the "
                              "two allowed rooflines but not the limiter: the sweeps are fp32 VALU work (valu_frac = sweep FMA flops / "
                              "157.3 TF/s vector peak)"}
        # fine level: descriptors in, log-plan out
        f_by = (2.0 * 264 * 145 * 4 + 145 * 145 * 4) * rows_step
        f_ach = f_by / (fine_ms * 1e-3) / 1e9
        f_parts = [traffic_of("pats::cost_mfma_kernel grid=%d" % (cap.rows_cap * 256))[0], traffic_of("pats::sinkhorn_blk145")[0]]
        f_traffic = float(sum(f_parts)) if all(v is not None for v in f_parts) else None
        fine_roof = {"bound": "hbm", "kernel": "fine-level launch pair as timed inside the steps: cost_mfma_kernel + sinkhorn_blk145[w2]_kernel (%d x 145x145 = the row capacity, %d rows in use)"
                     % (cap.rows_cap, rows_step), "achieved": f_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f_ach / HBM_PEAK_GBS,
                     "traffic": f_traffic, "_pmc": [P_COST, P_OT145], "traffic_unit": "bytes per launch pair (cost_mfma_kernel + sinkhorn_blk145[w2]_kernel: includes the "
                     "score matrix written by the first and read by the second)", "traffic_source": pmc_src if f_traffic is not None else None,
                     "algorithmic_bytes_per_launch": f_by, "avg_launch_ms": fine_ms, "launches": int(len(ev["fine"])),
                     "valu_frac": 2.0 * 2.0 * ITERS * 145 * 145 * rows_step / (fine_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS,
                     "note": "descriptors in (2 x 264 x 145 fp32), log-plan out (145 x 145 fp32) per problem; the 100 sweeps run on the "
                             "register-resident blocks (VALU-bound)"}
        # The fine level is TWO kernels inside one C call (pats_cost_ot_flags_counted_f32): the contract's roofline is per kernel,
        # so the call records an event between its two launches (ops.set_cost_ot_mid_event, armed by batch.fine_solve_stage) and
        # both are timed INSIDE the timed steps (round 3 re-timed them on their own afterwards: 3.39 against 3.80 ms in the trace)
        fine_split = None
        if ev.get("fine_mid") and len(ev["fine_mid"]) == len(ev["fine"]):
            c_ms = float(np.mean([a.elapsed_time(m_) for (a, _), m_ in zip(ev["fine"], ev["fine_mid"])]))
            s_ms = float(np.mean([m_.elapsed_time(b_) for (_, b_), m_ in zip(ev["fine"], ev["fine_mid"])]))
            fine_split = (c_ms, s_ms)
        split_roofs = []
        w2 = os.environ.get("PATS_FINE_W2", "1") != "0"
        fine_kernel = "sinkhorn_blk145w2_kernel<2>" if w2 else "sinkhorn_blk145_kernel<2>"
        if fine_split is not None:
            c_ms, s_ms = fine_split
            c_by = (2.0 * 264 * 145 * 4 + 145 * 145 * 4) * rows_step
            s_by = (2.0 * 145 * 145 * 4 + 144 * 4) * rows_step
            c_tr = traffic_of("pats::cost_mfma_kernel grid=%d" % (cap.rows_cap * 256))[0]
            s_tr = traffic_of("pats::sinkhorn_blk145")[0]
            split_roofs = [
                {"bound": "hbm", "kernel": "%s (fine-level OT: %%d x 145x145 in use of a capacity of %%d, 100 sweeps)" % fine_kernel % (rows_step, cap.rows_cap),
                 "achieved": s_by / (s_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": s_by / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "traffic": s_tr, "_pmc": [P_OT145], "traffic_unit": "bytes per launch", "traffic_source": pmc_src if s_tr is not None else None,
                 "algorithmic_bytes_per_launch": s_by, "avg_launch_ms": s_ms, "launches": int(len(ev["fine"])),
                 "valu_frac": 2.0 * 2.0 * ITERS * 145 * 145 * rows_step / (s_ms * 1e-3) / 1e12 / F32_PEAK_TFLOPS,
                 "timed": "inside the timed steps (event recorded between the two launches of the one C call); in-step pair %.3f ms" % fine_ms,
                 "note": "scores in, log-plan out (2 x 145 x 145 fp32 per problem); HBM is the nearer allowed roofline but not the limiter: "
                         "100 sweeps on register-resident blocks, VALU issue (valu_frac = sweep FMA flops / 157.3 TF/s).  "
                         + ("Two waves per problem, 9 x 18 blocks: 295 VALU instructions per wave and sweep, 164 of them packed FMAs, "
                            "one barrier; VALU 73 % busy at two waves per SIMD" if w2 else
                            "Four waves per problem, 9 x 9 blocks (PATS_FINE_W2=0): 208 VALU instructions per wave and sweep, 75 of them "
                            "packed, three barriers; VALU 95 % busy")},
                {"bound": "hbm", "kernel": "cost_mfma_kernel<true> (fine-level cost build: %d x [264,145]^2 in use of a capacity of %d)" % (rows_step, cap.rows_cap),
                 "achieved": c_by / (c_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": c_by / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "traffic": c_tr, "_pmc": [P_COST], "traffic_unit": "bytes per launch", "traffic_source": pmc_src if c_tr is not None else None,
                 "algorithmic_bytes_per_launch": c_by, "avg_launch_ms": c_ms, "launches": int(len(ev["fine"])),
                 "timed": "inside the timed steps; in-step pair %.3f ms" % fine_ms,
                 "note": "both descriptor blocks in, the score matrix out: a streaming kernel (the fp16-split MFMA passes hide under the "
                         "descriptor stream)"}]
        # the two descriptor gathers (a15 / a16): HBM-bound copies with index arithmetic
        fd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fine_desc"]]))
        td_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["third_desc"]]))
        # a15, algorithmic bytes per stacked image: every sampled input element once (64 ch x 144 points x 4 pooled taps on the
        # two high-resolution maps, 128 ch x 144 on the third, title + dustbin features) and the [264,145] block out
        FD_BYTES = (2 * 64 * 144 * 4 + 128 * 144 + 8 + 264) * 4 + 264 * 145 * 4
        fd_by = float(FD_BYTES) * 2 * rows_step
        cl = nets.channels_last
        fd_name, td_name = ("fine_desc_nhwc_kernel", "third_desc_nhwc_kernel") if cl else ("fine_desc_kernel", "third_desc_kernel")
        fd_roof = {"bound": "hbm", "kernel": "%s (a15: fine descriptor sampling, %d stacked crops, %s maps)"
                                             % (fd_name, 2 * rows_step, "channels-last" if cl else "NCHW"),
                   "achieved": fd_by / (fd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": fd_by / (fd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_of("pats::" + fd_name + " ")[0], "_pmc": ["pats::" + fd_name + " "],
                   "traffic_unit": "bytes per launch", "traffic_source": pmc_src, "algorithmic_bytes_per_launch": fd_by,
                   "algorithmic_bytes_per_image": FD_BYTES, "avg_launch_ms": fd_ms, "launches": int(len(ev["fine_desc"])),
                   "note": ("reads every sampled pixel of the three backbone maps once - a pixel's 64 / 128 channels are one run of "
                            "256 / 512 bytes in channels-last memory, so every 64-byte granule fetched is used in full - turns the "
                            "64-channel tiles through LDS and writes the [2,B,264,145] block as float4") if cl else
                           ("reads every sampled element of the three backbone maps once and writes the [2,B,264,145] block; the "
                            "2x2 pooled taps use 8 of every 16 bytes of half of the rows of the NCHW 48x48 maps, so the granules "
                            "touched are about 1.4x the algorithmic bytes (--maps nhwc: the channels-last gather)")}
        # a16: two 8x8 windows x 128 channels in, two [128,65] blocks out per point
        TD_BYTES = 2 * 128 * 64 * 4 + 128 * 4 + 2 * 128 * 65 * 4 + 2 * 2 * 4 + 8 + 2 * 2 * 8
        td_by = float(TD_BYTES) * P_step
        td_roof = {"bound": "hbm", "kernel": "%s (a16: third-level window gather, %d points, %s maps)"
                                             % (td_name, P_step, "channels-last" if cl else "NCHW"),
                   "achieved": td_by / (td_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": td_by / (td_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_of("pats::" + td_name + " ")[0], "_pmc": ["pats::" + td_name + " "],
                   "traffic_unit": "bytes per launch", "traffic_source": pmc_src, "algorithmic_bytes_per_launch": td_by,
                   "algorithmic_bytes_per_point": TD_BYTES, "avg_launch_ms": td_ms, "launches": int(len(ev["third_desc"])),
                   "note": ("a window cell is one 512-byte run (128 channels) of the channels-last 52x52 map: 64 such runs in per "
                            "(point, side), turned through LDS into the [128,65] block the cost build reads, kenc added on the way "
                            "out; XCD-aware workgroup order") if cl else
                           ("a window row is 32 bytes of a 208-byte row of a channel-major 52x52 map: 2.75 64-byte granules fetched "
                            "per 32 bytes used unless neighbouring points meet in L2 (XCD-aware workgroup order); --maps nhwc: the "
                            "channels-last gather")}
        # ranked by single KERNELS; the fine level's launch pair as measured inside the steps stays in the list for the cross-check
        ranked = sorted([third_roof, fd_roof, td_roof] + (split_roofs if split_roofs else [fine_roof]), key=lambda r: -r["avg_launch_ms"])
        dominant, other = ranked[0], ranked[1:] + ([fine_roof] if split_roofs else [])
        dominant["traffic_age"] = pmc_age                # where roofline.traffic comes from and which kernel sources changed since
        sweeps_per_pair = ITERS * (1 + (rows_step + P_step) / float(pairs))
        res = {
            "metric": "image-pairs/sec (coarse+fine OT) on 640x480 MegaDepth; OT iters/sec per pair",
            "value": value, "unit": "pairs/s", "n_gpus": n_gpus, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / max(steps, 1), "higher_is_better": True,
            "scaling": "strong" if args.total_pairs > 0 else "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": label + ": coarse + fine + third OT, cost volumes, area expansions, subdivision gather (crops), "
                                           "descriptor gathers (a15 / a16) from synthetic backbone maps, merge_patches per chunk in order, "
                                           "third-level inputs decided by the merge, result scatter, get_result",
                       "pairs_per_step_per_rank": pairs,
                       "batching": "each stage is one launch over all pairs and chunks of the step (pats_amd.batch); no host read inside a step"
                                   + ("" if streams is None else "; two HIP streams with disjoint CU masks (%d / %d of every 8 CUs per shader engine): "
                                      "the gathers + crops + coarse level of batch i beside the fine and third-level solvers of batches i, i - 1"
                                      % (min(max(args.overlap, 1), 7), 8 - min(max(args.overlap, 1), 7))),
                       "L1": "%d x [448,%d]^2 -> %dx%d (every pair its own descriptors)" % (pairs, h * w, h * w + 1, h * w + 1),
                       "L2": "%d rows x [264,145]^2 -> 145x145 in use per step (%.1f per pair; row capacity %d, at most %d chunks per pair)"
                             % (rows_step, rows_step / float(pairs), cap.rows_cap, cap.Cmax),
                       "L3": "%d x [128,65]^2 -> 65x65 per step, decided by the merge (%.1f per pair; capacity %d)"
                             % (P_step, P_step / float(pairs), cap.P_cap),
                       "map_layout": ("torch.channels_last (logical [B,C,H,W], memory [B,H,W,C]) for the five backbone maps the gathers read: "
                                      "what the reference's backbones emit after ops.prepare_backbones(model) (value_nhwc)"
                                      if nets.channels_last else
                                      "NCHW-contiguous backbone maps: what the UNCHANGED reference's backbones emit (second_layer.py:66-69, "
                                      "third_layer.py:113-117); value = value_nchw.  value_nhwc = the same steps on torch.channels_last maps"),
                       "rows_cap": "%s (%d rows; %d in use - the fine level's launches take the count from the device and skip the rest)"
                                   % ("worst case pairs * (N + (Cmax - 1) w)" if args.rows_cap == "worst" else "dry run of the step's own pairs + 1 %",
                                      cap.rows_cap, rows_step),
                       "result_handover": "inside the timed region, every step: the matches are regrouped by pair on the device "
                                          "(pats_matches_by_pair_f32: each pair's list contiguous, in the reference's order) and the step's "
                                          "status / P / M counters + per-pair offsets go to pinned host memory, checked one step behind (an "
                                          "overflow in any step raises).  The match coordinates themselves stay in HBM; their gather to rank 0 "
                                          "runs once, after the clock",
                       "setup_s": setup_s,
                       "resident_synthetic_GB": nets.resident_bytes() / 1e9, "sinkhorn_iters": ITERS,
                       "parallelism": "pairs sharded over %d rank(s), no data-path collective; matches gathered to rank 0 "
                                      "after the timed region (%s)" % (n_gpus, backend if dist is not None else "single process")},
            "ot_iters_per_sec": value * sweeps_per_pair,
            "rows_in_use_per_step": rows_step, "rows_cap": cap.rows_cap, "third_problems_per_step": P_step,
            "guard_fallbacks_per_step": fallbacks / max(1, steps),
            "gather_ms": gather_ms, "matches_per_pair": matches_per_pair,
            "rank_ms_per_step": rank_ms, "rank_setup_s": rank_setup,
            "roofline": dominant,
        }
    if rank == 0:
        assert res is not None, "rank 0 owns no pair"
        if dist is not None:
            res["gather_bytes"] = gather_bytes
        res["value_" + args.maps] = value
        if not args.no_secondary and n_gpus == 1:
            # the same steps on the same logical maps in the OTHER memory order (re-laid in place, one tensor at a time)
            other_maps = "nhwc" if args.maps == "nchw" else "nchw"
            nets.set_layout(other_maps == "nhwc")
            run_steps(batch, nets, cap, wl, None, 1, None)
            watch2 = StepWatch(cap)                      # (its pinned buffers before the clock, as in the headline leg: nine
            torch.cuda.synchronize()                     #  hipHostMalloc calls once cost 2 s inside this leg on one box)
            t1 = time.perf_counter()
            run_steps(batch, nets, cap, wl, None, steps, None, watch2)
            torch.cuda.synchronize()
            res["value_" + other_maps] = pairs * steps / (time.perf_counter() - t1)
            nets.set_layout(args.maps == "nhwc")
            if args.overlap == 0:
                # the same steps with the HBM-bound stages of one batch beside the VALU-bound stages of its neighbour (two streams with
                # disjoint CU masks, `--overlap 3`): what the pairing buys - reported beside the headline, which stays on ONE stream
                # (every kernel with the whole GPU: per-kernel rooflines unambiguous, the per-step status watch in place)
                try:
                    n_cu_ = torch.cuda.get_device_properties(dev).multi_processor_count
                    st2 = (ops.masked_stream([c for c in range(n_cu_) if c // 32 < 3]), ops.masked_stream([c for c in range(n_cu_) if c // 32 >= 3]))
                    run_steps(batch, nets, cap, wl, None, 2, st2)
                    torch.cuda.synchronize()
                    passes = []                          # three timed passes: in the FIRST bench process on a fresh box the first pass of this
                    for _ in range(3):                   # leg carries a one-off stall of ~1.3 s (round 6: 514-564 pairs/s at 20 steps, 324 at
                        t2 = time.perf_counter()         # 10, against 1 900-2 010 in a second process or stand-alone; cause not found)
                        run_steps(batch, nets, cap, wl, None, steps, st2)
                        torch.cuda.synchronize()
                        passes.append(pairs * steps / (time.perf_counter() - t2))
                    res["value_two_masked_streams"] = {"pairs_per_s": max(passes), "passes_pairs_per_s": passes, "overlap": 3,
                                                       "note": "gathers + crops of batch i on 3 of every 8 CUs of each shader engine beside the solvers of "
                                                               "batch i - 1 on the other 5 (bench.py --overlap 3); best of three passes of `steps` steps; not "
                                                               "the headline"}
                    del st2
                except Exception as e:                   # noqa: BLE001  (a runtime without CU-mask streams)
                    res["value_two_masked_streams"] = {"error": repr(e)[:200]}
            gnn, gnn_roof = gnn_secondary(ops, dev, pairs, rows_step, P_step, 1e3 * dt / max(steps, 1), wl["outdoor"])
            res["gnn"] = gnn
            res["gnn"]["measured"] = with_gnn_leg(ops, batch, dev, nets, cap, wl, h, w, 2)
            res["roofline_secondary"] = other + [gnn_roof] + secondary_rooflines(ops, dev)
            res["guard_trips"] = guard_trip_sweep(ops, batch, nets, cap, wl)
            res["step_determinism"] = step_determinism(batch, nets, cap, wl)
            res["gather_layouts"] = gather_layout_ab(ops, dev, cap, P_step)
            torch.cuda.empty_cache()
        if not args.no_latency and n_gpus == 1:
            # the reference's execution mode: ONE pair at a time (evaluate.py:20-35), chunk by chunk and as a whole pair, with and
            # without the layers' heads; ms per pair, launches and host reads per pair (tools/latency.py)
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import latency as latency_mod
            from pats_amd import pipeline
            try:
                res["latency"] = latency_mod.latency_leg(ops, batch, pipeline, sys.modules[__name__], dev, args.workload, n=20, with_gnn=True)
            except Exception as e:                       # noqa: BLE001
                res["latency"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline and n_gpus == 1:
            # one more step outside the clock, keeping the coarse tensors the parity leg needs
            o2 = batch.forward_pairs(nets.lefts, nets.rights, nets, cap, if_outdoor=wl["outdoor"], merge_new=wl["merge_new"], iters=ITERS)
            res["cpu_baseline"] = cpu_baseline(ops, batch, dev, nets, cap, wl, o2)
        else:
            res["cpu_baseline"] = None
        if not args.no_secondary and n_gpus == 1 and args.workload == "megadepth" and args.total_pairs == 0 and args.wild == 0.0:
            # BASELINE.json configs[2] / configs[3] in the same line: the headline's resident set goes first (two do not fit side by side)
            import gc
            out = o2 = watch = ev = gathered = local = per_pair = None
            nets.__dict__.clear()
            del nets
            gc.collect()
            torch.cuda.empty_cache()
            try:
                res["workloads_secondary"] = secondary_workloads(ops, batch, dev, rank, maps=args.maps)
            except Exception as e:                       # noqa: BLE001
                res["workloads_secondary"] = {"error": repr(e)[:300]}
            if want_live_pmc:
                pj_live = live_pmc(args)                 # nothing of this process is resident any more: the child fits
                roofs = [res["roofline"]] + [r for r in res.get("roofline_secondary", []) if isinstance(r, dict)]
                if pj_live is not None and int(pj_live.get("rows_cap", -1)) == cap.rows_cap:
                    import re
                    src = "live: bench.py ran itself under rocprofv3 at the end of this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes " \
                          "over bench.py --steps 3), factors from the cost build's known read count in the same run (x%.2f reads; writes x%.2f: tools/write_patterns.hip; the " \
                          "third-level kernel's 8-byte lane loads x1.38 as calibrated in round 2)" \
                          % (pj_live["calibration"]["fetch_factor"], pj_live["calibration"]["write_factor"])
                    for r in roofs:
                        parts = []
                        for prefix in r.get("_pmc", []):
                            hit = [v for k, v in pj_live["kernels"].items() if k.startswith(prefix) or re.sub(r"<[^<>]*>", "", k).startswith(prefix)]
                            parts.append(float(max(hit, key=lambda v: v["hbm_bytes"])["hbm_bytes"]) if hit else None)
                        if parts and all(v is not None for v in parts):
                            r["traffic"], r["traffic_source"] = float(sum(parts)), src
                            if "traffic_age" in r:
                                r["traffic_age"] = {"source": "live (this run)", "kernel_sources_changed_since": []}
        for r in [res.get("roofline", {})] + [r for r in res.get("roofline_secondary", []) if isinstance(r, dict)]:
            r.pop("_pmc", None)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
