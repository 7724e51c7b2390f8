#!/usr/bin/env python3
"""Latency of ONE pair at a time - the reference's execution mode (evaluate.py:20-35: batch_size 1, one PATS.forward per pair;
models/pats.py:18-85: chunk by chunk, host reads between the layers).  bench.py's headline is the throughput mode.

Legs (the same synthetic 640x480 pair, network outputs resident in HBM, as bench.py's BenchNets holds them at pairs = 1):
  a        pipeline.forward_path(batch_chunks=False): the reference's control flow - one fine / third launch chain per chunk,
           host reads for the chunk plan, P and M
  a_batch  pipeline.forward_path(batch_chunks=True): the chunks of the pair together, three host reads
  b        batch.forward_pairs(pairs=1) + the per-pair hand-over (counts and offsets read back: the one synchronisation)
  b_graph  the same chain captured ONCE in a HIP graph over the capacities and replayed per pair
  c_*      a / b / b_graph with the layers' heads inside (GnnNets: KeypointEncoder, 18 / 18 / 10 GNN layers, final_proj, scale heads)
For each leg: ms per pair (wall clock, synchronised at the hand-over of every pair), kernel launches per pair (kineto /
graph nodes) and host reads per pair (torch's sync-debug warnings).

`latency_leg(...)` is what bench.py puts into its line under "latency"; run as a script for one JSON line per leg
(`--profile` cuts the kernel trace to the timed pairs with the library's marker kernel).
"""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


class PipelineNets:
    """pipeline.forward_path's per-chunk callbacks on a pairs = 1 BenchNets (bench.py) - the same resident synthetic backbone
    maps, cut per chunk (the cuts are made once, in the warm-up pair: they stand for the backbone's output of that chunk) -
    and, with `gnn` (a bench.GnnNets over the same base), the layers' heads per chunk."""

    def __init__(self, base, ops, gnn=None):
        self.base, self.ops, self.gnn = base, ops, gnn
        self.cache, self.off, self.cur = {}, 0, (0, 0)
        R = base.cap.rows_cap
        reps = (R * 144 + base.scale3.shape[0] - 1) // base.scale3.shape[0]
        self.scale3 = base.scale3.repeat(reps, 1, 1)[:R * 144].contiguous()         # one scale-head row per third-level SLOT of any chunk layout

    def coarse(self, left, right):
        self.off = 0
        return self.gnn.coarse(left, right) if self.gnn is not None else self.base.coarse(left, right)

    def _cut(self, key, off, B):
        k = (key, off, B)
        if k not in self.cache:
            b, R = self.base, self.base.cap.rows_cap
            fmt = torch.channels_last if b.channels_last else torch.contiguous_format
            maps = [m.reshape((2, R) + tuple(m.shape[1:]))[:, off:off + B].reshape((2 * B,) + tuple(m.shape[1:])).contiguous(memory_format=fmt)
                    for m in (b.m0, b.m1, b.m2)]
            self.cache[k] = maps
        return self.cache[k]

    def fine(self, num, new_left, new_right, mask, sizes=None):
        b, ops, g = self.base, self.ops, self.gnn
        B = int(new_left.shape[0])
        off = 0 if num is None else self.off
        if off + B > b.cap.rows_cap:
            off = 0
        self.cur = (off, B)
        self.off = off + B
        desc = ops.fine_descriptors(self._cut("f", off, B), b.title[off:off + B], b.rubbish[off:off + B])       # a15
        sl = slice(off, off + B)
        if g is None:
            return desc[0], desc[1], b.sx[sl], b.sy[sl], b.ns2[sl]
        d0, d1 = ops.attentional_gnn(desc[0], desc[1], g.gnn2, g.names18)                                        # second_layer.py:89
        m0, m1 = ops.conv1d(d0, *g.proj2), ops.conv1d(d1, *g.proj2)
        _, (sx, sy) = ops.scale_head(m1, 12, 12, [g.sx2[0], g.sy2[0]], [g.sx2[1], g.sy2[1]], return_heads=True)
        return m0, m1, sx.contiguous(), sy.contiguous()

    def third(self, num, mk0, mk1, b_ids, sizes=None, count=None):
        """count (device int64 [1]): the tensors are a capacity (pipeline.forward_chunks_device); the roundings of
        ops.third_descriptors are handed back with the descriptors."""
        b, ops, g = self.base, self.ops, self.gnn
        off, B = self.cur
        P = int(mk0.shape[0])
        kenc = b.kenc
        if g is not None:
            from pats_amd import heads
            kenc = ops.keypoint_encoder(heads.grid_kpts(8, 8, mk0.device), g.kenc3).reshape(128, 64)
        t0, t1, ps, pt = ops.third_descriptors(b.ff0[off:off + B], b.ff1[off:off + B], mk0, mk1, b_ids, kenc, b.rubbish3[off:off + B],
                                               count=count)                                                      # a16
        if P > self.scale3.shape[0]:
            raise RuntimeError("latency: %d third-level slots exceed the synthetic scale pool" % P)
        if g is None:
            return (t0, t1, self.scale3[:P]) + ((ps, pt) if count is not None else ())
        f0, f1 = ops.attentional_gnn(t0, t1, g.gnn3, g.names10, count=count)                                     # third_layer.py:146-148
        return (f0, f1, ops.scale_head(f1, 8, 8, [g.scale3[0]], [g.scale3[1]])) + ((ps, pt) if count is not None else ())


class Handover:
    """The per-pair hand-over of the batched path: offsets / M / P / status - batch.group_by_pair's `summary` - into pinned memory,
    ONE copy, one synchronisation."""

    def __init__(self, cap):
        self.cap = cap
        self.buf = torch.empty(cap.pairs + 4, dtype=torch.int64).pin_memory()

    def __call__(self, out):
        self.buf.copy_(out["summary"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        M, P, status = self.buf[self.cap.pairs + 1:].tolist()
        if status or P > self.cap.P_cap:
            raise RuntimeError("latency: capacity overflow (status %d, P %d of %d)" % (status, P, self.cap.P_cap))
        return M


def count_host_reads(fn):
    """Synchronising torch calls of one fn() (torch.cuda.set_sync_debug_mode: .item(), .cpu(), nonzero, mask indexing ...)."""
    prev = torch.cuda.get_sync_debug_mode()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        try:
            fn()
        finally:
            torch.cuda.set_sync_debug_mode(prev)
    return sum(1 for x in w if "synchroniz" in str(x.message))


def count_launches(fn):
    """Kernel launches of one fn(): every kernel the HIP runtime starts in this process (the library's through ctypes and torch's
    own), seen by kineto.  None if the profiler is unavailable."""
    try:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            fn()
            torch.cuda.synchronize()
        n = mc = 0
        for e in prof.events():
            if str(getattr(e, "device_type", "")).endswith("CUDA"):
                name = e.name or ""
                if name.startswith("Memcpy") or name.startswith("Memset"):
                    mc += 1
                else:
                    n += 1
        return {"kernels": n, "copies_and_fills": mc}
    except Exception as e:                                   # noqa: BLE001
        return {"error": repr(e)[:160]}


def time_pairs(fn, n, warm=2, marker=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if marker is not None:
        marker(1)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()                                                 # fn synchronises at its own hand-over
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    if marker is not None:
        marker(2)
    ts = np.array(ts) * 1e3
    return {"ms_per_pair": float(ts.mean()), "ms_min": float(ts.min()), "ms_p50": float(np.median(ts)), "ms_max": float(ts.max()), "pairs": int(n)}


def latency_leg(ops, batch, pipeline, bench, dev, workload="megadepth", n=20, with_gnn=True, legs=None, profile=False, counts=True):
    h, w, if_local, outdoor, _, label = bench.WORKLOADS[workload]
    cap = batch.Capacities(1, h, w, if_local=if_local)
    gen = torch.Generator(device=dev)
    gen.manual_seed(bench.synth.SEED + 1000)
    base = bench.BenchNets(ops, dev, gen, cap, h, w, batch=batch, channels_last=False)
    marker = ops.profile_marker if profile else None
    kw = dict(if_outdoor=outdoor, merge_new=outdoor, iters=bench.ITERS)
    res = {"workload": label, "what": "ONE pair at a time, synchronised at every pair's hand-over (the reference's mode: evaluate.py:20-35); "
                                      "network outputs synthetic and resident (NCHW maps), %d pairs timed per leg" % n}
    left, right = base.lefts[0:1], base.rights[0:1]

    def leg(name, fn, n_):
        if legs is not None and name not in legs:
            return
        r = time_pairs(fn, n_, marker=marker)
        if counts:
            r["host_reads_per_pair"] = count_host_reads(fn)
            r["launches_per_pair"] = count_launches(fn)
        res[name] = r

    def make(nets_b, nets_p, tag):
        hand = Handover(cap)
        info = {}

        def run_a(batched, **more):
            o = pipeline.forward_path(left, right, nets_p, if_local=if_local, batch_chunks=batched, **kw, **more)
            info["M"] = int(o["matches_l"].shape[0])         # (a shape: the host already knows it)
            torch.cuda.current_stream().synchronize()
            return o

        def run_b():
            t0 = time.perf_counter()
            o = batch.forward_pairs(left, right, nets_b, cap, **kw)
            batch.group_by_pair(o, cap)
            info.setdefault("enq", []).append(time.perf_counter() - t0)      # host time to queue the pair's launches
            info["Mb"] = hand(o)
            return o
        leg(tag + "a_reference_control_flow", lambda: run_a(False), n)
        if tag + "a_reference_control_flow" in res:
            res[tag + "a_reference_control_flow"]["matches"] = info.get("M")
        for nm, more in (("a_device_counts", dict(device_counts=True)), ("a_device_counts_2_streams", dict(device_counts=True, streams=2)),
                         ("a_device_counts_3_streams", dict(device_counts=True, streams=3)),
                         ("a_device_counts_4_streams", dict(device_counts=True, streams=4))):
            leg(tag + nm, lambda: run_a(False, **more), n)
            if tag + nm in res:
                res[tag + nm]["matches"] = info.get("M")
        leg(tag + "a_chunks_batched", lambda: run_a(True), n)
        leg(tag + "b_forward_pairs_1", run_b, n)
        if tag + "b_forward_pairs_1" in res:
            res[tag + "b_forward_pairs_1"]["matches"] = info.get("Mb")
            res[tag + "b_forward_pairs_1"]["host_enqueue_ms"] = 1e3 * float(np.median(info["enq"]))
        name = tag + "b_graph"
        if legs is None or name in legs:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(2):
                        run_b()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    o = batch.forward_pairs(left, right, nets_b, cap, **kw)
                    batch.group_by_pair(o, cap)

                def run_g():
                    g.replay()
                    info["Mg"] = hand(o)
                r = time_pairs(run_g, n, marker=marker)
                r["matches"] = info.get("Mg")
                if counts:
                    r["host_reads_per_pair"] = count_host_reads(run_g)
                    r["launches_per_pair"] = count_launches(run_g)
                res[name] = r
                del g
            except Exception as e:                           # noqa: BLE001
                res[name] = {"error": repr(e)[:300]}
    make(base, PipelineNets(base, ops), "")
    if with_gnn:
        gn = bench.GnnNets(base, ops, dev, h, w)
        # the layers' overflow flag is read once, behind all legs (ops.set_gnn_redo("deferred")): no gated fp32 redo chains
        prev = ops.set_gnn_redo("deferred")
        ops.gnn_overflows(reset=True)
        try:
            make(gn, PipelineNets(base, ops, gnn=gn), "c_")
            res["c_gnn_redo"] = "deferred; overflow flag behind the legs: %s" % ("RAISED - c_ legs void" if ops.gnn_overflows(reset=True) else "not raised")
        finally:
            ops.set_gnn_redo(prev)
        del gn
    del base
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="megadepth")
    ap.add_argument("--pairs", type=int, default=20)
    ap.add_argument("--no-gnn", action="store_true")
    ap.add_argument("--legs", default=None, help="comma-separated leg names")
    ap.add_argument("--profile", action="store_true", help="marker kernels around the timed pairs of every leg; no counting passes")
    a = ap.parse_args()
    import bench
    from pats_amd import batch, ops, pipeline
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = latency_leg(ops, batch, pipeline, bench, dev, a.workload, n=a.pairs, with_gnn=not a.no_gnn,
                      legs=set(a.legs.split(",")) if a.legs else None, profile=a.profile, counts=not a.profile)
    print(json.dumps(out))
