"""bench.py: the synthetic network outputs of a step (BenchNets), the same with the layers' heads inside (GnnNets), the step loop
(run_steps), the per-step status watch and the with-GNN leg."""
import json  # noqa: F401
import os
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import REPO, synth  # noqa: F401


def correlated_pair(shape, dev, gen, noise=0.3, amp=3.0, chunk=2048, channels_last=False):
    """[2, rows, ...]: two views of the same random base with independent noise - what a backbone makes of the left and
    the right crop of a matching patch.  Built in row chunks so that the temporaries stay small.
    channels_last: [2, rows, C, H, W] whose [rows, C, H, W] halves lie in torch.channels_last memory order."""
    if channels_last:
        r, c, hh, ww = shape
        out = torch.empty((2, r, hh, ww, c), dtype=torch.float32, device=dev).permute(0, 1, 4, 2, 3)
    else:
        out = torch.empty((2,) + tuple(shape), dtype=torch.float32, device=dev)
    for r0 in range(0, shape[0], chunk):
        sub = (min(chunk, shape[0] - r0),) + tuple(shape[1:])
        base = torch.randn(sub, device=dev, generator=gen)
        out[0, r0:r0 + sub[0]] = amp * (base + noise * torch.randn(sub, device=dev, generator=gen))
        out[1, r0:r0 + sub[0]] = amp * (base + noise * torch.randn(sub, device=dev, generator=gen))
    return out


def scale_head(shape, dev, gen):
    x = 0.3 * torch.randn(shape, device=dev, generator=gen)
    return torch.exp(torch.sigmoid(x) * synth.LN256 - synth.LN256 / 2)


class BenchNets:
    """The network outputs the path consumes, synthetic and RESIDENT in HBM before the timed region (the callbacks of
    pats_amd.batch): coarse descriptors per pair; per row of the fine level's table the three ResNet2.forward2 maps of its
    left / right crop, title / dustbin features and the two scale heads; per row the two half-resolution maps of the third
    level, its dustbin features, and one scale-head row per third-level problem slot.  Inside the step the callbacks only
    run the path's own gathers (a15: ops.fine_descriptors, a16: ops.third_descriptors); GNN + final_proj = identity."""

    def __init__(self, ops, dev, gen, cap, h, w, batch=None, channels_last=True, rows_cap_policy="worst"):
        self.ops = ops
        self.channels_last = cl = bool(channels_last)
        pairs, N = cap.pairs, h * w
        c = correlated_pair((pairs, 448, N), dev, gen)
        self.d0, self.d1 = c[0].contiguous(), c[1].contiguous()
        gone = torch.rand((pairs, 1, N), device=dev, generator=gen) < 0.03         # a few coarse cells without a partner
        self.d0 = torch.where(gone, 3.12 * torch.randn((pairs, 448, N), device=dev, generator=gen), self.d0).contiguous()
        self.ns = scale_head((pairs, 1, N), dev, gen)
        self.alpha = torch.tensor(0.0, device=dev)
        img = torch.randint(0, 256, (2, pairs, 32 * h, 32 * w, 3), device=dev, generator=gen).float()
        self.lefts, self.rights = img[0].contiguous(), (0.5 * img[1] + 0.5 * torch.roll(img[1], 1, dims=2)).contiguous()
        # row capacity: the worst case N + (Cmax - 1) w per pair by default (the fine level's launches cover the capacity; rows
        # past the device-side total are skipped by every kernel).  --rows-cap dry-run (round 3): a dry run of the coarse stage
        # tells how many rows the table holds for THESE pairs and the capacity becomes that + 1 % - the benchmark peeking at its
        # data, kept as an option only
        if batch is not None and rows_cap_policy == "dry-run":
            total = int(batch.coarse_stage(self.lefts, self.rights, self, cap, ITERS, fine_inputs="rows_only")["rows"].chunk_base[-1].item())
            cap.rows_cap = min(cap.rows_cap, (int(total * 1.01) + 63) // 64 * 64)
        self.cap = cap
        R, Pc = cap.rows_cap, cap.P_cap
        # fine level: ResNet2.forward2 maps of the stacked (left | right) crops, second_layer.py:69-70
        # memory order of the backbone maps: torch.channels_last (the default: what a backbone run under MIOpen emits, and
        # the order in which the gathers' per-pixel reads are contiguous) or NCHW (--maps nchw: a torch conv's default)
        self.m0 = correlated_pair((R, 64, 48, 48), dev, gen, channels_last=cl).reshape(2 * R, 64, 48, 48)
        self.m1 = correlated_pair((R, 64, 24, 24), dev, gen, channels_last=cl).reshape(2 * R, 64, 24, 24)
        self.m2 = correlated_pair((R, 128, 12, 12), dev, gen, channels_last=cl).reshape(2 * R, 128, 12, 12)
        assert all(m.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format) for m in (self.m0, self.m1, self.m2))
        self.title = 0.5 * torch.randn((R, 8), device=dev, generator=gen)
        self.rubbish = 1.5 * torch.randn((R, 264), device=dev, generator=gen)
        self.sx, self.sy = scale_head((R, 1, 144), dev, gen), scale_head((R, 1, 144), dev, gen)
        self.ns2 = (self.sx * self.sy).contiguous()
        # outputs of the two gathers, double-buffered: with the stages of consecutive batches on different streams the
        # gather of batch i + 1 writes while a solver of batch i still reads
        self.desc = [torch.empty((2, R, 264, 145), dtype=torch.float32, device=dev) for _ in range(2)]
        self.fine_calls = self.third_calls = 0
        self.ev = None                       # dict of lists of (start, end) HIP events while the timed steps run
        # third level: the 1/2-resolution maps (padded to 52x52) of both crops, third_layer.py:112-120
        f = correlated_pair((R, 128, 52, 52), dev, gen, chunk=1024, channels_last=cl)
        self.ff0, self.ff1 = f[0], f[1]
        assert self.ff0.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
        self.kenc = 0.1 * torch.randn((128, 64), device=dev, generator=gen)
        self.rubbish3 = 1.5 * torch.randn((R, 128, 144), device=dev, generator=gen)
        self.scale3 = scale_head((Pc, 1, 64), dev, gen)
        self.t0 = [torch.empty((Pc, 128, 65), dtype=torch.float32, device=dev) for _ in range(2)]
        self.t1 = [torch.empty((Pc, 128, 65), dtype=torch.float32, device=dev) for _ in range(2)]

    def set_layout(self, channels_last):
        """Re-lay the five backbone maps (same logical tensors) in the other memory order, one tensor at a time."""
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        for name in ("m0", "m1", "m2", "ff0", "ff1"):
            t = getattr(self, name)
            setattr(self, name, None)
            t2 = t.contiguous(memory_format=fmt)
            del t
            setattr(self, name, t2)
            assert t2.is_contiguous(memory_format=fmt)
        self.channels_last = bool(channels_last)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def resident_bytes(self):
        flat = []
        for v in vars(self).values():
            flat += v if isinstance(v, list) else [v]
        return sum(t.numel() * t.element_size() for t in flat if isinstance(t, torch.Tensor))

    def coarse(self, lefts, rights):
        return self.d0, self.d1, self.ns, self.alpha

    def _timed(self, tag):
        if self.ev is None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.ev.setdefault(tag, []).append((e0, e1))
        e0.record()
        return e1

    def fine(self, rows, new_left, new_right):
        desc = self.desc[self.fine_calls & 1]
        self.fine_calls += 1
        e = self._timed("fine_desc")
        self.ops.fine_descriptors([self.m0, self.m1, self.m2], self.title, self.rubbish, out=desc,
                                  count=rows.chunk_base[-1:])                                                 # a15
        if e is not None:
            e.record()
        return desc[0], desc[1], self.sx, self.sy, self.ns2

    def third(self, rows, mk0, mk1, b_ids, P_dev):
        k = self.third_calls & 1
        self.third_calls += 1
        e = self._timed("third_desc")
        t0, t1, ps, pt = self.ops.third_descriptors(self.ff0, self.ff1, mk0, mk1, b_ids, self.kenc, self.rubbish3,
                                                    count=P_dev, out=(self.t0[k], self.t1[k]))               # a16
        if e is not None:
            e.record()
        return t0, t1, self.scale3, ps, pt


class GnnNets:
    """BenchNets with the HEADS inside the step (round-4 verdict item 3): what the three layers run between their backbone and their
    optimal-transport problem - KeypointEncoder, the 18 / 18 / 10-layer AttentionalGNN stacks (first_layer.py:100-102,
    second_layer.py:89, third_layer.py:146-148), final_proj, the scale heads - on random weights, as callbacks of
    pats_amd.batch.forward_pairs.  The backbones stay what BenchNets holds (synthetic maps, resident).  Weights: the reference's
    initialisation (synth.gnn_params / kenc_params) with the LAST Conv1d of every MLP scaled by 0.02, final_proj orthogonal and
    the scale heads' stencils small: the residual stacks then perturb the synthetic descriptors instead of scrambling them, so the
    optimal-transport problems behind them keep the headline's match structure and the step's counts (rows, P, M) stay comparable -
    the arithmetic per layer does not depend on the values.  Every launch that runs over a capacity takes its count from the device
    (rows: chunk_base[-1]; third-level problems: P)."""

    def __init__(self, base, ops, dev, h, w):
        from pats_amd import heads
        self.base, self.ops, self.h, self.w = base, ops, h, w
        self.lefts, self.rights = base.lefts, base.rights
        g = torch.Generator(device=dev)
        g.manual_seed(99)

        def gnn(C, layers, seed):
            out = []
            for i in range(layers):
                p = synth.gnn_params(seed=seed + i, C=C)
                p["mlp.3.weight"] = (0.02 * p["mlp.3.weight"]).astype(np.float32)
                out.append(ops.PropagationParams(p, device=dev))
            return out

        def kenc(dim, seed):
            p = synth.kenc_params(seed=seed, feature_dim=dim)
            last = max(int(k.split(".")[1]) for k in p if k.endswith(".weight") and p[k].ndim == 3)
            p["encoder.%d.weight" % last] = (0.02 * p["encoder.%d.weight" % last]).astype(np.float32)
            return ops.MLPParams(p, device=dev, prefix="encoder.")

        def ortho(C):
            q, _ = torch.linalg.qr(torch.randn((C, C), device=dev, generator=g))
            return q.contiguous().reshape(C, C, 1), torch.zeros((C,), device=dev)

        def stencil(C):
            return (0.002 * torch.randn((1, C, 3, 3), device=dev, generator=g)).contiguous(), torch.zeros((1,), device=dev)
        self.names18, self.names10 = ["self", "cross"] * 9, ["self", "cross"] * 5
        self.coarse_heads = heads.CoarseHeads(kenc(448, 501), gnn(448, 18, 510), self.names18, ortho(448), stencil(448), bin_score=0.0)
        self.gnn2, self.proj2 = gnn(264, 18, 540), ortho(264)
        self.sx2, self.sy2 = stencil(264), stencil(264)
        self.kenc3, self.gnn3, self.scale3 = kenc(128, 502), gnn(128, 10, 570), stencil(128)
        R, Pc = base.cap.rows_cap, base.cap.P_cap
        # outputs of the stacks over the capacities, resident (rows past the device-side counts are never written: zeros)
        self.g2 = (torch.zeros((R, 264, 145), device=dev), torch.zeros((R, 264, 145), device=dev))
        self.g3 = (torch.zeros((Pc, 128, 65), device=dev), torch.zeros((Pc, 128, 65), device=dev))
        for t in base.desc + base.t0 + base.t1:
            t.zero_()                                       # the gathers' padding rows: zeros, not whatever torch.empty left
        self.ev = None

    def _timed(self, tag):
        if self.ev is None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.ev.setdefault(tag, []).append((e0, e1))
        e0.record()
        return e1

    def coarse(self, lefts, rights):
        b = self.base
        e = self._timed("coarse_heads")
        pairs = b.d0.shape[0]
        out = self.coarse_heads(b.d0.reshape(pairs, 448, self.h, self.w), b.d1.reshape(pairs, 448, self.h, self.w))
        if e is not None:
            e.record()
        return out

    def fine(self, rows, new_left, new_right):
        b, ops = self.base, self.ops
        live = rows.chunk_base[-1:]
        desc = b.fine(rows, new_left, new_right)                                                     # a15 (counted)
        e = self._timed("fine_gnn")
        d0, d1 = ops.attentional_gnn(desc[0], desc[1], self.gnn2, self.names18, count=live, out=self.g2)     # second_layer.py:89
        if e is not None:
            e.record()
        e = self._timed("fine_proj_scale")
        m0, m1 = ops.conv1d(d0, *self.proj2), ops.conv1d(d1, *self.proj2)                             # :91
        _, (sx, sy) = ops.scale_head(m1, 12, 12, [self.sx2[0], self.sy2[0]], [self.sx2[1], self.sy2[1]], return_heads=True)   # :92-97
        if e is not None:
            e.record()
        return m0, m1, sx.contiguous(), sy.contiguous()

    def third(self, rows, mk0, mk1, b_ids, P_dev):
        b, ops = self.base, self.ops
        from pats_amd import heads
        k3 = ops.keypoint_encoder(heads.grid_kpts(8, 8, mk0.device), self.kenc3)                    # third_layer.py:132-140
        kk = b.third_calls & 1
        b.third_calls += 1
        t0, t1, ps, pt = ops.third_descriptors(b.ff0, b.ff1, mk0, mk1, b_ids, k3.reshape(128, 64), b.rubbish3, count=P_dev,
                                               out=(b.t0[kk], b.t1[kk]))                             # a16
        e = self._timed("third_gnn")
        f0, f1 = ops.attentional_gnn(t0, t1, self.gnn3, self.names10, count=P_dev, out=self.g3)      # :146-148
        if e is not None:
            e.record()
        scale = ops.scale_head(f1, 8, 8, [self.scale3[0]], [self.scale3[1]])                         # :151-152
        return f0, f1, scale, ps, pt


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif hasattr(obj, "__slots__"):
        for k in obj.__slots__:
            yield from _tensors(getattr(obj, k, None))


class StepWatch:
    """Every step's counters (table status, third-level problem count P, match count M) come back to the host INSIDE the timed
    region - asynchronously into pinned memory, checked one step behind, so the steps still queue ahead of the GPU - and a
    capacity overflow in ANY step raises (batch.split_by_pair checks only the step it is handed)."""

    def __init__(self, cap, depth=2):
        self.cap, self.q, self.depth = cap, [], depth
        self.pool = [torch.empty(cap.pairs + 4, dtype=torch.int64).pin_memory() for _ in range(depth + 1)]
        self.steps = 0                                      # (one plain D2H copy a step: no kernel outside pats:: enters the steps)

    def push(self, out):
        buf = self.pool[self.steps % len(self.pool)]
        buf.copy_(out["summary"], non_blocking=True)        # batch.group_by_pair: the pairs + 1 offsets, then M, P, table status
        e = torch.cuda.Event()
        e.record()
        self.q.append((e, buf))
        self.steps += 1
        while len(self.q) > self.depth:
            self._check(*self.q.pop(0))

    def _check(self, e, buf):
        e.synchronize()
        v = buf.tolist()
        off, (M, P, status) = v[:self.cap.pairs + 1], v[self.cap.pairs + 1:]
        if status or P > self.cap.P_cap:
            raise RuntimeError("bench: a step overflowed a capacity (status %d, P %d of %d)" % (status, P, self.cap.P_cap))
        if off[-1] not in (0, M) or any(b_ < a_ for a_, b_ in zip(off, off[1:])):
            raise RuntimeError("bench: the per-pair offsets of a step do not add up to its match count")
        self.last = (status, P, M)

    def drain(self):
        while self.q:
            self._check(*self.q.pop(0))


def run_steps(batch, nets, cap, wl, ev, n, streams, watch=None):
    """n complete steps (batches).  streams = None: the stages of a batch one after the other on the current stream.
    streams = (sG, sS): two HIP streams with DISJOINT compute-unit masks (ops.masked_stream) -
        sG  the HBM-bound stages: coarse level + chunk rows + crops + fine descriptor gather of batch i, third-level window
            gather of batch i - 1
        sS  the VALU-bound stages: fine cost + OT + expansion + merges of batch i, third-level OT + results of batch i - 1
    Consecutive batches are independent (pairs are), so the memory-bound gathers of one batch run beside the solvers of its
    neighbour on different CUs (plain streams only time-slice: every kernel of the path fills all CUs' registers on its own).
    Every batch still goes through every kernel inside the timed region; nothing leaves the function unfinished (the
    caller's stream waits for both).  The gather outputs are double-buffered (BenchNets)."""
    kw = dict(if_outdoor=wl["outdoor"], iters=ITERS)
    if n <= 0:
        return None
    if streams is None:
        out = None
        for _ in range(n):
            co = batch.coarse_stage(nets.lefts, nets.rights, nets, cap, ITERS)
            fs = batch.fine_stage(co, nets, cap, merge_new=wl["merge_new"], events=ev, **kw)
            out = batch.third_stage(fs, nets, cap, events=ev, **kw)
            batch.group_by_pair(out, cap)                 # the hand-over: every pair's match list contiguous, offsets on the device
            if watch is not None:
                watch.push(out)
        if watch is not None:
            watch.drain()
        return out
    sG, sS = streams
    cur = torch.cuda.current_stream()
    sG.wait_stream(cur)
    sS.wait_stream(cur)

    def hand_over(obj, to):
        for t in _tensors(obj):                          # allocated on one stream, read on the other
            t.record_stream(to)

    def mark(stream):
        e = torch.cuda.Event()
        e.record(stream)
        return e
    co, fs, eC, eFS, eG, eT, out = {}, {}, {}, {}, {}, {}, None
    for i in range(n + 1):
        j = i - 1
        with torch.cuda.stream(sG):
            if i < n:                                    # (the fine-descriptor buffer of batch i - 2 is free: sG already waited
                co[i] = batch.coarse_stage(nets.lefts, nets.rights, nets, cap, ITERS)    # for eFS[i - 2] one tick ago)
                eC[i] = mark(sG)
            if 0 <= j < n:
                sG.wait_event(eFS[j])                    # the points of batch j exist
                if j - 2 in eT:
                    sG.wait_event(eT[j - 2])             # the third-level descriptor buffer of batch j - 2 has been read
                hand_over(fs[j], sG)
                batch.third_gather_stage(fs[j], nets, cap)
                eG[j] = mark(sG)
        with torch.cuda.stream(sS):
            if i < n:
                sS.wait_event(eC[i])
                hand_over(co[i], sS)
                fs[i] = batch.fine_solve_stage(co[i], nets, cap, merge_new=wl["merge_new"], events=ev, **kw)
                eFS[i] = mark(sS)
            if 0 <= j < n:
                sS.wait_event(eG[j])
                hand_over(fs[j], sS)
                out = batch.third_stage(fs[j], nets, cap, events=ev, **kw)
                eT[j] = mark(sS)
                co.pop(j, None)
                if j - 1 in fs:
                    fs.pop(j - 1)
    cur.wait_stream(sG)
    cur.wait_stream(sS)
    hand_over(out, cur)
    return out


def with_gnn_leg(ops, batch, dev, base, cap, wl, h, w, steps, warm=1):
    """`steps` whole steps with the heads inside (GnnNets), timed like the headline's: barrier, wall clock, markers for a kernel trace."""
    nets = GnnNets(base, ops, dev, h, w)
    # the layers' overflow protocol in its deferred form (ops.set_gnn_redo): no gated fp32 redo chain behind the fast kernels (~340
    # empty launches per pair, 21 ms of a 48-pair step in round 5) - the device's sticky flag is read HERE, after the steps, and a
    # raised flag repeats the leg under the inline protocol
    prev_mode = ops.set_gnn_redo(os.environ.get("PATS_BENCH_GNN_REDO", "deferred"))
    ops.gnn_overflows(reset=True)
    try:
        for attempt in range(2):
            run_steps(batch, nets, cap, wl, None, warm, None)
            torch.cuda.synchronize()
            ev = {}
            nets.ev = ev
            watch = StepWatch(cap)
            torch.cuda.synchronize()
            ops.profile_marker(1)
            t0 = time.perf_counter()
            out = run_steps(batch, nets, cap, wl, None, steps, None, watch)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ops.profile_marker(2)
            nets.ev = None
            overflow = ops.gnn_overflows(reset=True)
            if not overflow:
                break
            ops.set_gnn_redo("inline")                   # an activation left the fp16 range: the results above are void
    finally:
        redo_mode = ops.set_gnn_redo(prev_mode)
    mean = lambda tag: float(np.mean([a.elapsed_time(b_) for a, b_ in ev[tag]])) if tag in ev else None
    rep = {"pairs_per_s_with_gnn_measured": cap.pairs * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "ms_in_step": {"coarse_heads (KeypointEncoder + 18 layers + final_proj + scale head)": mean("coarse_heads"),
                          "fine_gnn (18 layers, both descriptor sets, every row in use)": mean("fine_gnn"),
                          "fine_proj_scale (final_proj x 2 + two scale heads)": mean("fine_proj_scale"),
                          "third_gnn (10 layers, both sets, every problem in use)": mean("third_gnn")},
           "rows_in_use": int(out["rows"].chunk_base[-1].item()), "third_problems": int(out["P"].item()), "matches": int(out["M"].item()),
           "gnn_redo": "%s (overflow flag read after the timed steps: %s)" % (redo_mode, "raised - repeated inline" if attempt else "not raised"),
           "note": "the headline step with the layers' heads as callbacks of batch.forward_pairs (bench.py::GnnNets): random weights, "
                   "backbones synthetic and resident; launches over capacities take their counts from the device"}
    del nets
    torch.cuda.empty_cache()
    return rep
