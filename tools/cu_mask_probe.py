#!/usr/bin/env python3
"""Gather (HBM-bound) and solver (VALU-bound) under complementary CU masks, alone and together."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
R, P = 8192, 110592
m0 = torch.randn((2 * R, 64, 48, 48), device=dev, generator=g); m1 = torch.randn((2 * R, 64, 24, 24), device=dev, generator=g)
m2 = torch.randn((2 * R, 128, 12, 12), device=dev, generator=g)
title = torch.randn((R, 8), device=dev); rub = torch.randn((R, 264), device=dev)
desc = torch.empty((2, R, 264, 145), device=dev)
d0 = torch.randn((P, 128, 65), device=dev, generator=g); d1 = d0 + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g)
sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device=dev, generator=g)) * synth.LN256 - synth.LN256 / 2)
ps = torch.randint(1, 23, (P, 2), device=dev) * 4; pt = torch.randint(0, 25, (P, 2), device=dev) * 4
def gather(): ops.fine_descriptors([m0, m1, m2], title, rub, out=desc)
def solve(): ops.third_level(d0, d1, sc, ps, pt)
def timed(fn, stream, reps=6):
    with torch.cuda.stream(stream):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps
full = torch.cuda.Stream()
timed(solve, full)
tg0, ts0 = timed(gather, full), timed(solve, full)
print("all CUs: gather %.2f ms, solver %.2f ms, sum %.2f" % (tg0, ts0, tg0 + ts0))
for name, sel in (("k<2", lambda c: c // 32 < 2), ("k<3", lambda c: c // 32 < 3), ("k<4", lambda c: c // 32 < 4), ("k<5", lambda c: c // 32 < 5),
                  ("k in 0,2,4", lambda c: (c // 32) in (0, 2, 4)), ("k<3 (as G), swap", lambda c: c // 32 >= 5)):
    cg = [c for c in range(256) if sel(c)]; cs = [c for c in range(256) if not sel(c)]
    sg, ss = ops.masked_stream(cg), ops.masked_stream(cs)
    tg, ts = timed(gather, sg), timed(solve, ss)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        with torch.cuda.stream(sg): gather()
        with torch.cuda.stream(ss): solve()
    torch.cuda.synchronize()
    both = 1e3 * (time.perf_counter() - t0) / 6
    print("%-8s gather on %3d CUs %.2f ms (x%.2f) | solver on %3d CUs %.2f ms (x%.2f) | both at once %.2f ms (serial on all CUs: %.2f)"
          % (name, len(cg), tg, tg / tg0, len(cs), ts, ts / ts0, both, tg0 + ts0))
