#!/usr/bin/env python3
"""How do the path's two kinds of kernels scale with the number of compute units they may use?  An HBM-bound gather
(ops.fine_descriptors) and a VALU-bound solver (ops.third_level) on streams masked to every k-th CU, alone and together."""
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
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps

full = torch.cuda.Stream()
print("all 256 CUs: gather %.2f ms, solver %.2f ms" % (timed(gather, full), timed(solve, full)))
for num, den in ((1, 4), (1, 3), (1, 2), (2, 3), (3, 4)):
    cus_g = [c for c in range(256) if (c * num) % den < num] if False else [c for c in range(256) if (c % den) < num]
    cus_s = [c for c in range(256) if c not in set(cus_g)]
    sg, ss = ops.masked_stream(cus_g), ops.masked_stream(cus_s)
    tg, ts = timed(gather, sg), timed(solve, ss)
    # together: both streams busy at once
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        with torch.cuda.stream(sg): gather()
        with torch.cuda.stream(ss): solve()
    torch.cuda.synchronize()
    both = 1e3 * (time.perf_counter() - t0) / 6
    print("gather on %3d CUs %.2f ms | solver on %3d CUs %.2f ms | both at once %.2f ms per (gather + solve)" % (len(cus_g), tg, len(cus_s), ts, both))
