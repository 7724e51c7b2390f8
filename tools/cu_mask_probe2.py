#!/usr/bin/env python3
"""Which CU masks does the runtime honour, and how does a VALU-bound kernel's time scale with them?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
P = 110592
d0 = torch.randn((P, 128, 65), device=dev, generator=g); d1 = d0 + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g)
sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device=dev, generator=g)) * synth.LN256 - synth.LN256 / 2)
ps = torch.randint(1, 23, (P, 2), device=dev) * 4; pt = torch.randint(0, 25, (P, 2), device=dev) * 4
def solve(): ops.third_level(d0, d1, sc, ps, pt)
def timed(fn, stream, reps=5):
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps
base = timed(solve, torch.cuda.Stream())
print("all CUs: %.2f ms" % base)
pats = {"first 64": range(64), "first 128": range(128), "first 192": range(192), "c%8==0": [c for c in range(256) if c % 8 == 0],
        "c%8<4": [c for c in range(256) if c % 8 < 4], "c%16<8": [c for c in range(256) if c % 16 < 8], "c%32<16": [c for c in range(256) if c % 32 < 16],
        "c%64<32": [c for c in range(256) if c % 64 < 32], "c%3==0": [c for c in range(256) if c % 3 == 0], "c%5<2": [c for c in range(256) if c % 5 < 2],
        "c%2==0": [c for c in range(256) if c % 2 == 0], "c%4<3": [c for c in range(256) if c % 4 < 3], "odd 2 of 3": [c for c in range(256) if c % 3 != 0],
        "c%10<7": [c for c in range(256) if c % 10 < 7], "c%7<5": [c for c in range(256) if c % 7 < 5]}
for name, cus in pats.items():
    cus = list(cus)
    t = timed(solve, ops.masked_stream(cus))
    print("%-12s %3d CUs: %.2f ms  -> behaves like %.0f CUs" % (name, len(cus), t, 256 * base / t))
