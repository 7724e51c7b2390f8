#!/usr/bin/env python3
"""One launch each of the third-level kernel on a fixed problem count, for rocprofv3 --pmc passes.
env: PMC_P (problems, default 414720 = the bench's launch), PMC_ITERS (sweeps, default 100)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth
P = int(os.environ.get("PMC_P", "414720"))
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.randn((P, 128, 65), device=dev, generator=g)
d0 = (3 * (base + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g))).contiguous()
d1 = (3 * (base + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g))).contiguous()
del base
sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device=dev, generator=g)) * synth.LN256 - synth.LN256 / 2)
ps = torch.randint(1, 23, (P, 2), device=dev) * 4
pt = torch.randint(0, 25, (P, 2), device=dev) * 4
torch.cuda.synchronize()
for _ in range(3):
    ops.third_level(d0, d1, sc, ps, pt, iters=int(os.environ.get("PMC_ITERS", "100")))
    if os.environ.get("PMC_CALIB"):
        ops.cost(d0[:25920], d1[:25920])
torch.cuda.synchronize()
print("done P=%d bytes_in_per_problem=%d" % (P, 2 * 128 * 65 * 4))
