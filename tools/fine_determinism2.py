#!/usr/bin/env python3
"""Does the run-to-run difference of the fine-level Sinkhorn follow the device's power state?  (GPU box)
A: 10 launches queued back to back; B: a synchronize + 30 ms of idle before every launch; C: A again after B."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pats_amd import ops  # noqa: E402
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(3)
base = torch.randn((R, 264, 145), device=dev, generator=g)
d0 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
d1 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
ns = torch.exp(0.3 * torch.randn((R, 1, 144), device=dev, generator=g))
S = ops.cost(d0, d1)
del d0, d1, base
one = torch.ones(1, device=dev)
torch.cuda.synchronize()

def count(outs, ref):
    return [int(((Z != ref) & ~(torch.isnan(Z) & torch.isnan(ref))).flatten(1).any(1).sum().item()) for Z in outs]

def series(idle, n=10):
    outs = []
    for _ in range(n):
        if idle:
            torch.cuda.synchronize(); time.sleep(idle)
        outs.append(ops.log_optimal_transport2(S, one, ns, 100))
    torch.cuda.synchronize()
    return outs

# reference: the majority value per problem over a long back-to-back series (drop its head)
warm = series(0.0, 14)
ref = warm[-1]
print("A  back to back (14):        problems differing from the last launch:", count(warm, ref))
b = series(0.03, 10)
print("B  30 ms idle before each:   ", count(b, ref))
c = series(0.0, 10)
print("C  back to back again:       ", count(c, ref))
d = series(0.002, 10)
print("D  sync + 2 ms before each:  ", count(d, ref))
e = series(1e-9, 10)
print("E  sync only before each:    ", count(e, ref))
