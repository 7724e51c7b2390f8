#!/usr/bin/env python3
"""Fine level: fused cost -> OT kernel against cost_mfma_kernel + sinkhorn_blk145_kernel (GPU box)."""
import os, sys, time, subprocess
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pats_amd import ops  # noqa: E402
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20224
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(3)
d = torch.empty((2, R, 264, 145), device=dev)
for r0 in range(0, R, 2048):
    n = min(2048, R - r0)
    base = torch.randn((n, 264, 145), device=dev, generator=g)
    d[0, r0:r0 + n] = 3.0 * (base + 0.3 * torch.randn((n, 264, 145), device=dev, generator=g))
    d[1, r0:r0 + n] = 3.0 * (base + 0.3 * torch.randn((n, 264, 145), device=dev, generator=g))
ns = torch.exp(0.3 * torch.randn((R, 1, 144), device=dev, generator=g))
one = torch.ones(1, device=dev)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
f = lambda: ops.cost_ot(d[0], d[1], 2, one, ns, 100, bias_k=2.0, return_flags=True)
ms = timed(f)
tag = "fused" if os.environ.get("PATS_FINE_FUSED") else "two kernels"
print("%s: %d rows, cost + OT %.3f ms, fallbacks %d" % (tag, R, ms, ops.sinkhorn_fallbacks(reset=True)))
