#!/usr/bin/env python3
"""What would the fine-level sweeps cost at the occupancy a fused cost -> OT kernel could have?  (GPU box, diagnostic library.)

sinkhorn_blk145_kernel runs three workgroups per CU (149 VGPRs, 13.7 KB LDS); the cost build it would have to absorb needs
41 KB of LDS staging and 112 accumulator + ~60 staging registers per lane (cost_mfma_kernel<true>: 256 VGPRs at two
workgroups per CU).  PATS_BLK_LDS_PAD (libpats_amd_diag.so) adds untouched dynamic LDS to the Sinkhorn launch, which lowers
its occupancy without changing a single instruction: 41216 B -> two workgroups per CU, 70000 B -> one.
usage: PATS_AMD_DIAG_LIB=1 python tools/fine_fusion_probe.py [rows]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("PATS_AMD_DIAG_LIB", "1")
from pats_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda"
g = torch.Generator(device=dev)
g.manual_seed(3)
base = torch.randn((R, 264, 145), device=dev, generator=g)
d0 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
d1 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
ns = torch.exp(0.3 * torch.randn((R, 1, 144), device=dev, generator=g))


def timed(fn, n=4):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


S = ops.cost(d0, d1)
cost_ms = timed(lambda: ops.cost(d0, d1))
print("rows %d: cost build alone %.3f ms" % (R, cost_ms))
ref = None
for pad in (0, 41216, 41216, 70000, 0):
    os.environ["PATS_BLK_LDS_PAD"] = str(pad)
    Z = ops.log_optimal_transport2(S, 1.0, ns, 100)
    ms = timed(lambda: ops.log_optimal_transport2(S, 1.0, ns, 100))
    if ref is None:
        ref, ms0 = Z, ms
    diff = (Z != ref) & ~(torch.isnan(Z) & torch.isnan(ref))
    nprob = int(diff.flatten(1).any(1).sum().item())
    print("  LDS pad %6d B: 100 sweeps %.3f ms  (x%.3f)  problems differing from the first run: %d of %d, max |diff| %.3g, fallbacks %d"
          % (pad, ms, ms / ms0, nprob, R, float((Z - ref).abs().nan_to_num().max().item()), ops.sinkhorn_fallbacks(reset=True)))
