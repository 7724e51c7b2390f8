#!/usr/bin/env python3
"""Error-rate statistic of the fine-level solve for one library variant (PATS_AMD_DIAG_LIB=<suffix>): 100 sweeps, 16 launches
per marginal mode, problems differing from the per-problem MAJORITY result."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pats_amd import ops  # noqa: E402
R, N = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(3)
base = torch.randn((R, 264, 145), device=dev, generator=g)
d0 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
d1 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
ns = torch.exp(0.3 * torch.randn((R, 1, 144), device=dev, generator=g))
S = ops.cost(d0, d1)
del d0, d1, base
one = torch.ones(1, device=dev)
nsf = ns.reshape(R, 144)
norm = -torch.log(144.0 + nsf.sum(1, keepdim=True))
log_mu = torch.cat([norm.expand(R, 144), torch.log(nsf.sum(1, keepdim=True)) + norm], 1).contiguous()
log_nu = torch.cat([torch.log(nsf) + norm, torch.log(torch.full((R, 1), 144.0, device=dev)) + norm], 1).contiguous()

def stat(fn):
    # a checksum per problem and launch; the majority checksum of a problem is "the" result
    sums = []
    for _ in range(N):
        Z = fn()
        sums.append(Z.flatten(1).view(torch.int32).to(torch.int64).sum(1))
        del Z
    torch.cuda.synchronize()
    M = torch.stack(sums)                               # [N, R]
    maj = torch.mode(M, dim=0).values
    bad = (M != maj)
    return bad.sum(1).tolist(), int(bad.any(0).sum())

tag = os.environ.get("PATS_AMD_DIAG_LIB", "production")
for name, fn in (("ot2  ", lambda: ops.log_optimal_transport2(S, one, ns, 100)), ("mode0", lambda: ops.log_sinkhorn_iterations(S, log_mu, log_nu, 100))):
    per, probs = stat(fn)
    print("%-12s %s: wrong problems per launch %s  total %d over %d launches x %d problems" % (tag, name, per, sum(per), N, R))
