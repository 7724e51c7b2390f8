#!/usr/bin/env python3
"""Where does the run-to-run difference of the fine-level solve come from?  Sweep count and marginal mode (GPU box)."""
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

def probs(outs):
    ref = outs[-1]
    return [int(((Z != ref) & ~(torch.isnan(Z) & torch.isnan(ref))).flatten(1).any(1).sum().item()) for Z in outs[:-1]]

for iters in (0, 1, 3, 10, 30, 100):
    outs = [ops.log_optimal_transport2(S, one, ns, iters) for _ in range(6)]
    torch.cuda.synchronize()
    print("ot2   iters %3d: %s" % (iters, probs(outs)))
# marginals handed in (MODE 0): log_mu / log_nu of modules.py:169-179 computed by torch
nsf = ns.reshape(R, 144)
ms = 144.0
norm = -torch.log(ms + nsf.sum(1, keepdim=True))
log_mu = torch.cat([norm.expand(R, 144), torch.log(nsf.sum(1, keepdim=True)) + norm], 1).contiguous()
log_nu = torch.cat([torch.log(nsf) + norm, torch.log(torch.full((R, 1), ms, device=dev)) + norm], 1).contiguous()
for iters in (1, 10, 100):
    outs = [ops.log_sinkhorn_iterations(S, log_mu, log_nu, iters) for _ in range(6)]
    torch.cuda.synchronize()
    print("mode0 iters %3d: %s" % (iters, probs(outs)))
