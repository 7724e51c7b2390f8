import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pats_amd import ops
g = torch.Generator(device="cuda"); g.manual_seed(5)
S = 2.0 * torch.randn((8192, 145, 145), device="cuda", generator=g)
ns = torch.exp(0.3 * torch.randn((8192, 1, 144), device="cuda", generator=g))
for _ in range(3):
    Z = ops.log_optimal_transport2(S, 1.0, ns, 100)
torch.cuda.synchronize()
