#!/usr/bin/env python3
"""One launch each of the other kernels of the path for rocprofv3 --pmc passes: the cost GEMM at the config-5 and
fine-level shapes, the fine-level block Sinkhorn kernel, the streaming solver at 4097^2 (200 sweeps), the coarse one-CU
kernel (16 problems of 301^2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
r = synth.roofline_inputs()
d0, d1, ns = [torch.from_numpy(r[k]).to(dev) for k in ("d0", "d1", "ns")]
B = 6912
f0 = torch.randn((B, 264, 145), device=dev, generator=g); f1 = f0 + 0.3 * torch.randn((B, 264, 145), device=dev, generator=g)
ns2 = torch.rand((B, 144), device=dev, generator=g) + 0.5
c = synth.coarse_inputs()
c0, c1, cns = [torch.from_numpy(c[k]).to(dev).repeat(16, 1, 1).contiguous() for k in ("d0", "d1", "ns")]
torch.cuda.synchronize()
for _ in range(2):
    S = ops.cost(d0, d1)                                        # cost_mfma_kernel, 4096^2 x 448
    ops.log_optimal_transport(S, 1.0, ns, 200)                  # stream kernels, 4097^2
    S2 = ops.cost(f0, f1)                                       # cost_mfma_kernel, 6912 x [264,145]^2
    ops.log_optimal_transport2(S2, 1.0, ns2, 100, 2.0)          # sinkhorn_blk145_kernel
    ops.cost_ot(c0, c1, 1, 0.0, cns, 100)                       # sinkhorn_cu_kernel
torch.cuda.synchronize()
print("done")
