#!/usr/bin/env python3
"""The resident streaming kernel's give-up path, on purpose (diagnostic library: PATS_AMD_DIAG_LIB=1 PATS_STREAM_RESIDENT=2): the
241-workgroup kernel launched on a stream masked to 160 CUs can never be co-resident - every workgroup must leave its bounded spins,
the guard flag must route the problem to the log-domain kernel, and the result must still be the solution."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth
inp = synth.roofline_inputs()
d0, d1, ns = (torch.from_numpy(inp[k]).cuda() for k in ("d0", "d1", "ns"))
alpha = torch.tensor(float(inp["alpha"]), device="cuda")
S = ops.cost(d0, d1)
ref = ops.log_optimal_transport(S, alpha, ns, 200); torch.cuda.synchronize()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
ms = ops.masked_stream([c for c in range(n_cu) if c // 32 < 5])
ms.wait_stream(torch.cuda.current_stream())
ops.sinkhorn_fallbacks(reset=True)
with torch.cuda.stream(ms):
    t0 = time.perf_counter(); z = ops.log_optimal_transport(S, alpha, ns, 200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("forced resident launch on 160 CUs: %.2f s, fallbacks %s, finite %s, max |d log-plan| against the full-GPU solve %.2e"
      % (dt, ops.sinkhorn_fallbacks(reset=True), bool(torch.isfinite(z).all()), (z - ref).abs().max().item()))
