#!/usr/bin/env python3
"""Host cost per op wrapper (enqueue only): what makes the chunk walk of pipeline.forward_chunks_device host-bound."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, pipeline, _lib
dev = torch.device("cuda")
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
print("profile_marker (ctypes, 2 args)      %.1f us" % t(lambda: ops.profile_marker(0)))
print("torch.empty                          %.1f us" % t(lambda: torch.empty((54, 144), device=dev)))
B = 54
f0 = torch.randn((B, 264, 145), device=dev); f1 = torch.randn((B, 264, 145), device=dev)
ns = torch.rand((B, 1, 144), device=dev) + 0.5
one = torch.tensor(1.0, device=dev)
print("cost_ot(fine, flags)                 %.1f us" % t(lambda: ops.cost_ot(f0, f1, 2, one, ns, 100, bias_k=2.0, return_flags=True), 100))
Z, fl = ops.cost_ot(f0, f1, 2, one, ns, 100, bias_k=2.0, return_flags=True)
sx = torch.rand((B, 1, 144), device=dev) + 0.5
print("est_position_second                  %.1f us" % t(lambda: ops.est_position_second(Z, sx, sx, [96, 96], 8, col_nomatch=fl), 100))
tr, pts2, _, _, ifn, _ = ops.est_position_second(Z, sx, sx, [96, 96], 8, col_nomatch=fl)
print("third_inputs(sync=False)             %.1f us" % t(lambda: ops.third_inputs(ifn, pts2, capacity=B * 144, sync=False), 100))
mk0, mk1, bi, P = ops.third_inputs(ifn, pts2, capacity=B * 144, sync=False)
ff = torch.randn((B, 128, 52, 52), device=dev); kenc = torch.randn((128, 64), device=dev); rub = torch.randn((B, 128, 144), device=dev)
print("third_descriptors(count)             %.1f us" % t(lambda: ops.third_descriptors(ff, ff, mk0, mk1, bi, kenc, rub, count=P), 100))
t0_, t1_, ps, pt = ops.third_descriptors(ff, ff, mk0, mk1, bi, kenc, rub, count=P)
sc3 = torch.rand((B * 144, 1, 64), device=dev) + 0.5
print("third_level(count)                   %.1f us" % t(lambda: ops.third_level(t0_, t1_, sc3, ps, pt, outdoor=True, count=P), 100))
m0, m1, lab, ifm = ops.third_level(t0_, t1_, sc3, ps, pt, outdoor=True, count=P)
print("refine_scatter                       %.1f us" % t(lambda: ops.refine_scatter(ifn, pts2, m1, lab), 100))
