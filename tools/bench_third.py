#!/usr/bin/env python3
"""Phase timing of the fused third-level kernel: time vs. sweep count (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 103680
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.randn((P, 128, 65), device=dev, generator=g)
d0 = (3 * (base + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g))).contiguous()
d1 = (3 * (base + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g))).contiguous()
del base
sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device=dev, generator=g)) * synth.LN256 - synth.LN256 / 2)
ps = torch.randint(1, 23, (P, 2), device=dev) * 4
pt = torch.randint(0, 25, (P, 2), device=dev) * 4


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mode in ("kernel", "log"):
    ops.set_sinkhorn_mode(mode)
    for it in (0, 50, 100, 200):
        t = timeit(lambda: ops.third_level(d0, d1, sc, ps, pt, iters=it))
        print("mode=%s iters=%3d  %.3f ms  (%.1f ns/problem)" % (mode, it, t, 1e6 * t / P))
ops.set_sinkhorn_mode("kernel")
S = ops.cost(d0, d1)
print("cost65 alone            %.3f ms" % timeit(lambda: ops.cost(d0, d1)))
print("OT2 alone (100)         %.3f ms" % timeit(lambda: ops.log_optimal_transport2(S, 1.0, sc, 100)))
Z = ops.log_optimal_transport2(S, 1.0, sc, 100)
sxy = torch.sqrt(sc + 1e-8)
print("Compute_result alone    %.3f ms" % timeit(lambda: ops.Compute_result(Z, 8, 5, sxy, sxy, ps, pt, input_is_log=True)))
B = 1728
f0 = torch.randn((B, 264, 145), device=dev); f1 = torch.randn((B, 264, 145), device=dev)
ns = torch.rand((B, 1, 144), device=dev) + 0.5
print("L2 cost (B=%d)        %.3f ms" % (B, timeit(lambda: ops.cost(f0, f1))))
S2 = ops.cost(f0, f1)
for it in (0, 100):
    print("L2 OT2 iters=%3d        %.3f ms" % (it, timeit(lambda: ops.log_optimal_transport2(S2, 1.0, ns, it, 2.0))))
Z2 = ops.log_optimal_transport2(S2, 1.0, ns, 100, 2.0)
print("L2 est_position         %.3f ms" % timeit(lambda: ops.est_position_second(Z2, ns, ns, [96, 96], 8)))
