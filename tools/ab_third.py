"""A/B: block-layout fused third-level kernel (v2) vs sinkhorn65_kernel<2,1,1> (v1), same process."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth
P = 103680
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
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for it in (0, 100, 200):
    print("v2 iters=%3d: %.3f ms" % (it, timeit(lambda: ops.third_level(d0, d1, sc, ps, pt, iters=it))))
    print("v1 iters=%3d: %.3f ms" % (it, timeit(lambda: ops.third_level(d0, d1, sc, ps, pt, iters=it, return_plan=True))), "(also writes the plan)")
a = ops.third_level(d0[:4096], d1[:4096], sc[:4096], ps[:4096], pt[:4096])
b = ops.third_level(d0[:4096], d1[:4096], sc[:4096], ps[:4096], pt[:4096], return_plan=True)
print("v1 vs v2 max |d mkpts1| = %.3g ; flags equal: %s ; labels equal: %s" % ((a[1] - b[1]).abs().max().item(), torch.equal(a[3], b[3]), torch.equal(a[2], b[2])))
