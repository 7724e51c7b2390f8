"""Time compile-time variants of the fused third-level kernel (PATS_TF_VAR) in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pats_amd import ops
P = 103680
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
d0, d1 = bench.desc_pair((P, 128, 65), dev, g)
sc = bench.scale_head((P, 1, 64), dev, g)
ps = torch.randint(1, 23, (P, 2), device=dev) * 4
pt = torch.randint(0, 25, (P, 2), device=dev) * 4
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for var in (0, 1, 2, 3, 0, 1, 2, 3):
    os.environ["PATS_TF_VAR"] = str(var)
    f = lambda it: timeit(lambda: ops.third_level(d0, d1, sc, ps, pt, iters=it))
    print("var=%d : iters=0 %.3f  iters=100 %.3f  iters=200 %.3f ms" % (var, f(0), f(100), f(200)), flush=True)
print("v1 : iters=100 %.3f ms" % timeit(lambda: ops.third_level(d0, d1, sc, ps, pt, iters=100, return_plan=True)))
