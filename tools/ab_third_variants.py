#!/usr/bin/env python3
"""A/B of the third-level kernel variants (PATS_THIRD_VARIANT digits: waves per SIMD, column reduction, dustbin
sums; PATS_THIRD_V2 = the second-generation kernel): one subprocess per variant, same data, kernel time at
100 and 200 sweeps (the difference isolates the sweep loop)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r'''
import sys, os, torch
sys.path.insert(0, os.path.dirname(%r))
from pats_amd import ops, synth
P = 414720
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.randn((P, 128, 65), device=dev, generator=g)
d0 = (3 * (base + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g))).contiguous()
d1 = (3 * (base + 0.3 * torch.randn((P, 128, 65), device=dev, generator=g))).contiguous()
del base
sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device=dev, generator=g)) * synth.LN256 - synth.LN256 / 2)
ps = torch.randint(1, 23, (P, 2), device=dev) * 4
pt = torch.randint(0, 25, (P, 2), device=dev) * 4
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t100 = timeit(lambda: ops.third_level(d0, d1, sc, ps, pt, iters=100))
t200 = timeit(lambda: ops.third_level(d0, d1, sc, ps, pt, iters=200))
print("%%s  100 sweeps %%.3f ms   200 sweeps %%.3f ms   per sweep %%.1f us   rest %%.2f ms" %% (os.environ.get("TAG"), t100, t200, (t200 - t100) * 10, 2 * t100 - t200))
''' % HERE
for tag, env in [("v2", {"PATS_THIRD_V2": "1"})] + [(v, {"PATS_THIRD_VARIANT": v, "PATS_AMD_DIAG_LIB": "1", "PATS_THIRD_ABLATION": "1"}) for v in sys.argv[1:]]:
    e = dict(os.environ, TAG=tag, **env)
    out = subprocess.run([sys.executable, "-c", CODE], env=e, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-500:])
