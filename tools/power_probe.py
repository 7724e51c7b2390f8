#!/usr/bin/env python3
"""Clock and power while the third-level kernel runs back to back (rocm-smi sampled from a thread)."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout
        samples.append(out.strip()[:600])
        time.sleep(0.3)
iters = int(os.environ.get("ITERS", "200"))
ops.third_level(d0, d1, sc, ps, pt, iters=iters); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter()
n = 120
for _ in range(n):
    ops.third_level(d0, d1, sc, ps, pt, iters=iters)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
stop[0] = True; th.join()
print("TAG=%s iters=%d  %.3f ms per launch" % (os.environ.get("TAG"), iters, dt * 1e3))
for s in samples[1:6]:
    print(s)
