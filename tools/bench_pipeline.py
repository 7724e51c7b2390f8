#!/usr/bin/env python3
"""Latency mode: ONE 640x480 pair at a time through pats_amd.pipeline.forward_path (PATS.forward's control
flow: per-chunk launches, host reads for the chunk plan and the P / M counts), synthetic network outputs
generated on the GPU.  bench.py is the throughput mode (stages batched over pairs and chunks)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pats_amd import pipeline, synth

dev = torch.device("cuda")
gen = torch.Generator(device=dev); gen.manual_seed(7)
c = synth.coarse_inputs()
left, right = [torch.from_numpy(x).to(dev) for x in synth.image_pair()]
d0, d1, ns = [torch.from_numpy(c[k]).to(dev) for k in ("d0", "d1", "ns")]
POOL_B, POOL_P = 64, 64 * 60
f0, f1 = bench.desc_pair((POOL_B, 264, 145), dev, gen, drop=0.12)
f0[:, :, -1] *= 0.5; f1[:, :, -1] *= 0.5
sx, sy = bench.scale_head((POOL_B, 1, 144), dev, gen), bench.scale_head((POOL_B, 1, 144), dev, gen)
t0, t1 = bench.desc_pair((POOL_P, 128, 65), dev, gen, drop=0.12)
t0[:, :, -1] *= 0.5; t1[:, :, -1] *= 0.5
sc3 = bench.scale_head((POOL_P, 1, 64), dev, gen)

class Nets:
    def coarse(self, l, r): return d0, d1, ns, 0.0
    def fine(self, num, nl, nr, mask, sizes=None):
        idx = torch.arange(nl.shape[0], device=dev) % POOL_B
        return f0[idx], f1[idx], sx[idx], sy[idx]
    def third(self, num, mk0, mk1, b_ids, sizes=None):
        P = mk0.shape[0]
        idx = torch.arange(P, device=dev) % POOL_P
        return t0[idx], t1[idx], sc3[idx]

for batched in (False, True):
    run = lambda: pipeline.forward_path(left, right, Nets(), if_local=True, if_outdoor=True, merge_new=True, batch_chunks=batched)
    out = run(); torch.cuda.synchronize()
    n = 10
    t = time.perf_counter()
    for _ in range(n): out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(json.dumps({"mode": "latency: one pair at a time through pipeline.forward_path(batch_chunks=%s)" % batched,
                      "ms_per_pair": dt * 1e3, "pairs_per_s": 1.0 / dt, "chunks": len(out["chunks"]),
                      "B_total": sum(c[0] for c in out["chunks"])}))      # (the two modes draw different rows of the synthetic pool)
