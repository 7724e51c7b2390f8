#!/usr/bin/env python3
"""attention(query, key, value) (modules.py:84-88) at the path's three shapes: the fused HIP kernel against
the same expression in stock PyTorch on the same GPU.  Algorithmic bytes = q + k + v + out."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops

def ref_attention(q, k, v):
    dim = q.shape[1]
    scores = torch.einsum('bdhn,bdhm->bhnm', q, k) / dim ** .5
    prob = torch.nn.functional.softmax(scores, dim=-1)
    return torch.einsum('bhnm,bdhm->bdhn', prob, v)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

res = []
for name, b, dim, n in (("third level, one pair", 25920, 32, 65), ("fine level, one pair", 432, 66, 145), ("coarse level", 2, 112, 300)):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    q, k, v = (torch.randn((b, dim, 4, n), device="cuda", generator=g) for _ in range(3))
    t_hip = timeit(lambda: ops.attention(q, k, v, return_prob=False))
    t_ref = timeit(lambda: ref_attention(q, k, v))
    x, _ = ops.attention(q, k, v, return_prob=False)
    err = (x - ref_attention(q, k, v)).abs().max().item()
    byts = 4.0 * dim * 4 * n * 4 * b
    res.append({"shape": name, "batch": b, "dim": dim, "heads": 4, "tokens": n, "hip_ms": t_hip, "torch_ms": t_ref,
                "speedup": t_ref / t_hip, "algorithmic_GBps": byts / (t_hip * 1e-3) / 1e9, "max_abs_diff_vs_torch": err})
    print(json.dumps(res[-1]), flush=True)
