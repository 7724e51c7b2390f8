#!/usr/bin/env python3
"""The third level's fused AttentionalPropagation layer (gnn_fused.hip, 25 920 x [128, 65]) on random against all-zero operands:
the same instruction stream, what the power budget (DVFS) costs it - the fine level's tile kernel loses 21 % to it
(profiles/r05_gnn_fine_power_zeros_ab.txt)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pats_amd import ops, synth


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


b, C, n = 25920, 128, 65
res = {}
for zeros in (False, True, False, True):
    q = synth.gnn_params(seed=3, C=C)
    if zeros:
        q = {k: (np.ones_like(v) if k.endswith("running_var") else np.zeros_like(v)) for k, v in q.items()}
    P = ops.PropagationParams(q)
    x = torch.zeros((b, C, n), device="cuda") if zeros else torch.randn((b, C, n), device="cuda")
    s = torch.zeros((b, C, n), device="cuda") if zeros else torch.randn((b, C, n), device="cuda")
    t = timeit(lambda: ops.attentional_propagation(x, s, P, residual=x))
    res.setdefault("zeros" if zeros else "random", []).append(round(t, 3))
print(json.dumps({"ms_per_25920_problems": res}))
