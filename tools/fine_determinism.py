#!/usr/bin/env python3
"""Run-to-run determinism of the fine-level Sinkhorn kernel (sinkhorn_blk145_kernel) on fixed inputs (GPU box).
usage: python tools/fine_determinism.py [rows] [runs] [preheat 0|1]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pats_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
PREHEAT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = "cuda"
g = torch.Generator(device=dev)
g.manual_seed(3)
base = torch.randn((R, 264, 145), device=dev, generator=g)
d0 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
d1 = 3.0 * (base + 0.3 * torch.randn((R, 264, 145), device=dev, generator=g))
ns = torch.exp(0.3 * torch.randn((R, 1, 144), device=dev, generator=g))
S = ops.cost(d0, d1)
del d0, d1, base
torch.cuda.synchronize()
if PREHEAT:
    x = torch.randn((8192, 8192), device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.8:
        for _ in range(8):
            x = (x @ x) * 1e-4
        torch.cuda.synchronize()
outs = []
for r in range(RUNS):
    outs.append(ops.log_optimal_transport2(S, 1.0, ns, 100))
    torch.cuda.synchronize()
    if r == RUNS // 2:
        time.sleep(2.0)          # let the device idle once in the middle
print("mode %s, rows %d, fallbacks %d" % (os.environ.get("PATS_SINKHORN", "auto"), R, ops.sinkhorn_fallbacks(reset=True)))
# majority vote per problem is overkill: compare every run with the last one
ref = outs[-1]
sets = []
for r, Z in enumerate(outs):
    diff = (Z != ref) & ~(torch.isnan(Z) & torch.isnan(ref))
    idx = torch.nonzero(diff.flatten(1).any(1)).flatten().tolist()
    sets.append(set(idx))
    mx = float((Z - ref).abs().nan_to_num().max().item())
    print("run %2d vs last: %3d problems differ, max |dZ| %.3g  %s" % (r, len(idx), mx, idx[:8]))
allp = sorted(set().union(*sets))
print("problems ever differing: %d %s" % (len(allp), allp[:24]))
if allp:
    p = allp[0]
    vals = torch.stack([Z[p] for Z in outs])
    uniq = [int((vals[i] != vals[-1]).sum().item()) for i in range(RUNS)]
    print("problem %d: entries differing from the last run, per run: %s" % (p, uniq))
    print("  its plan: max Z %.3f, min Z %.3f, ns range %.3f..%.3f" % (float(ref[p].max()), float(ref[p].min()), float(ns[p].min()), float(ns[p].max())))
# anatomy of the differences: for a few (run, problem) pairs, is D = Z_run - Z_ref a constant, u_i + v_j, or one entry?
shown = 0
for r, Z in enumerate(outs[:-1]):
    for p in sorted(sets[r] - sets[-2 if r != RUNS - 2 else 0])[:2]:
        D = (Z[p] - ref[p]).double()
        fin = torch.isfinite(D)
        D = torch.where(fin, D, torch.zeros_like(D))
        u = D.mean(1, keepdim=True)
        v = (D - u).mean(0, keepdim=True)
        res = D - u - v
        i, j = divmod(int(res.abs().argmax()), 145)
        print("run %d problem %d: max|D| %.3g, after removing u_i + v_j: max %.3g at (%d, %d); D there %.3g; mean D %.3g; "
              "u range %.3g..%.3g, v range %.3g..%.3g" % (r, p, float(D.abs().max()), float(res.abs().max()), i, j, float(D[i, j]),
                                                           float(D.mean()), float(u.min()), float(u.max()), float(v.min()), float(v.max())))
        shown += 1
    if shown >= 8:
        break
