#!/usr/bin/env python3
"""Accuracy of the 65x65 cost contraction against float64: fp32 MFMA (bitwise a k-ordered fma chain) vs the fp16-split
three-product MFMA of the fused third-level kernel (PATS_COST65_F16=1 routes pats_cost_f32 through it)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(%r))
from pats_amd import ops, synth
inp = synth.third_inputs(seed=5, P=512)
d0, d1 = inp["d0"], inp["d1"]
truth = np.einsum("bdn,bdm->bnm", d0.astype(np.float64), d1.astype(np.float64)) / np.sqrt(128.0) * 0.1
S = ops.cost(torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()).cpu().numpy()
e = np.abs(S - truth)
print("%%-8s max |dS| %%.3e  mean %%.3e  p99.9 %%.3e   (core 64x64: max %%.3e)" %% (os.environ.get("TAG"), e.max(), e.mean(), np.quantile(e, 0.999), e[:, :64, :64].max()))
''' % HERE
for tag, env in (("fp32", {}), ("f16x2", {"PATS_COST65_F16": "1"})):
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, TAG=tag, **env), capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-800:])
