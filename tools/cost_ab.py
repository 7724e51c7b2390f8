#!/usr/bin/env python3
"""General cost kernel (csrc/cost.hip): fp16-split three-pass MFMA path (default) against the fp32 MFMA path
(PATS_COST_F32=1) - accuracy against float64 and time, at the coarse / fine / config-5 shapes and on ragged ones;
plus the in-kernel range fallback (|x| > 1023)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(%r))
from pats_amd import ops
tag = os.environ.get("TAG")
rng = np.random.default_rng(3)
def check(b, D, n, m, scale=1.0, spike=None):
    d0 = (rng.standard_normal((b, D, n)) * scale).astype(np.float32)
    d1 = (rng.standard_normal((b, D, m)) * scale).astype(np.float32)
    if spike is not None:
        d0[0, D // 2, n // 3] = spike
    truth = np.einsum("bdn,bdm->bnm", d0.astype(np.float64), d1.astype(np.float64)) / np.sqrt(float(D)) * 0.1
    S = ops.cost(torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()).cpu().numpy()
    e = np.abs(S - truth)
    print("%%-6s b=%%d D=%%d %%dx%%d scale %%g spike %%s: max |dS| %%.3e mean %%.3e  (max |S| %%.3g)" %% (tag, b, D, n, m, scale, spike, e.max(), e.mean(), np.abs(truth).max()), flush=True)
check(2, 448, 300, 300)
check(5, 264, 145, 145)
check(3, 128, 40, 77)
check(2, 100, 161, 33)
check(2, 7, 16, 500)
check(2, 264, 145, 145, scale=0.01)
check(2, 264, 145, 145, scale=30.0)
check(2, 264, 145, 145, spike=5000.0)
check(2, 264, 145, 145, spike=float("inf"))
def timeit(b, D, n, m, reps=10):
    d0 = torch.randn(b, D, n, device="cuda"); d1 = torch.randn(b, D, m, device="cuda")
    out = ops.cost(d0, d1)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): ops.cost(d0, d1)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    fl = 2.0 * b * D * n * m
    by = 4.0 * b * (D * (n + m) + n * m)
    print("%%-6s time b=%%d D=%%d %%dx%%d: %%.3f ms  %%.1f TFLOP/s  %%.0f GB/s" %% (tag, b, D, n, m, ms, fl / ms / 1e9, by / ms / 1e6), flush=True)
timeit(1, 448, 4096, 4096, 20)
timeit(20736, 264, 145, 145, 5)
timeit(48, 448, 300, 300, 20)
''' % HERE
for tag, env in (("fp32", {"PATS_COST_F32": "1"}), ("split", {})):
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, TAG=tag, **env), capture_output=True, text=True)
    print(out.stdout.strip())
    if out.returncode:
        print(out.stderr[-1500:])
