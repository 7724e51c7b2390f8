#!/usr/bin/env python3
"""sinkhorn_cu2_kernel against sinkhorn_cu_kernel (diagnostic library: PATS_CU_V1=1 selects the first version): the coarse level's
log-plans of 48 + 1 problems at 301x301 and a ragged 250x290 case must be BIT-IDENTICAL (the packed row pass and the once-formed
column sums keep every operand and every summation order), and the time per solve is printed.  Run with PATS_AMD_DIAG_LIB=1; the
script re-runs itself as two child processes (the switch is read once per process)."""
import os, subprocess, sys, time
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, REPO)
    from pats_amd import ops
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    outs = {}
    for name, (b, M, N) in {"coarse48": (48, 300, 300), "one": (1, 300, 300), "ragged": (3, 250, 290)}.items():
        base = torch.randn((b, 448, max(M, N)), device="cuda", generator=g)
        d0 = 3.0 * (base[:, :, :M] + 0.3 * torch.randn((b, 448, M), device="cuda", generator=g))
        d1 = 3.0 * (base[:, :, :N] + 0.3 * torch.randn((b, 448, N), device="cuda", generator=g))
        ns = torch.exp(torch.randn((b, 1, N), device="cuda", generator=g) * 0.5)
        Z = ops.cost_ot(d0.contiguous(), d1.contiguous(), 1, 0.25, ns, 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            Z = ops.cost_ot(d0, d1, 1, 0.25, ns, 100)
        torch.cuda.synchronize()
        outs[name] = Z.cpu().numpy()
        print("%s %s: %.1f us per call (cost build + solve), fallbacks %d" % (os.environ.get("PATS_CU_V1", "0"), name, (time.perf_counter() - t0) / 20 * 1e6, ops.sinkhorn_fallbacks()))
    np.savez(sys.argv[1], **outs)
    sys.exit(0)
files = []
for v1 in ("1", "0"):
    f = "/tmp/cu2_ab_%s.npz" % v1
    env = dict(os.environ, PATS_AMD_DIAG_LIB="1", PATS_CU_V1=v1)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), f], env=env, capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    if r.returncode:
        sys.exit(r.stderr[-2000:])
    files.append(np.load(f))
for k in files[0].files:
    same = np.array_equal(files[0][k], files[1][k])
    print("%s: bit-identical %s (finite %s)" % (k, same, bool(np.isfinite(files[1][k]).all())))
    assert same
