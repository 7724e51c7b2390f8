#!/usr/bin/env python3
"""The register-resident sweep kernel of csrc/sinkhorn_stream.hip (config 5: 4097 x 4097, 200 sweeps) against the two-launch form:
run with PATS_AMD_DIAG_LIB=1 (the diagnostic library reads PATS_STREAM_RESIDENT); a child process solves the same problem with
PATS_STREAM_RESIDENT=0.  Prints the rates, the difference of the two plans and run-to-run identity."""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pats_amd import ops, synth


def solve():
    inp = synth.roofline_inputs()
    d0, d1, ns = (torch.from_numpy(inp[k]).cuda() for k in ("d0", "d1", "ns"))
    alpha = torch.tensor(float(inp["alpha"]), device="cuda")
    S = ops.cost(d0, d1)
    run = lambda: ops.log_optimal_transport(S, alpha, ns, 200)
    z0 = run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        z1 = run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 3
    return z0, z1, t


def main():
    if "--child" in sys.argv:
        z0, z1, t = solve()
        np.save(sys.argv[sys.argv.index("--child") + 1], z0.cpu().numpy())
        print(json.dumps({"ms": t, "sweeps_per_s": 200e3 / t}))
        return
    z0, z1, t = solve()
    res = {"resident": {"ms": round(t, 3), "sweeps_per_s": round(200e3 / t), "run_to_run_identical": bool(torch.equal(z0, z1)),
                        "finite": bool(torch.isfinite(z0).all())}}
    path = "/tmp/config5_two_launch.npy"
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=dict(os.environ, PATS_STREAM_RESIDENT="0"),
                         capture_output=True, text=True, timeout=600)
    res["two_launch"] = json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else out.stderr[-500:]
    if out.returncode == 0:
        ref = np.load(path)
        d = np.abs(z0.cpu().numpy() - ref)
        res["max_abs_diff_of_log_plans"] = float(d.max())
        res["max_abs_diff_of_masses"] = float(np.abs(np.exp(z0.cpu().numpy().astype(np.float64)) - np.exp(ref.astype(np.float64))).max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
