#!/usr/bin/env python3
"""BASELINE config 5: synthetic 4096x4096 patch cost matrix, 200 Sinkhorn sweeps (GPU only).
Prints the MFMA cost-GEMM rate and the sweep rate against the SURVEY 8d streaming model
(8*M*N algorithmic bytes per sweep)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    inp = synth.roofline_inputs()
    d0, d1, ns = (torch.from_numpy(inp[k]).cuda() for k in ("d0", "d1", "ns"))
    alpha = torch.tensor(float(inp["alpha"]), device="cuda")
    N, D = 4096, 448
    t_cost = timeit(lambda: ops.cost(d0, d1))
    S = ops.cost(d0, d1)
    res = {"cost_ms": t_cost, "cost_tflops": 2.0 * D * N * N / (t_cost * 1e-3) / 1e12}
    for mode in ("kernel", "log"):
        ops.set_sinkhorn_mode(mode)
        its = 200 if mode == "kernel" else 20
        t0 = timeit(lambda: ops.log_optimal_transport(S, alpha, ns, 0), 1) if mode == "kernel" else 0.0
        t = timeit(lambda: ops.log_optimal_transport(S, alpha, ns, its), 2 if mode == "kernel" else 1)
        per = (t - (t0 if mode == "kernel" and False else 0.0)) / its
        M = N + 1
        res[mode] = {"iters": its, "ms": t, "ms_per_sweep": per, "sweeps_per_s": 1e3 / per,
                     "algorithmic_GBps_8MN": 8.0 * M * M / (per * 1e-3) / 1e9}
    ops.set_sinkhorn_mode("kernel")
    # YFCC-sized coarse level (24x32 grid)
    c = synth.coarse_inputs(seed=synth.SEED + 21, h=24, w=32)
    e0, e1, ens = (torch.from_numpy(c[k]).cuda() for k in ("d0", "d1", "ns"))
    res["coarse_769_ms"] = timeit(lambda: ops.cost_ot(e0, e1, 1, 0.0, ens, 100))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
