#!/usr/bin/env python3
"""Development check of the fine level's one-kernel AttentionalPropagation (csrc/gnn_fine.hip): the layer at [264, 145] against
the oracle for a few problems and for more problems than workgroups (scratch blocks and the LDS slot are reused), against the
round-4 composition (PATS_GNN_FINE=0 in a child process), and its time per 4 096 problems.
    python tools/fine_layer_check.py [--time-only]"""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pats_amd import ops, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda")
    C, n = 264, 145
    params = synth.gnn_params(seed=7, C=C)
    P = ops.PropagationParams(params)
    if "--child" in sys.argv:                       # the round-4 composition on the same inputs -> a file
        i = synth.gnn_inputs(seed=8, b=700, C=C, n=n)
        y = ops.attentional_propagation(torch.from_numpy(i["x"]).to(dev), torch.from_numpy(i["source"]).to(dev), P,
                                        residual=torch.from_numpy(i["x"]).to(dev))
        np.save(sys.argv[sys.argv.index("--child") + 1], y.cpu().numpy())
        return
    if "--time-only" not in sys.argv:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import pats_oracle as oracle
        i = synth.gnn_inputs(seed=8, b=5, C=C, n=n)
        x, s = torch.from_numpy(i["x"]).to(dev), torch.from_numpy(i["source"]).to(dev)
        for res in (None, x):
            y = ops.attentional_propagation(x, s, P, residual=res).cpu().numpy()
            want = oracle.attentional_propagation(i["x"], i["source"], params, residual=None if res is None else i["x"])
            d = np.abs(y - want)
            print("b=5 residual=%s: max abs err %.3e (at %s), mean %.3e, finite %s" % (res is not None, d.max(), np.unravel_index(d.argmax(), d.shape),
                                                                                   d.mean(), np.isfinite(y).all()))
        # self attention (source is x): the same image serves both
        y = ops.attentional_propagation(x, x, P, residual=x).cpu().numpy()
        want = oracle.attentional_propagation(i["x"], i["x"], params, residual=i["x"])
        print("b=5 self: max abs err %.3e" % np.abs(y - want).max())
        # more problems than workgroups: against the round-4 composition in a child process
        i = synth.gnn_inputs(seed=8, b=700, C=C, n=n)
        x, s = torch.from_numpy(i["x"]).to(dev), torch.from_numpy(i["source"]).to(dev)
        y = ops.attentional_propagation(x, s, P, residual=x).cpu().numpy()
        y2 = ops.attentional_propagation(x, s, P, residual=x).cpu().numpy()
        path = "/tmp/fine_layer_child.npy"
        env = dict(os.environ, PATS_GNN_FINE="0")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", path], env=env)
        ref = np.load(path)
        d = np.abs(y - ref).reshape(700, -1).max(1)
        print("b=700 against the composition: max abs err %.3e, problems over 1e-4: %d, run-to-run identical %s"
              % (d.max(), int((d > 1e-4).sum()), np.array_equal(y, y2)))
        if (d > 1e-4).any():
            print("  first bad problems:", np.nonzero(d > 1e-4)[0][:20])
    if "--stack" in sys.argv:                        # four layers on 2 x 2048 rows = 4096 problems a layer
        names = ["self", "cross", "self", "cross"]
        def pars(i):
            q = synth.gnn_params(seed=20 + i, C=C)
            if "--zeros" in sys.argv:                # the same instruction stream on all-zero operands: what the power budget costs (DVFS)
                q = {k: (np.ones_like(v) if k.endswith("running_var") else np.zeros_like(v)) for k, v in q.items()}
            return q
        layers = [ops.PropagationParams(pars(i)) for i in range(4)]
        d0, d1 = torch.randn((2048, C, n), device=dev), torch.randn((2048, C, n), device=dev)
        if "--zeros" in sys.argv:                    # same instruction stream on all-zero operands: what the power budget costs (DVFS)
            d0.zero_(); d1.zero_()
        out = (torch.empty_like(d0), torch.empty_like(d1))
        t = timeit(lambda: ops.attentional_gnn(d0, d1, layers, names, out=out), n=3)
        print(json.dumps({"stack_ms": round(t, 3), "layer_ms_per_4096": round(t / 4, 3)}))
        return
    b = 4096
    x = torch.randn((b, C, n), device=dev)
    s = torch.randn((b, C, n), device=dev)
    t = timeit(lambda: ops.attentional_propagation(x, s, P, residual=x))
    flops = b * n * (2.0 * C * C * 4 + 2.0 * 2 * C * 2 * C + 2.0 * 2 * C * C) + b * 4 * (2.0 * n * n * (C // 4)) * 2
    print(json.dumps({"layer_ms_per_4096": round(t, 3), "algorithmic_TF": round(flops / t / 1e9, 1),
                      "f16_pipe_frac_3x": round(3 * flops / t / 1e9 / 2500.0, 3), "fine": os.environ.get("PATS_GNN_FINE", "1")}))


if __name__ == "__main__":
    main()
