import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np, torch
import pats_oracle as oracle
from pats_amd import ops, synth
def cu(x): return torch.from_numpy(np.ascontiguousarray(x)).cuda()
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    C, n = 264, 145
    stack = rng.integers(0, 3) == 0
    b = int(rng.choice([1, 2, 3, 4, 5, 7, 9, 16, 33, 65, 70])) if not stack else int(rng.choice([1, 3, 5, 17]))
    wamp = float(rng.choice([0.3, 1.0, 2.0])); amp = float(rng.choice([0.1, 1.0, 4.0]))
    def pars(sd):
        q = synth.gnn_params(seed=sd, C=C)
        for k in list(q):
            if k.endswith("weight") and q[k].ndim == 3: q[k] = (q[k] * wamp).astype(np.float32)
        return q
    x = (amp * rng.standard_normal((b, C, n))).astype(np.float32); src = (amp * rng.standard_normal((b, C, n))).astype(np.float32)
    spike = rng.integers(0, 16) == 0
    if spike: x[int(rng.integers(0, b)), int(rng.integers(0, C)), int(rng.integers(0, n))] = 2500.0
    print("seed", seed, "stack", stack, "b", b, "wamp", wamp, "amp", amp, "spike", spike)
    if not stack:
        params = pars(int(rng.integers(0, 1 << 30))); self_ = bool(rng.integers(0, 2)); res = bool(rng.integers(0, 2))
        s_ = x if self_ else src
        y = ops.attentional_propagation(cu(x), cu(s_), ops.PropagationParams(params), residual=cu(x) if res else None).cpu().numpy()
        want = oracle.attentional_propagation(x, s_, params, residual=x if res else None)
        d = np.abs(y - want); print("  layer self", self_, "res", res, "max|want|", np.abs(want).max(), "max err", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "nbad", int((d > 1e-3 * np.abs(want) + 3e-5 * max(1, np.abs(want).max())).sum()))
        continue
    L = int(rng.integers(2, 5)); plist = [pars(int(rng.integers(0, 1 << 30))) for _ in range(L)]; names = [str(rng.choice(["self", "cross"])) for _ in range(L)]
    d0, d1 = x, src
    for l in range(1, L + 1):
        y0, y1 = ops.attentional_gnn(cu(x), cu(src), [ops.PropagationParams(q) for q in plist[:l]], names[:l])
        q, nm = plist[l - 1], names[l - 1]
        s0, s1 = (d1, d0) if nm == "cross" else (d0, d1)
        d0, d1 = oracle.attentional_propagation(d0, s0, q, residual=d0), oracle.attentional_propagation(d1, s1, q, residual=d1)
        e0, e1 = np.abs(y0.cpu().numpy() - d0), np.abs(y1.cpu().numpy() - d1)
        print("  after layer", l, nm, "max|want|", max(np.abs(d0).max(), np.abs(d1).max()), "max err", e0.max(), e1.max(), "at", np.unravel_index(e0.argmax(), e0.shape), np.unravel_index(e1.argmax(), e1.shape),
              "mean err", e0.mean(), e1.mean())
