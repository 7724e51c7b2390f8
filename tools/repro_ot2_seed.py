import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tools")
import fuzz_parity as fz
from pats_amd import ops
import pats_oracle as oracle
seed = 4242018695
for mode in ("log", "auto"):
    ops.set_sinkhorn_mode(mode)
    rng = np.random.default_rng(seed)
    m, n = fz.rand_shape(rng); m, n = max(m, 3), max(n, 3)
    b = int(rng.integers(1, 6))
    S = (rng.standard_normal((b, m, n)) * rng.choice([0.5, 3.0])).astype(np.float32)
    ns = np.exp(rng.uniform(-2.7, 2.7, (b, 1, n - 1))).astype(np.float32)
    k = float(rng.choice([0.0, 2.0, 3.0]))
    want = oracle.log_optimal_transport2(S, 1.0, ns, 100)
    if k: want = oracle.dustbin_bias(want, k)
    outs = [ops.log_optimal_transport2(fz.cu(S), 1.0, fz.cu(ns), 100, bias_k=k).cpu().numpy() for _ in range(4)]
    print(mode, "shape", b, m, n, "bias", k, "runs identical:", all(np.array_equal(outs[0], o) for o in outs[1:]))
    eg, ew = np.exp(outs[0].astype(np.float64)), np.exp(want.astype(np.float64))
    d = np.abs(eg - ew); i = np.unravel_index(d.argmax(), d.shape)
    print("  max |mass diff| %.3e at %s: got %.6f want %.6f (Z got %.6f want %.6f); rel %.2e" % (d.max(), i, eg[i], ew[i], outs[0][i], want[i], d.max()/ew[i]))
    # float64 reference of the same iteration
    import scipy.special as sp
    Z = S.astype(np.float64)
    msz = float(m - 1); nsd = ns.reshape(b, n - 1).astype(np.float64)
    norm = -np.log(msz + nsd.sum(1, keepdims=True))
    log_mu = np.concatenate([np.broadcast_to(norm, (b, m - 1)), np.log(nsd.sum(1, keepdims=True)) + norm], 1)
    log_nu = np.concatenate([np.log(nsd) + norm, np.log(msz) + norm * np.ones((b, 1))], 1)
    u = np.zeros((b, m)); v = np.zeros((b, n))
    for _ in range(100):
        u = log_mu - sp.logsumexp(Z + v[:, None, :], axis=2)
        v = log_nu - sp.logsumexp(Z + u[:, :, None], axis=1)
    Z64 = Z + u[:, :, None] + v[:, None, :] - norm[:, :, None]
    if k:
        Z64[:, -1, :] += np.log(k); Z64[:, :, -1] += np.log(k)
    e64 = np.exp(Z64)
    print("  vs float64: HIP %.3e  oracle %.3e  (at that entry: f64 %.6f)" % (np.abs(eg - e64).max(), np.abs(ew - e64).max(), e64[i]))
