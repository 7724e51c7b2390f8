import os, sys, subprocess, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from pats_amd import ops
g = torch.Generator(device="cuda"); g.manual_seed(5)
S = 2.0 * torch.randn((8, 145, 145), device="cuda", generator=g)
ns = torch.exp(0.3 * torch.randn((8, 1, 144), device="cuda", generator=g))
Z = ops.log_optimal_transport2(S, 1.0, ns, int(sys.argv[2]))
np.save(sys.argv[1], Z.cpu().numpy())
''' % REPO
for iters in (1, 2, 100):
    out = {}
    for mode in ("0", "1"):
        path = "/tmp/w2_%s.npy" % mode
        subprocess.run([sys.executable, "-c", code, path, str(iters)], env=dict(os.environ, PATS_FINE_W2=mode), check=True)
        out[mode] = np.load(path)
    d = np.abs(out["1"] - out["0"])
    print("iters", iters, "max diff", d.max())
    rows = d.max(axis=(0, 2)); cols = d.max(axis=(0, 1))
    print(" rows with diff > 1e-4:", np.nonzero(rows > 1e-4)[0].tolist()[:40])
    print(" cols with diff > 1e-4:", np.nonzero(cols > 1e-4)[0].tolist()[:40])
    print(" row diffs (first 20):", np.round(rows[:20], 5).tolist())
    print(" col diffs (first 40):", np.round(cols[:40], 5).tolist())
