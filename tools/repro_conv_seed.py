import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tools")
import fuzz_parity as fz
from pats_amd import ops
import pats_oracle as oracle
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 777003354680
rng = np.random.default_rng(seed)
b, n = int(rng.integers(1, 6)), int(rng.integers(1, 400))
K, M = int(rng.integers(1, 300)), int(rng.integers(1, 300))
amp = float(rng.choice([0.1, 1.0, 10.0]))
x = (amp * rng.standard_normal((b, K, n))).astype(np.float32)
w = (rng.standard_normal((M, K, 1)) / np.sqrt(K)).astype(np.float32)
bias = rng.standard_normal(M).astype(np.float32) if rng.integers(0, 2) else None
res = rng.standard_normal((b, M, n)).astype(np.float32) if rng.integers(0, 2) else None
fold = bool(rng.integers(0, 2))
print("b=%d K=%d M=%d n=%d amp=%g bias=%s res=%s fold=%s" % (b, K, M, n, amp, bias is not None, res is not None, fold))
xa, sc, sh = x, None, None
if fold:
    gam = rng.uniform(0.5, 1.5, K).astype(np.float32); bet = rng.standard_normal(K).astype(np.float32)
    sc, sh = fz.ops.bn_fold(fz.cu(x), fz.cu(gam), fz.cu(bet), 1e-5)
    xa = np.maximum(x * sc.cpu().numpy()[None, :, None] + sh.cpu().numpy()[None, :, None], 0).astype(np.float32)
ys = [ops.conv1d(fz.cu(x), fz.cu(w), None if bias is None else fz.cu(bias), sc, sh, None if res is None else fz.cu(res)).cpu().numpy() for _ in range(3)]
print("runs identical:", all(np.array_equal(ys[0], y) for y in ys[1:]))
want = oracle.conv1d(xa, w, bias)
if res is not None: want = want + res
w64 = np.einsum("mk,bkn->bmn", w[:, :, 0].astype(np.float64), xa.astype(np.float64))
if bias is not None: w64 += bias[None, :, None]
if res is not None: w64 += res
scale = max(1.0, float(np.abs(xa).max()))
print("gate atol %.3e; |HIP - oracle| max %.3e; |HIP - f64| max %.3e; |oracle - f64| max %.3e; |xa| max %.3g" % (3e-6 * scale * np.sqrt(K) + 1e-6, np.abs(ys[0] - want).max(), np.abs(ys[0] - w64).max(), np.abs(want - w64).max(), scale))
os.environ["PATS_COST_F32"] = "1"
