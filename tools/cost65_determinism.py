#!/usr/bin/env python3
"""Run-to-run determinism of the third-level score matrix (diagnostic builds PATS_THIRD_VARIANT=306 / 1306 of
third_fused3_kernel stop after the cost build and return per-problem checksums instead of matches): the check that
separated the MFMA operand write-after-read hazard of the fp16-split build (different rows wrong in every run) from a
numerical difference between the two contractions (cost65_device.hpp)."""
import sys, os, numpy as np, torch, subprocess
CODE = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
from pats_amd import ops, synth
inp = synth.third_inputs(seed=synth.SEED + 60, P=4096)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d0, d1, sc, ps, pt = cu(inp["d0"]), cu(inp["d1"]), cu(inp["scale"]), cu(inp["p_s"]), cu(inp["p_t"])
outs = []
for r in range(3):
    m0, m1, lab, ifm = ops.third_level(d0, d1, sc, ps, pt, outdoor=True)
    outs.append(m1.cpu().numpy().reshape(4096, 32)[:, :5].copy())
np.save(os.environ["OUT"], np.stack(outs))
'''
res = {}
for v in ("306", "1306"):
    out = "/tmp/dbg_%s.npy" % v
    subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, PATS_THIRD_VARIANT=v, PATS_THIRD_ABLATION="1", PATS_AMD_DIAG_LIB="1", OUT=out, REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), check=True)
    res[v] = np.load(out)
for v, a in res.items():
    print(v, "run-to-run identical:", np.array_equal(a[0], a[1]), np.array_equal(a[1], a[2]), " problems differing between runs:", int((a[0] != a[1]).any(1).sum()))
a, b = res["306"][0], res["1306"][0]
rel = np.abs(a - b) / (np.abs(a) + 1e-6)
print("fp32 vs f16 checksums: max rel diff per column", rel.max(0), " problems with rel diff > 1e-4:", int((rel > 1e-4).any(1).sum()))
bad = np.argwhere((rel > 1e-4).any(1)).ravel()[:5]
for p in bad: print(" problem", p, "fp32", a[p], "f16", b[p])
