#!/usr/bin/env python3
"""Run-to-run determinism of ops.third_level at a bench-size launch: which problems differ between launches on the
same inputs, by how much, and which of the answers the CPU oracle sides with.
usage: third_determinism.py [P=414720] [launches=6]   (PATS_THIRD_VARIANT / PATS_SINKHORN_LOG select the kernel)"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
from pats_amd import ops, synth
import pats_oracle as oracle
P = int(sys.argv[1]) if len(sys.argv) > 1 else 414720
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
gen = torch.Generator(device="cuda"); gen.manual_seed(synth.SEED + 300)
shape = (P, 128, 65)
base = torch.randn(shape, device="cuda", generator=gen)
d0 = 3.0 * (base + 0.3 * torch.randn(shape, device="cuda", generator=gen))
d1 = 3.0 * (base + 0.3 * torch.randn(shape, device="cuda", generator=gen))
gone = torch.rand((P, 1, 65), device="cuda", generator=gen) < 0.12
d0 = torch.where(gone, 3.12 * torch.randn(shape, device="cuda", generator=gen), d0)
d0[:, :, -1] *= 0.5; d1[:, :, -1] *= 0.5
del base, gone
sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device="cuda", generator=gen)) * synth.LN256 - synth.LN256 / 2)
p_s = torch.randint(1, 23, (P, 2), device="cuda", generator=gen) * 4
p_t = torch.randint(0, 25, (P, 2), device="cuda", generator=gen) * 4
ops.sinkhorn_fallbacks(reset=True)
import time
if os.environ.get("LDSPOISON"):
    os.environ["PATS_THIRD_LDS_POISON"] = "0x00000001"
if os.environ.get("PREHEAT"):       # a different heavy kernel right before launch 0: the GPU is not idle when it starts
    hx = torch.randn((8192, 8192), device="cuda")
    for _ in range(int(os.environ["PREHEAT"])):
        hx = (hx @ hx) * 1e-4
if os.environ.get("WARM_SAME"):      # a tiny launch of the same kernel first
    n = int(os.environ["WARM_SAME"])
    ops.third_level(d0[:n], d1[:n], sc[:n], p_s[:n], p_t[:n], outdoor=True)
    torch.cuda.synchronize()
if os.environ.get("WARM_OTHER"):     # another kernel of the library first
    xs0 = torch.randn((2048, 264, 145), device="cuda")
    for _ in range(int(os.environ["WARM_OTHER"])):
        ops.cost(xs0, xs0)
    torch.cuda.synchronize()
ref = ops.third_level(d0, d1, sc, p_s, p_t, outdoor=True, iters=int(os.environ.get("ITERS", "100")))
bad = torch.zeros((P,), dtype=torch.bool, device="cuda")
per_launch = []
keep = [ref[1]]
scrub = os.environ.get("SCRUB")
xs = torch.randn((64, 264, 145), device="cuda")
for i in range(1, L):
    if scrub:      # other kernels in between leave different LDS / cache contents behind
        xs = torch.randn((64, 264, 145), device="cuda")
        ops.cost(xs, xs)
        (xs @ xs.transpose(1, 2)).sum()
    if os.environ.get("IDLE"):      # let the GPU fall idle between launches
        torch.cuda.synchronize()
        time.sleep(float(os.environ["IDLE"]))
    if os.environ.get("LDSPOISON"):     # diag library: a different LDS fill pattern per launch
        os.environ["PATS_THIRD_LDS_POISON"] = ["0x7fc00000", "0x0", "0x3f800000", "0xff800000", "0x42f60000", "0x00000001"][i % 6]
    if os.environ.get("POISON"):    # hand the allocator blocks full of a byte pattern: does the kernel read memory it never wrote?
        val = int(os.environ["POISON"], 0)
        junk = [torch.full((n,), val, dtype=torch.uint8, device="cuda") for n in (P * 128, P * 128, P * 128, P * 16, P * 256)]
        torch.cuda.synchronize()
        del junk
    r = ops.third_level(d0, d1, sc, p_s, p_t, outdoor=True, iters=int(os.environ.get("ITERS", "100")))
    if os.environ.get("COLS"):      # diagnostic builds that return checksums in the first few floats of every problem's slot
        c = int(os.environ["COLS"])
        b = (r[1].reshape(P, 32)[:, :c] != ref[1].reshape(P, 32)[:, :c]).any(-1)
    elif os.environ.get("PATS_THIRD_FINGERPRINT"):
        lab, lab0 = r[2].reshape(P, 32), ref[2].reshape(P, 32)
        fp_diff, k_diff, ab_diff = lab[:, 1] != lab0[:, 1], lab[:, 3] != lab0[:, 3], lab[:, 5] != lab0[:, 5]
        res_diff = (r[1] != ref[1]).any(-1).any(-1)
        b = fp_diff | res_diff | k_diff | ab_diff
        if int(b.sum()) and i == 1:
            for pp in torch.nonzero(b).flatten().cpu().tolist()[:6]:
                tr = [bool(lab[pp, 2 * (3 + k) + 1] != lab0[pp, 2 * (3 + k) + 1]) for k in range(7)]
                print("   p=%d: (a, b) fingerprint differs after sweep 1,2,4,8,16,32,64: %s" % (pp, tr))
        if int(b.sum()):
            print("launch %d: problems differing in: scores %d, kernel matrix + marginals %d, scalings after the sweeps %d, results %d"
                  % (i, int(fp_diff.sum()), int(k_diff.sum()), int(ab_diff.sum()), int(res_diff.sum())))
    else:
        b = (r[1] != ref[1]).any(-1).any(-1) | (r[2].reshape(P, 16, 2)[..., 0] != ref[2].reshape(P, 16, 2)[..., 0]).any(-1)
    per_launch.append(int(b.sum()))
    bad |= b
    if i < 6:
        keep.append(r[1])
torch.cuda.synchronize()
print("guard fallbacks over %d launches: %d" % (L, ops.sinkhorn_fallbacks(reset=True)))
idx = torch.nonzero(bad).flatten().cpu().numpy()
print("problems differing from launch 0, per launch:", per_launch)
print("problems whose result differs between launches: %d of %d:" % (len(idx), P), idx[:40])
if os.environ.get("COLS"):
    for p in idx[:8]:
        print("p=%d" % p, [k[p].reshape(32)[:int(os.environ["COLS"])].cpu().numpy() for k in keep[:3]])
m1 = torch.stack(keep)
L = m1.shape[0]
for p in ([] if os.environ.get('COLS') else idx[:12]):
    S = oracle.cost(d0[p:p + 1].cpu().numpy(), d1[p:p + 1].cpu().numpy())
    scn = sc[p:p + 1].cpu().numpy()
    Zr = oracle.log_optimal_transport2(S, 1.0, scn, 100)
    sq = np.sqrt(scn + np.float32(1e-8)).astype(np.float32)
    r0, r1, _, rl, rifm = oracle.compute_result(np.exp(Zr), sq, sq, p_s[p:p + 1].cpu().numpy(), p_t[p:p + 1].cpu().numpy(), True)
    v = m1[:, p].cpu().numpy()
    err = np.abs(v - r1[0][None]).reshape(L, -1).max(1)
    rows = np.nonzero((v != v[0:1]).any(0).any(-1))[0]
    print("   max |launch0 - launch1| = %.3e px; launch 0 row %d: %s, launch 1: %s" % (np.abs(v[0] - v[1]).max(), rows[0], v[0][rows[0]], v[1][rows[0]]))
    print("p=%d (wave slot %d): max |mkpts1 - oracle| per launch = %s ; centre rows differing %s" % (p, p % 2048, np.array2string(err, precision=4), rows))
