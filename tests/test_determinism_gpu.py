"""Run-to-run determinism at the launch sizes the bench uses.  Two things are guarded here.

(1) The fine-level Sinkhorn kernels (test_fine_level_solve_...): hipcc left `s_waitcnt lgkmcnt(0)` out in front of the
barrier at the top of their sweep loops, a wave passed the barrier with its ds_write of the scaling vector still queued,
and 1-9 of 8 192 problems per launch ended with perturbed duals (round 3; since round 4 the wait is in the source - wg_barrier() of
csrc/common.hpp - and tests/test_host_abi.py checks the shipped code).  A 388-problem parity sample sees that once in ten runs; 8 192
problems x several launches see it every time.

(2) The three kernels that carry the MFMA operand write-after-read workaround
(cost65_device.hpp / mfma_tile.hpp / gnn.hip: `sched_barrier` + `s_nop` fences between the VALU conversions that rewrite
the A / B registers and the MFMAs that read them), AT THE LAUNCH SIZES THE BENCH USES.  The hazard only showed with
three waves per SIMD queueing on the matrix pipe and hit 100-600 of 65 536 rows, different ones every run - a
4 096-problem parity test does not load the chip enough to see it.  Each case: three launches on the same inputs must
be bit-identical, and a slice is held to the CPU oracle.  A compiler bump that re-opens the hazard fails here."""
import numpy as np
import pytest
import torch

from pats_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from pats_amd import ops as o
    return o


def _desc_pair(shape, gen, drop=0.12):
    dev = "cuda"
    base = torch.randn(shape, device=dev, generator=gen)
    d0 = 3.0 * (base + 0.3 * torch.randn(shape, device=dev, generator=gen))
    d1 = 3.0 * (base + 0.3 * torch.randn(shape, device=dev, generator=gen))
    gone = torch.rand((shape[0], 1, shape[2]), device=dev, generator=gen) < drop
    d0 = torch.where(gone, 3.12 * torch.randn(shape, device=dev, generator=gen), d0)
    d0[:, :, -1] *= 0.5
    d1[:, :, -1] *= 0.5
    return d0.contiguous(), d1.contiguous()


THIRD_CHILD = r"""
import sys, torch
sys.path.insert(0, %(repo)r)
from pats_amd import ops, synth
P = 414720
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
runs = [ops.third_level(d0.contiguous(), d1.contiguous(), sc, p_s, p_t, outdoor=True) for _ in range(4)]   # launch 0 is the process's first
torch.cuda.synchronize()
touched, worst = [], 0.0
for r in runs[1:]:
    d = (r[1] - runs[0][1]).abs()
    touched.append(int((d.reshape(P, -1).max(dim=1).values > 0).sum()))
    worst = max(worst, float(d.max()))
    for a, b in ((runs[0][0], r[0]), (runs[0][2], r[2]), (runs[0][3], r[3])):
        assert torch.equal(a, b), "index outputs differ between two launches"
print("TOUCHED", touched, "WORST", worst)
"""


def _third_level_child(variant):
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("PATS_THIRD_VARIANT", None)
    if variant:
        env["PATS_THIRD_VARIANT"] = variant
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", THIRD_CHILD % {"repo": repo}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("TOUCHED")][-1]
    touched = eval(line.split("WORST")[0].replace("TOUCHED", ""))
    return touched, float(line.split("WORST")[1])


def test_third_level_414720_problems_identical_from_the_first_launch_of_a_process(ops, oracle):
    """ops.third_level (third_fused3_kernel, 3 waves per SIMD) at the 414 720 problems of a 16-pair launch, in a FRESH process
    with no pre-heat: launches 0..3 on the same inputs are bit-identical - launch 0 included.

    History.  Round 3 found that the first full-size launch of the fp16-split instantiation in a process differs from every
    later launch in 0-6 problems (<= 1e-4 px; scores and kernel matrix identical, the scalings leave the common trajectory
    between sweep 16 and 64) and called it a power-state transient.  Round 4 measured that explanation away
    (profiles/r04_third_first_launch.md): launches after 30 / 100 s of idle are clean, a warm-up on other memory removes it,
    touching the inputs or pre-heating with matmuls does not, the stagger of the first wave front is irrelevant, the affected
    problems sit at two instants of that first launch.  No cause was found, so the DEFAULT became the instantiation that has
    never shown it (fp32-MFMA cost build, same sweep loop: 0 of 41 fresh processes) and this test holds it to the contract;
    the fp16-split build (8 % faster) left the production library - one of 60 fresh processes had a problem 0.25 px off on
    its first launch, a hundred times the parity gate - and lives in libpats_amd_diag.so (tools/third_first_launch.py)."""
    touched, worst = _third_level_child(None)
    assert touched == [0, 0, 0] and worst == 0.0, (touched, worst)
    # and the values are the right ones: the first 4 096 problems against the oracle (this process, default build)
    P, n = 8192, 4096
    gen = torch.Generator(device="cuda")
    gen.manual_seed(synth.SEED + 300)
    d0, d1 = _desc_pair((P, 128, 65), gen)
    sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device="cuda", generator=gen)) * synth.LN256 - synth.LN256 / 2)
    p_s = torch.randint(1, 23, (P, 2), device="cuda", generator=gen) * 4
    p_t = torch.randint(0, 25, (P, 2), device="cuda", generator=gen) * 4
    m0, m1, label, ifm = ops.third_level(d0, d1, sc, p_s, p_t, outdoor=True)
    S = oracle.cost(d0[:n].cpu().numpy(), d1[:n].cpu().numpy())
    scn = sc[:n].cpu().numpy()
    Zr = oracle.log_optimal_transport2(S, 1.0, scn, 100)
    sq = np.sqrt(scn + np.float32(1e-8)).astype(np.float32)
    r0, r1, _, rlabel, rifm = oracle.compute_result(np.exp(Zr), sq, sq, p_s[:n].cpu().numpy(), p_t[:n].cpu().numpy(), True)
    assert np.array_equal(label[:n * 16].cpu().numpy(), rlabel)
    assert np.array_equal(ifm[:n].cpu().numpy().astype(bool), rifm.astype(bool))
    assert np.array_equal(m0[:n].cpu().numpy(), r0)
    assert np.abs(m1[:n].cpu().numpy() - r1).max() <= 3e-4 * 8


def test_cost_20736_fine_problems_three_launches_identical(ops, oracle):
    """ops.cost (cost_mfma_kernel, mfma_tile.hpp) at 20 736 x [264,145]: the fine level of a 48-pair step."""
    B = 20736
    gen = torch.Generator(device="cuda")
    gen.manual_seed(synth.SEED + 301)
    d0, d1 = _desc_pair((B, 264, 145), gen)
    outs = [ops.cost(d0, d1) for _ in range(3)]       # no pre-heat: the first launch counts
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for sl in (slice(0, 48), slice(B - 48, B)):
        want = np.einsum("bdn,bdm->bnm", d0[sl].cpu().numpy().astype(np.float64), d1[sl].cpu().numpy().astype(np.float64)) \
            / np.sqrt(264.0) * 0.1
        got = outs[0][sl].cpu().numpy()
        assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())
        np.testing.assert_allclose(got, oracle.cost(d0[sl].cpu().numpy(), d1[sl].cpu().numpy()), atol=2e-5, rtol=1e-5)


FINE_OT_CHILD = r"""
import sys, torch, numpy as np
sys.path.insert(0, %(repo)r)
from pats_amd import ops, synth
g = torch.Generator(device="cuda"); g.manual_seed(synth.SEED + 310)
B = 20736

def pair(shape):
    base = torch.randn(shape, device="cuda", generator=g)
    return 3.0 * (base + 0.3 * torch.randn(shape, device="cuda", generator=g)), 3.0 * (base + 0.3 * torch.randn(shape, device="cuda", generator=g))
d0, d1 = pair((B, 264, 145))
# launch 0 of cost_mfma_kernel<true, false> in this process, nothing before it
S = [ops.cost(d0, d1).clone() for _ in range(4)]
torch.cuda.synchronize()
print("COST", [int((s != S[0]).flatten(1).any(1).sum()) for s in S[1:]], bool(torch.isfinite(S[0]).all()))
P = 19995
ns = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 144), device="cuda", generator=g)) * synth.LN256 - synth.LN256 / 2)
one = torch.tensor(1.0, device="cuda")
# launch 0 of sinkhorn_blk145w2_kernel<2> in this process: the production call of the fine level (batch.fine_solve_stage)
Z = [ops.cost_ot(d0[:P].contiguous(), d1[:P].contiguous(), 2, one, ns, 100, bias_k=2.0).clone() for _ in range(4)]
torch.cuda.synchronize()
print("OT", [int((z != Z[0]).flatten(1).any(1).sum()) for z in Z[1:]], bool(torch.isfinite(Z[0]).all()), ops.sinkhorn_fallbacks())
np.savez(sys.argv[1], d0=d0[:24].cpu().numpy(), d1=d1[:24].cpu().numpy(), ns=ns[:24].cpu().numpy(), S=S[0][:24].cpu().numpy(), Z=Z[0][:24].cpu().numpy(),
         d0t=d0[P - 8:P].cpu().numpy(), d1t=d1[P - 8:P].cpu().numpy(), nst=ns[P - 8:P].cpu().numpy(), Zt=Z[0][P - 8:P].cpu().numpy())
"""


def test_fine_level_cost_build_and_ot_identical_from_the_first_launch_of_a_process(oracle, tmp_path):
    """Round-5 verdict 3b: the headline's own split-fp16 MFMA kernel - cost_mfma_kernel<true, false> at 20 736 x [264, 145], the
    fine level of a 48-pair step - and the fine-level solver sinkhorn_blk145w2_kernel<2> at 19 995 problems, each in a FRESH
    process with nothing launched before it: launches 0..3 on the same inputs are bit-identical, launch 0 included (the property the
    quarantined fp16-split third-level build lacked, profiles/r05_third_first_launch.md), no problem left the guard, and the first
    24 + last 8 problems of launch 0 agree with the oracle (scores to 2e-5, both argmax vectors exactly, transport mass to 1e-4)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "fine_first_launch.npz")
    r = subprocess.run([sys.executable, "-c", FINE_OT_CHILD % {"repo": repo}, dump], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = {ln.split()[0]: ln for ln in r.stdout.splitlines() if ln.startswith(("COST", "OT"))}
    assert lines["COST"].endswith("[0, 0, 0] True"), lines
    assert lines["OT"].endswith("[0, 0, 0] True 0"), lines
    g = np.load(dump)
    for d0, d1, ns, Z, S in ((g["d0"], g["d1"], g["ns"], g["Z"], g["S"]), (g["d0t"], g["d1t"], g["nst"], g["Zt"], None)):
        So = oracle.cost(d0, d1)
        if S is not None:
            np.testing.assert_allclose(S, So, atol=2e-5, rtol=1e-5)
        Zr = oracle.dustbin_bias(oracle.log_optimal_transport2(So, 1.0, ns, 100), 2.0)
        wr, wc = oracle.argmax(Zr)
        assert np.array_equal(Z.argmax(2), wr) and np.array_equal(Z.argmax(1), wc)
        e, er = np.exp(Z.astype(np.float64)), np.exp(Zr.astype(np.float64))
        assert np.abs(e[:, :-1, :-1] - er[:, :-1, :-1]).max() <= 1e-4


def test_weights_stationary_conv_8192_problems_three_launches_identical(ops, oracle):
    """ops.conv1d on a 128 -> 128 channel product over 8 192 x 65 columns: conv_ws_kernel (gnn.hip), the tile the
    third-level GNN layers run on."""
    b = 8192
    gen = torch.Generator(device="cuda")
    gen.manual_seed(synth.SEED + 302)
    x = torch.randn((b, 128, 65), device="cuda", generator=gen)
    w = torch.randn((128, 128, 1), device="cuda", generator=gen) / 11.0
    bias = torch.randn((128,), device="cuda", generator=gen)
    outs = [ops.conv1d(x, w, bias) for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for sl in (slice(0, 40), slice(b - 40, b)):
        want = oracle.conv1d(x[sl].cpu().numpy(), w.cpu().numpy(), bias.cpu().numpy())
        np.testing.assert_allclose(outs[0][sl].cpu().numpy(), want, atol=5e-5, rtol=2e-4)


@pytest.mark.parametrize("mode", ["ot2", "given_marginals", "ot2_four_wave", "ot2_log_domain"])
def test_fine_level_solve_8192_problems_six_launches_identical(ops, oracle, mode):
    """ops.log_optimal_transport2 / ops.log_sinkhorn_iterations on 8 192 x 145 x 145, 100 sweeps: six launches on the same
    scores are bit-identical, and a slice is held to the oracle.  Both fine-level solvers are pinned while both ship:
      ot2, given_marginals   sinkhorn_blk145w2_kernel (two waves per problem, csrc/sinkhorn_blk2w.hip: the default since round 4)
      ot2_four_wave          sinkhorn_blk145_kernel (four waves, csrc/sinkhorn_blk.hip; PATS_FINE_W2=0, read once per process: a child)
      ot2_log_domain         sinkhorn_rc_kernel in the forced log domain (ops.set_sinkhorn_mode('log'), a child)
    Before the round-3 barrier fix the four-wave kernel failed this in every run (3-9 problems per pair of launches, |dZ| up to 2e-2)."""
    import os
    import subprocess
    import sys
    if mode in ("ot2_log_domain", "ot2_four_wave"):
        env = dict(os.environ, PATS_FINE_W2="0") if mode == "ot2_four_wave" else dict(os.environ)
        n_prob = 8192 if mode == "ot2_four_wave" else 4096
        code = ("import sys, torch; sys.path.insert(0, %r); from pats_amd import ops\n"
                "%s"
                "g = torch.Generator(device='cuda'); g.manual_seed(5)\n"
                "S = 2.0 * torch.randn((%d, 145, 145), device='cuda', generator=g)\n"
                "ns = torch.exp(0.3 * torch.randn((%d, 1, 144), device='cuda', generator=g))\n"
                "outs = [ops.log_optimal_transport2(S, 1.0, ns, 100) for _ in range(6)]\n"
                "torch.cuda.synchronize()\n"
                "assert all(torch.equal(outs[0], o) for o in outs[1:]), 'launches differ'\n"
                "print('FALLBACKS', ops.sinkhorn_fallbacks(reset=True))\n") % (
                    os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "ops.set_sinkhorn_mode('log')\n" if mode == "ot2_log_domain" else "", n_prob, n_prob)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return
    R = 8192
    gen = torch.Generator(device="cuda")
    gen.manual_seed(synth.SEED + 310)
    d0, d1 = _desc_pair((R, 264, 145), gen, drop=0.05)
    S = ops.cost(d0, d1)
    del d0, d1
    ns = torch.exp(0.3 * torch.randn((R, 1, 144), device="cuda", generator=gen))
    if mode == "ot2":
        fn = lambda: ops.log_optimal_transport2(S, 1.0, ns, 100)
    else:
        nsf = ns.reshape(R, 144)
        norm = -torch.log(144.0 + nsf.sum(1, keepdim=True))
        log_mu = torch.cat([norm.expand(R, 144), torch.log(nsf.sum(1, keepdim=True)) + norm], 1).contiguous()
        log_nu = torch.cat([torch.log(nsf) + norm, torch.log(torch.full((R, 1), 144.0, device="cuda")) + norm], 1).contiguous()
        fn = lambda: ops.log_sinkhorn_iterations(S, log_mu, log_nu, 100)
    ops.sinkhorn_fallbacks(reset=True)
    outs = [fn() for _ in range(6)]
    torch.cuda.synchronize()
    for k, o in enumerate(outs[1:]):
        bad = int((o != outs[0]).flatten(1).any(1).sum())
        assert bad == 0, "launch %d differs from launch 0 in %d of %d problems" % (k + 1, bad, R)
    if mode == "ot2":
        n = 24
        want = oracle.log_optimal_transport2(S[:n].cpu().numpy(), 1.0, ns[:n].cpu().numpy(), 100)
        got = outs[0][:n].cpu().numpy().astype(np.float64)
        # the mass gate of tests/test_gpu_parity.py: 1e-4 absolute, 5e-6 relative where masses exceed 1 (the dustbin corner, ~137 here)
        np.testing.assert_allclose(np.exp(got), np.exp(want.astype(np.float64)), atol=1e-4, rtol=5e-6)


GNN_CHILD = r"""
import sys, torch
sys.path.insert(0, %(repo)r)
from pats_amd import ops, synth
b = 25920
P = ops.PropagationParams(synth.gnn_params(seed=3, C=128))
g = torch.Generator(device="cuda"); g.manual_seed(11)
x = torch.randn((b, 128, 65), device="cuda", generator=g); s = torch.randn((b, 128, 65), device="cuda", generator=g)
for train in (False, True):
    runs = [ops.attentional_propagation(x, s, P, bn_train=train, residual=x) for _ in range(4)]      # launch 0 = the process's first
    torch.cuda.synchronize()
    print("DIFF", train, [int((r != runs[0]).flatten(1).any(1).sum()) for r in runs[1:]])
"""


def test_fused_gnn_layer_25920_problems_identical_from_the_first_launch_of_a_process():
    """The fused AttentionalPropagation kernel (csrc/gnn_fused.hip: fp16-split MFMA contractions beside VALU re-splits and an
    in-register softmax, two waves per SIMD) at the 25 920 problems of one pair, eval and batch-statistics BatchNorm, in a FRESH
    process with no pre-heat: four launches on the same inputs are bit-identical, the first included - the property the
    fp16-split third-level build did not have (profiles/r04_third_first_launch.md)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", GNN_CHILD % {"repo": repo}], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DIFF")]
    assert len(lines) == 2
    for ln in lines:
        assert ln.endswith("[0, 0, 0]"), ln


FINE_STACK_CHILD = r"""
import sys, torch
sys.path.insert(0, %(repo)r)
from pats_amd import ops, synth
C, n, rows = 264, 145, 2048
layers = [ops.PropagationParams(synth.gnn_params(seed=20 + i, C=C)) for i in range(4)]
names = ["self", "cross", "self", "cross"]
g = torch.Generator(device="cuda"); g.manual_seed(12)
d0 = torch.randn((rows, C, n), device="cuda", generator=g); d1 = torch.randn((rows, C, n), device="cuda", generator=g)
runs = []
for _ in range(4):                                   # launch 0 = the process's first
    a, b = ops.attentional_gnn(d0, d1, layers, names)
    runs.append(torch.cat([a, b]).clone())
torch.cuda.synchronize()
print("DIFF", [int((r != runs[0]).flatten(1).any(1).sum()) for r in runs[1:]], bool(torch.isfinite(runs[0]).all()))
"""


def test_fine_level_gnn_stack_4096_problems_identical_from_the_first_launch_of_a_process():
    """The fine level's three-kernel layer (csrc/gnn_fine.hip: gnn_fine_tile_kernel on flattened 64-column tiles - fp16-split MFMA
    products, LDS DMA gathers, masked stores through a sink - and gnn_fine_attn_kernel in wave roles with its two staging waves) as a
    four-layer self / cross stack on 2 x 2 048 rows of [264, 145] in a FRESH process with no pre-heat: four runs on the same inputs are
    bit-identical, the first included (every workgroup owns 36 tiles / 16 problems: the LDS slots, the projections' scratch blocks and
    the staging buffers are all reused)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", FINE_STACK_CHILD % {"repo": repo}], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("DIFF")]
    assert len(lines) == 1 and lines[0].endswith("[0, 0, 0] True"), lines


def test_resident_streaming_solve_4097_identical_and_every_workgroup_takes_part(ops):
    """csrc/sinkhorn_stream.hip's stream_resident_kernel (BASELINE config 5's shape: all 200 sweeps of a 4097 x 4097 problem in ONE
    launch, K in registers, partials and the scaling vector handed between the 241 workgroups through `sc1` stores, a counter barrier
    and tagged granules - no fences): three solves on the same scores are bit-identical (a torn or stale hand-over would perturb the
    duals), finite, and their marginals hold (a workgroup whose partials were missed would show in the column sums)."""
    import torch
    from pats_amd import synth
    inp = synth.roofline_inputs()
    d0, d1, ns = (torch.from_numpy(inp[k]).cuda() for k in ("d0", "d1", "ns"))
    alpha = torch.tensor(float(inp["alpha"]), device="cuda")
    S = ops.cost(d0, d1)
    runs = [ops.log_optimal_transport(S, alpha, ns, 200) for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.isfinite(runs[0]).all()
    for k, r in enumerate(runs[1:]):
        assert torch.equal(r, runs[0]), "solve %d differs from solve 0" % (k + 1)
    # a sweep ends with the column update (modules.py:142): after any number of sweeps target j holds exactly its area ns_j and the
    # dustbin column the 4096 sources' mass
    cols = torch.exp(runs[0].double()).sum(1)
    nsd = ns.reshape(1, -1).double()
    assert ((cols[:, :-1] - nsd).abs() / nsd).max().item() <= 2e-5
    assert abs(cols[0, -1].item() - 4096.0) / 4096.0 <= 2e-5


def test_resident_streaming_solve_is_not_chosen_on_a_stream_with_too_few_cus(ops):
    """launch_stream asks the stream for its CU mask (hipExtStreamGetCUMask): on a stream masked to 160 of the CUs the 241-workgroup
    resident kernel could never be co-resident - the two-launch form runs instead (milliseconds, not the resident kernel's bounded
    0.2 s give-up), and agrees with the full-GPU solve to summation order."""
    import time
    import torch
    from pats_amd import synth
    inp = synth.roofline_inputs()
    d0, d1, ns = (torch.from_numpy(inp[k]).cuda() for k in ("d0", "d1", "ns"))
    alpha = torch.tensor(float(inp["alpha"]), device="cuda")
    S = ops.cost(d0, d1)
    ref = ops.log_optimal_transport(S, alpha, ns, 200)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    ms = ops.masked_stream([c for c in range(n_cu) if c // 32 < 5])
    ms.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ms):
        ops.log_optimal_transport(S, alpha, ns, 200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = ops.log_optimal_transport(S, alpha, ns, 200)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    torch.cuda.current_stream().wait_stream(ms)
    assert dt < 0.05, "the solve took %.3f s on the masked stream" % dt
    assert (z - ref).abs().max().item() <= 5e-5
