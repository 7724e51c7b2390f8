"""GPU parity tests (-m gpu): the HIP path (through the C-ABI, via pats_amd.ops) against
 (a) the golden fixtures the REFERENCE produced (tests/golden, tools/make_golden.py) and
 (b) the CPU oracle on the same seeded inputs,
plus size-independent properties at the reference's full sizes.
Gates: indices identical; |exp(Z)_hip - exp(Z)_ref| <= 1e-4 element-wise and on
marginals; crops <= 1e-4 abs on 0-255 data.  Nothing here reads /root/reference."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, golden
from pats_amd import synth

pytestmark = pytest.mark.gpu
MASS_TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from pats_amd import ops as o
    return o


@pytest.fixture(autouse=True, params=["kernel", "log"])
def sinkhorn_mode(request, ops):
    """Every test runs twice: linear-domain ("kernel") solver with its guard, and the log-sum-exp
    solver forced (include/pats_amd.h PATS_SINKHORN_*)."""
    prev = ops.set_sinkhorn_mode(request.param)
    yield request.param
    ops.set_sinkhorn_mode(prev)


def cu(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def assert_mass(Zh, Zr):
    Zh = Zh.detach().cpu().numpy() if isinstance(Zh, torch.Tensor) else Zh
    eo, er = np.exp(Zh.astype(np.float64)), np.exp(np.asarray(Zr).astype(np.float64))
    # 1e-4 absolute on transport mass; the dustbin entries carry mass >> 1 (corner ~ 3e2..3e3), where
    # two fp32 evaluations of Z + u + v differ by a few ulp of Z (|Z| ~ 8 -> 1e-6 relative in exp)
    np.testing.assert_allclose(eo, er, atol=MASS_TOL, rtol=2e-6)
    np.testing.assert_allclose(eo.sum(-1), er.sum(-1), atol=MASS_TOL, rtol=3e-6)
    np.testing.assert_allclose(eo.sum(-2), er.sum(-2), atol=MASS_TOL, rtol=3e-6)
    big = er > 1e-6
    assert np.abs(Zh[big] - np.asarray(Zr)[big]).max() <= 2e-4


def test_native_library_loaded(ops):
    from pats_amd import _lib
    assert _lib.lib().pats_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libpats_amd.so" in maps


# ---- OT known answers and edge shapes ----------------------------------------------------------
def test_kat(ops):
    g = golden("ot_kat.npz")
    z1 = ops.log_optimal_transport(cu(g["s1"]), float(g["a1"]), cu(g["n1"]), 100)
    assert z1.shape == (1, 3, 4)
    assert_mass(z1, g["z1"])
    z1b = ops.log_optimal_transport(cu(g["s1"]), cu(g["a1"]), cu(g["n1"]), 100)   # 0-d tensor alpha
    assert torch.equal(z1, z1b)
    z2 = ops.log_optimal_transport2(cu(g["s2"]), cu(np.float32(1.0)), cu(g["n2"]), 100)
    assert_mass(z2, g["z2"])


@pytest.mark.parametrize("iters", [1, 3, 100])
def test_sinkhorn_raw_ragged(ops, iters):
    g = golden("sinkhorn_raw.npz")
    out = ops.log_sinkhorn_iterations(cu(g["Z"]), cu(g["log_mu"]), cu(g["log_nu"]), iters)
    np.testing.assert_allclose(out.cpu().numpy(), g["out%d" % iters], atol=3e-5, rtol=0)


def test_exact_ties_first_index(ops):
    g = golden("ot_ties.npz")
    z = ops.log_optimal_transport(cu(g["s"]), float(g["alpha"]), cu(g["ns"]), 100)
    assert_mass(z, g["z"])
    r, c = ops.argmax(z)
    assert np.array_equal(r.cpu().numpy(), g["max0"]) and np.array_equal(c.cpu().numpy(), g["max1"])
    zc = z.cpu().numpy()
    assert np.array_equal(zc[0, :, 3], zc[0, :, 7]) and np.array_equal(zc[0, 2, :], zc[0, 9, :])


def test_empty_batch_and_zero_iters(ops):
    z = ops.log_optimal_transport2(torch.zeros(0, 65, 65).cuda(), 1.0, torch.zeros(0, 1, 64).cuda(), 100)
    assert z.shape == (0, 65, 65)
    s = cu(np.random.default_rng(0).standard_normal((2, 5, 7)).astype(np.float32))
    out = ops.log_sinkhorn_iterations(s, torch.zeros(2, 5).cuda(), torch.zeros(2, 7).cuda(), 0)
    assert torch.equal(out, s)        # u = v = 0 -> Z + 0 + 0


# ---- coarse level ------------------------------------------------------------------------------
@pytest.mark.parametrize("name,seed,h,w", [("coarse_301.npz", synth.SEED, 15, 20),
                                           ("coarse_portrait.npz", synth.SEED + 20, 20, 15)])
def test_coarse_level(ops, oracle, name, seed, h, w):
    g = golden(name)
    H, W = int(g["H"]), int(g["W"])
    inp = synth.coarse_inputs(seed=seed, h=h, w=w)
    d0, d1, ns = cu(inp["d0"]), cu(inp["d1"]), cu(inp["ns"])
    S = ops.cost(d0, d1)
    np.testing.assert_allclose(S.cpu().numpy().reshape(-1)[g["S_idx"]], g["S_val"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(S.cpu().numpy(), oracle.cost(inp["d0"], inp["d1"]), atol=2e-5, rtol=1e-5)
    Z = ops.log_optimal_transport(S, float(inp["alpha"]), ns, 100)
    assert_mass(Z, g["Z"])
    Zf = ops.cost_ot(d0, d1, 1, float(inp["alpha"]), ns, 100)
    assert torch.equal(Z, Zf)
    scales = ops.colmass_sqrt(Z)
    np.testing.assert_allclose(scales.cpu().numpy(), g["scales"], atol=2e-5)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_first(Z, scales, (H, W), 32)
    r, c = ops.argmax(Z)
    assert np.array_equal(r[:, :-1].cpu().numpy(), g["max0"]) and np.array_equal(c[:, :-1].cpu().numpy(), g["max1"])
    assert np.array_equal(ifn1.cpu().numpy(), g["ifn1"]) and np.array_equal(ifn2.cpu().numpy(), g["ifn2"])
    np.testing.assert_allclose(trust.cpu().numpy(), g["whole_cost"], atol=3e-6, rtol=5e-5)
    np.testing.assert_allclose(pts.cpu().numpy(), g["average_point"], atol=1e-4)
    np.testing.assert_allclose(xs.cpu().numpy(), g["x_scale"], rtol=5e-5)
    np.testing.assert_allclose(ys.cpu().numpy(), g["y_scale"], rtol=5e-5)
    # the reference-signature entry point on the reference's own exp(Z): bounds bit-exact
    positions, ranges = ops.Compute_positions_and_ranges(H // 32, W // 32, "cuda")
    sc = cu(g["scales"]).reshape(1, -1, 1)
    whole, core, avg, xs2, ys2, bound = ops.Iterative_expand_matrix(
        cu(np.exp(g["Z"])), sc, sc, torch.tensor([0, H // 32, 0, W // 32]).cuda(), ranges, positions,
        height=H // 32, width=W // 32, iter_num=15, lower_bound=1e-5)
    assert bound.dtype == torch.int64 and np.array_equal(bound.cpu().numpy(), g["bound"])
    np.testing.assert_allclose(core.cpu().numpy(), g["core_cost"], atol=3e-6, rtol=5e-4)
    np.testing.assert_allclose(whole.cpu().numpy(), g["whole_cost"], atol=3e-6, rtol=5e-5)
    # foreign (un-annotated) positions/ranges tensors work too
    pf, rf = positions.clone(), ranges.clone()
    b2 = ops.Iterative_expand_matrix(cu(np.exp(g["Z"])), sc, sc, [0, H // 32, 0, W // 32], rf, pf,
                                     iter_num=15, lower_bound=1e-5)[5]
    assert torch.equal(b2, bound)


@pytest.mark.parametrize("name,seed,h,w", [("coarse_769.npz", synth.SEED + 21, 24, 32),          # YFCC, BASELINE configs[3]
                                           ("coarse_1901.npz", synth.SEED + 22, 38, 50)])        # the demo's size (demo.py:36): the
def test_coarse_769(ops, name, seed, h, w):                                                       # streaming solver between 769^2 and 4097^2
    g = golden(name)
    inp = synth.coarse_inputs(seed=seed, h=h, w=w)
    Z = ops.cost_ot(cu(inp["d0"]), cu(inp["d1"]), 1, float(inp["alpha"]), cu(inp["ns"]), 100)
    zs = Z.cpu().numpy().reshape(-1)[g["Z_idx"]]
    assert np.abs(np.exp(zs.astype(np.float64)) - np.exp(g["Z_val"].astype(np.float64))).max() <= MASS_TOL
    r, c = ops.argmax(Z)
    assert np.array_equal(r[:, :-1].cpu().numpy(), g["max0"]) and np.array_equal(c[:, :-1].cpu().numpy(), g["max1"])
    e = np.exp(Z.cpu().numpy().astype(np.float64))
    np.testing.assert_allclose(e.sum(2), g["row_mass"], atol=MASS_TOL, rtol=1e-5)
    np.testing.assert_allclose(e.sum(1), g["col_mass"], atol=MASS_TOL, rtol=1e-5)
    scales = ops.colmass_sqrt(Z)
    out = ops.est_position_first(Z, scales, (32 * h, 32 * w), 32)
    np.testing.assert_allclose(out[1].cpu().numpy(), g["average_point"], atol=1e-4)
    assert np.array_equal(out[4].cpu().numpy(), g["ifn1"]) and np.array_equal(out[5].cpu().numpy(), g["ifn2"])


def test_split_and_compute_imgs(ops, oracle):
    g = golden("coarse_301.npz")
    ifn1 = cu(g["ifn1"])
    sum_cycle = torch.cumsum(torch.logical_not(ifn1).int(), dim=1)
    n, second, third = ops.split_patches(sum_cycle[0], 15, 20, 40)
    assert n == int(g["split40_cycle"]) and second == g["split40_second"].tolist() \
        and third == g["split40_third"].tolist()
    left, right = synth.image_pair()
    nl, nr, xsn, ysn, avn, bound5 = ops.Compute_imgs_ex(cu(g["x_scale"]), cu(g["y_scale"]),
                                                        cu(g["average_point"]), ifn1, cu(left), cu(right),
                                                        width=20, height=15)
    assert np.array_equal(bound5.cpu().numpy(), g["resize_bound"])
    assert nr.shape == (int(g["K"]), 96, 96, 3) and nl.shape == nr.shape
    np.testing.assert_allclose(xsn.cpu().numpy(), g["x_scale_new"], rtol=1e-6)
    np.testing.assert_allclose(ysn.cpu().numpy(), g["y_scale_new"], rtol=1e-6)
    np.testing.assert_allclose(avn.cpu().numpy(), g["average_new"], atol=1e-5)
    nrc, nlc = nr.cpu().numpy(), nl.cpu().numpy()
    np.testing.assert_allclose(nrc[g["crop_pick"]], g["right_pick"], atol=1e-4)
    np.testing.assert_allclose(nrc.astype(np.float64).sum((1, 2, 3)), g["right_sum"], rtol=1e-6)
    wts = np.arange(96 * 96 * 3, dtype=np.float64).reshape(96, 96, 3)
    np.testing.assert_allclose((nrc.astype(np.float64) * wts).sum((1, 2, 3)), g["right_wsum"], rtol=1e-6)
    np.testing.assert_array_equal(nlc[g["crop_pick"]][:, ::4, ::4], g["left_pick"])
    np.testing.assert_allclose(nlc.astype(np.float64).sum((1, 2, 3)), g["left_sum"], rtol=1e-9)
    # the drop-in native boundary: module `tensor_resize`, padded CHW source, same crops
    import tensor_resize
    src = torch.nn.functional.pad(cu(right), (0, 0, 128, 128, 128, 128)).permute(0, 3, 1, 2).contiguous()
    crops = tensor_resize.tensor_resize(src, bound5)
    assert crops.shape == (int(g["K"]), 3, 96, 96) and crops.device == src.device
    assert torch.allclose(crops.permute(0, 2, 3, 1), nr, atol=1e-5)
    np.testing.assert_allclose(crops.cpu().numpy(), oracle.tensor_resize(src.cpu().numpy(), g["resize_bound"]),
                               atol=1e-4)


def test_chunk_crops_are_slices(ops):
    """first_layer.py:136-146 calls Compute_imgs once per chunk mask; because the cumsum of matched
    flags is monotone, chunk (lo, hi] is rows [lo, hi) of the all-matched crops - one gather per pair."""
    g = golden("coarse_301.npz")
    ifn1 = cu(g["ifn1"])
    sum_cycle = torch.cumsum(torch.logical_not(ifn1).int(), dim=1)
    left, right = synth.image_pair()
    L, R = cu(left), cu(right)
    xs, ys, pts = cu(g["x_scale"]), cu(g["y_scale"]), cu(g["average_point"])
    nl, nr = ops.Compute_imgs(xs, ys, pts, ifn1, L, R, width=20, height=15)[:2]
    K = nr.shape[0]
    for lo, hi in g["split40_second"].tolist():
        mask = torch.where(torch.logical_and(ifn1 == False,  # noqa: E712  (first_layer.py:137-138)
                                             torch.logical_and(sum_cycle > lo, sum_cycle <= hi)), False, True)
        cl, cr = ops.Compute_imgs(xs, ys, pts, mask, L, R, width=20, height=15)[:2]
        assert torch.equal(cr, nr[lo:min(hi, K)]) and torch.equal(cl, nl[lo:min(hi, K)])


def test_compute_imgs_batch_of_images(ops):
    """utils.py:1343-1393 with a batch of two images equals the two single-image calls concatenated
    (checked against the reference itself when this test was written: sequence = img * 10000 + patch),
    with or without the caller supplying the crop counts."""
    g = golden("coarse_301.npz")
    left, right = synth.image_pair()
    L, R = cu(left), cu(right)
    xs, ys, pts, ifn = cu(g["x_scale"]), cu(g["y_scale"]), cu(g["average_point"]), cu(g["ifn1"])
    ifn_b = ifn.clone()
    ifn_b[0, ::7] = True
    xs2, ys2, pts2 = torch.cat([xs, xs * 0.9]), torch.cat([ys, ys * 1.1]), torch.cat([pts, pts])
    ifn2, L2, R2 = torch.cat([ifn, ifn_b]), torch.cat([L, L.flip(2)]), torch.cat([R, R.flip(1)])
    both = ops.Compute_imgs(xs2, ys2, pts2, ifn2, L2, R2, width=20, height=15)
    a = ops.Compute_imgs(xs2[:1], ys2[:1], pts2[:1], ifn2[:1], L2[:1], R2[:1], width=20, height=15)
    b = ops.Compute_imgs(xs2[1:], ys2[1:], pts2[1:], ifn2[1:], L2[1:], R2[1:], width=20, height=15)
    for k in range(5):
        assert torch.equal(both[k], torch.cat([a[k], b[k]]))
    counts = [int((~ifn2[0]).sum()), int((~ifn2[1]).sum())]
    known = ops.Compute_imgs(xs2, ys2, pts2, ifn2, L2, R2, width=20, height=15, known_count=counts)
    for k in range(5):
        assert torch.equal(known[k], both[k])


def test_tensor_resize_edges_and_empty(ops):
    import tensor_resize
    g = golden("resize_small.npz")
    out = tensor_resize.tensor_resize(cu(g["src"]), cu(g["bound"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], atol=1e-4)
    empty = tensor_resize.tensor_resize(cu(g["src"]), torch.zeros(0, 5, dtype=torch.int64).cuda())
    assert empty.shape == (0, 3, 96, 96)
    src_before = cu(g["src"])
    keep = src_before.clone()
    tensor_resize.tensor_resize(src_before, cu(g["bound"]))
    assert torch.equal(src_before, keep)                      # inputs are borrowed, never written
    with pytest.raises(RuntimeError):
        tensor_resize.tensor_resize(cu(g["src"]).double(), cu(g["bound"]))


def test_tensor_resize_against_compiled_reference(ops):
    """When the reference's own library.cpp build travelled (oracle/_ref), compare against it
    directly (the reference on CPU tensors, in a subprocess) - the strongest pin of the native boundary."""
    from conftest import reference_tensor_resize
    rng = np.random.default_rng(7)
    src = rng.uniform(0, 255, (2, 3, 200, 240)).astype(np.float32)
    y0 = rng.integers(0, 150, 64); x0 = rng.integers(0, 180, 64)
    bound = np.stack([y0, y0 + rng.integers(1, 50, 64), x0, x0 + rng.integers(0, 59, 64),
                      rng.integers(0, 2, 64) * 10000 + np.arange(64)], 1).astype(np.int64)
    want = reference_tensor_resize(src, bound)
    if want is None:
        pytest.skip("oracle/_ref not built")
    import tensor_resize
    got = tensor_resize.tensor_resize(torch.from_numpy(src).cuda(), torch.from_numpy(bound).cuda())
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-4)
    np.testing.assert_allclose(ops.tensor_resize(torch.from_numpy(src).cuda(), torch.from_numpy(bound).cuda()).cpu().numpy(),
                               want, atol=1e-4)                      # the ctypes route over the same C-ABI symbol


# ---- fine level --------------------------------------------------------------------------------
@pytest.mark.parametrize("name,k", [("fine_145.npz", 2.0), ("fine_145_indoor.npz", 3.0)])
def test_fine_level(ops, name, k):
    g = golden(name)
    B = int(g["B"])
    inp = synth.fine_inputs(seed=int(g["seed"]), B=B)
    d0, d1 = cu(inp["d0"]), cu(inp["d1"])
    sx, sy = cu(inp["scale_x"]), cu(inp["scale_y"])
    S = ops.cost(d0, d1)
    np.testing.assert_allclose(S.cpu().numpy().reshape(-1)[g["S_idx"]], g["S_val"], atol=2e-5, rtol=1e-5)
    Z0 = ops.log_optimal_transport2(S, 1.0, sx * sy, 100)
    e = np.exp(Z0.cpu().numpy().astype(np.float64))
    np.testing.assert_allclose(e.sum(2), g["row_mass"], atol=MASS_TOL, rtol=1e-5)
    np.testing.assert_allclose(e.sum(1), g["col_mass"], atol=MASS_TOL, rtol=1e-5)
    Z = ops.dustbin_bias_(Z0.clone(), k)
    assert_mass(Z, g["Z"])
    Zf = ops.cost_ot(d0, d1, 2, 1.0, sx * sy, 100, bias_k=k)     # bias folded into the epilogue
    assert torch.equal(Z, Zf)
    trust, pts, xs, ys, ifn1, ifn2 = ops.est_position_second(Z, sx, sy, [96, 96], 8)
    assert np.array_equal(ifn1.cpu().numpy(), g["ifn1"]) and np.array_equal(ifn2.cpu().numpy(), g["ifn2"])
    np.testing.assert_allclose(trust.cpu().numpy(), g["whole_cost"], atol=3e-6, rtol=5e-5)
    np.testing.assert_allclose(pts.cpu().numpy(), g["average_point"], atol=1e-4)
    np.testing.assert_allclose(xs.cpu().numpy(), g["x_scale"], rtol=5e-5)
    positions, ranges = ops.Compute_positions_and_ranges(12, 12, "cuda")
    out = ops.Iterative_expand_matrix(cu(np.exp(g["Z"])), sx.reshape(B, -1, 1), sy.reshape(B, -1, 1),
                                      [0, 12, 0, 12], ranges, positions, iter_num=8, lower_bound=1e-3)
    assert np.array_equal(out[5].cpu().numpy(), g["bound"])
    np.testing.assert_allclose(out[1].cpu().numpy(), g["core_cost"], atol=3e-6, rtol=5e-4)


def test_expansion_index_tables_are_checked_not_ignored(ops):
    """a9 / a10: the expansion kernel forms positions / ranges itself.  Tables that arrive WITHOUT the annotation of
    ops.Compute_positions_and_ranges - the reference's own tensors, uploaded - are compared with the canonical ones: equal ->
    same bits as the annotated call and the reference's rectangles; a shifted `ranges` (which the reference honours: 311 bounds
    of this case change, positions_ranges.npz) -> refused, never silently replaced."""
    g = golden("positions_ranges.npz")
    f = synth.fine_inputs(seed=synth.SEED + 1, B=2)
    Z = ops.cost_ot(cu(f["d0"]), cu(f["d1"]), 2, 1.0, cu(f["scale_x"] * f["scale_y"]), 100)
    P = ops.exp(Z)
    sx, sy = cu(f["scale_x"]).reshape(2, -1, 1), cu(f["scale_y"]).reshape(2, -1, 1)
    kw = dict(iter_num=8, lower_bound=1e-3, width=12, height=12)
    pos, rng_ = ops.Compute_positions_and_ranges(12, 12, "cuda")
    want = ops.Iterative_expand_matrix(P, sx, sy, [0, 12, 0, 12], rng_, pos, **kw)
    assert np.array_equal(want[5].cpu().numpy(), g["bound_canonical"])
    fpos, frng = cu(g["positions_12x12"]), cu(g["ranges_12x12"])             # foreign: no annotation
    assert not hasattr(fpos, "_pats_grid")
    got = ops.Iterative_expand_matrix(P, sx, sy, [0, 12, 0, 12], frng, fpos, **kw)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="only the index tables"):
        ops.Iterative_expand_matrix(P, sx, sy, [0, 12, 0, 12], cu(g["wrong_ranges_12x12"]), cu(g["positions_12x12"]), **kw)
    for h, w in ((15, 20), (20, 15), (12, 12), (24, 32)):                    # the device tables themselves
        p2, r2 = ops.Compute_positions_and_ranges(h, w, "cuda")
        assert np.array_equal(p2.cpu().numpy(), g["positions_%dx%d" % (h, w)]) and np.array_equal(r2.cpu().numpy(), g["ranges_%dx%d" % (h, w)])


# ---- third level -------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["third_65.npz", "third_65_indoor.npz"])
def test_third_level(ops, oracle, name):
    g = golden(name)
    P, outdoor = int(g["P"]), bool(g["outdoor"])
    inp = synth.third_inputs(seed=int(g["seed"]), P=P)
    d0, d1, scale = cu(inp["d0"]), cu(inp["d1"]), cu(inp["scale"])
    Z = ops.cost_ot(d0, d1, 2, 1.0, scale, 100)
    assert_mass(Z, g["Z"])
    sxy = torch.sqrt(scale + 1e-8)
    ps, pt = cu(inp["p_s"]), cu(inp["p_t"])
    for src, is_log in ((ops.exp(Z), False), (Z, True), (cu(np.exp(g["Z"])), False)):
        m0, m1, wl, label, ifm = ops.Compute_result(src, 8, 5, sxy, sxy, ps, pt, "cuda", outdoor=outdoor,
                                                    input_is_log=is_log)
        np.testing.assert_array_equal(m0.cpu().numpy(), g["mkpts0_f"])
        np.testing.assert_allclose(m1.cpu().numpy(), g["mkpts1_f"], atol=3e-4)
        np.testing.assert_allclose(wl.cpu().numpy(), g["whole_loss"], atol=1e-6)
        assert np.array_equal(ifm.cpu().numpy(), g["if_matching1"])
        np.testing.assert_array_equal(label.cpu().numpy(), g["label"])
    # the whole step fused in one launch: descriptors in, matches out
    m0, m1, label, ifm, Zf = ops.third_level(d0, d1, scale, ps, pt, outdoor=outdoor, return_plan=True)
    assert torch.equal(Zf, Z)
    np.testing.assert_array_equal(m0.cpu().numpy(), g["mkpts0_f"])
    np.testing.assert_allclose(m1.cpu().numpy(), g["mkpts1_f"], atol=3e-4)
    assert np.array_equal(ifm.cpu().numpy(), g["if_matching1"])
    np.testing.assert_array_equal(label.cpu().numpy(), g["label"])
    m0b, m1b, labelb, ifmb = ops.third_level(d0, d1, scale, ps, pt, outdoor=outdoor)
    # the no-plan call runs the register-block kernel (different summation order than the
    # plan-writing one): same matches, sub-pixel positions within the golden tolerance
    np.testing.assert_array_equal(m0b.cpu().numpy(), g["mkpts0_f"])
    np.testing.assert_allclose(m1b.cpu().numpy(), g["mkpts1_f"], atol=3e-4)
    assert np.array_equal(ifmb.cpu().numpy(), g["if_matching1"])
    np.testing.assert_array_equal(labelb.cpu().numpy(), g["label"])
    # the 65-wide cost fast path against the oracle
    np.testing.assert_allclose(ops.cost(d0, d1).cpu().numpy(), oracle.cost(inp["d0"], inp["d1"]),
                               atol=2e-5, rtol=1e-5)
    # a6 on the 65x65 register-resident kernel with explicit marginals, against the oracle
    rng = np.random.default_rng(11)
    Zr = (2 * rng.standard_normal((5, 65, 65))).astype(np.float32)
    mu = rng.uniform(.5, 2, (5, 65)).astype(np.float32)
    nu = rng.uniform(.5, 2, (5, 65)).astype(np.float32)
    nu *= mu.sum(1, keepdims=True) / nu.sum(1, keepdims=True)
    out = ops.log_sinkhorn_iterations(cu(Zr), cu(np.log(mu)), cu(np.log(nu)), 100)
    np.testing.assert_allclose(out.cpu().numpy(), oracle.log_sinkhorn_iterations(Zr, np.log(mu), np.log(nu), 100),
                               atol=3e-5)


def test_wide_dynamic_range_trips_guard_and_falls_back(ops, oracle, sinkhorn_mode):
    """Scores spanning +-150 nats: exp(Z - r - c) underflows for most entries and the scaling
    vectors leave the 2^30 guard, so the linear-domain path must hand the problem to the
    log-sum-exp sweeps (which ATen's logsumexp-based reference handles natively)."""
    rng = np.random.default_rng(21)
    for n, P in ((65, 6), (145, 3)):
        Z = (60.0 * rng.standard_normal((P, n, n))).astype(np.float32)
        Z[0] *= 0.02                                    # one tame problem in the same launch
        ns = rng.uniform(0.5, 2.0, (P, 1, n - 1)).astype(np.float32)
        ops.sinkhorn_fallbacks(reset=True)
        got = ops.log_optimal_transport2(cu(Z), 1.0, cu(ns), 100).cpu().numpy()
        # the counter sees the wild problems leave the linear path (and nothing in forced-log mode)
        trips = ops.sinkhorn_fallbacks(reset=True)
        assert (1 <= trips <= P - 1) if sinkhorn_mode == "kernel" else trips == 0
        want = oracle.log_optimal_transport2(Z, 1.0, ns, 100)
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-5)     # |Z| ~ 200: fp32 ulp ~ 1.5e-5
        np.testing.assert_allclose(np.exp(got[0].astype(np.float64)), np.exp(want[0].astype(np.float64)),
                                   atol=MASS_TOL, rtol=2e-6)
        # structural zeros / -inf scores: rows keep mass 1 on what is left
        Zi = Z.copy() * 0.02
        Zi[:, :5, 7:20] = -np.inf
        got = ops.log_optimal_transport2(cu(Zi), 1.0, cu(ns), 100).cpu().numpy()
        want = oracle.log_optimal_transport2(Zi, 1.0, ns, 100)
        fin = np.isfinite(want)
        assert np.array_equal(np.isneginf(got), np.isneginf(want))
        np.testing.assert_allclose(got[fin], want[fin], atol=3e-5)


# ---- descriptor gathers (a15, a16) --------------------------------------------------------------
def test_fine_descriptors(ops, oracle):
    g = golden("fine_desc.npz")
    inp = synth.fine_maps()
    desc = ops.fine_descriptors([cu(inp["f0"]), cu(inp["f1"]), cu(inp["f2"])], cu(inp["title"]), cu(inp["rubbish"]))
    d = desc.cpu().numpy()
    np.testing.assert_array_equal(d.reshape(-1)[g["idx"]], g["val"])
    np.testing.assert_array_equal(d, oracle.fine_descriptors(inp["f0"], inp["f1"], inp["f2"], inp["title"], inp["rubbish"]))
    # the descriptor block it produces is exactly what the cost build consumes
    S = ops.cost(desc[0], desc[1])
    assert S.shape == (3, 145, 145) and torch.isfinite(S).all()


def test_third_descriptors(ops, oracle):
    g = golden("third_desc.npz")
    inp = synth.third_maps()
    o0, o1, ps, pt = ops.third_descriptors(cu(inp["ff0"]), cu(inp["ff1"]), cu(inp["mk0"]), cu(inp["mk1"]),
                                           cu(inp["b_ids"]), cu(inp["kenc"]), cu(inp["rubbish"]))
    assert np.array_equal(ps.cpu().numpy(), g["p_s"]) and np.array_equal(pt.cpu().numpy(), g["p_t"])
    np.testing.assert_array_equal(o0.cpu().numpy()[:, ::4, :], g["out0"])
    np.testing.assert_array_equal(o1.cpu().numpy()[:, ::4, :], g["out1"])
    r0, r1, rps, rpt = oracle.third_descriptors(inp["ff0"], inp["ff1"], inp["mk0"], inp["mk1"], inp["b_ids"],
                                                inp["kenc"], inp["rubbish"])
    np.testing.assert_array_equal(o0.cpu().numpy(), r0)
    np.testing.assert_array_equal(o1.cpu().numpy(), r1)
    # ... and feeds the fused third-level step directly
    scale = torch.ones(o0.shape[0], 1, 64, device="cuda")
    m0, m1, label, ifm = ops.third_level(o0, o1, scale, ps, pt)
    assert m0.shape == (o0.shape[0], 16, 2) and torch.isfinite(m1).all()


def test_descriptor_gathers_on_channels_last_maps_give_the_same_bits(ops, oracle):
    """Maps in torch.channels_last (what a backbone run under MIOpen emits) take the channels-last gathers: golden
    vectors, the oracle and the NCHW kernels' outputs, bit for bit; with the edge cases of the window gather (points at
    0 and 96 whose windows leave the map and wrap / clamp, first and last image) and a device-side count."""
    cl = lambda a: cu(a).contiguous(memory_format=torch.channels_last)
    inp = synth.fine_maps()
    maps = [cl(inp["f0"]), cl(inp["f1"]), cl(inp["f2"])]
    assert not maps[0].is_contiguous()
    desc = ops.fine_descriptors(maps, cu(inp["title"]), cu(inp["rubbish"]))
    np.testing.assert_array_equal(desc.cpu().numpy(), oracle.fine_descriptors(inp["f0"], inp["f1"], inp["f2"], inp["title"], inp["rubbish"]))
    g = golden("fine_desc.npz")
    np.testing.assert_array_equal(desc.cpu().numpy().reshape(-1)[g["idx"]], g["val"])
    assert torch.equal(desc, ops.fine_descriptors([cu(inp["f0"]), cu(inp["f1"]), cu(inp["f2"])], cu(inp["title"]), cu(inp["rubbish"])))
    # mixed memory formats fall back to the NCHW kernel
    assert torch.equal(desc, ops.fine_descriptors([maps[0], cu(inp["f1"]), maps[2]], cu(inp["title"]), cu(inp["rubbish"])))

    inp = synth.third_maps()
    g = golden("third_desc.npz")
    o0, o1, ps, pt = ops.third_descriptors(cl(inp["ff0"]), cl(inp["ff1"]), cu(inp["mk0"]), cu(inp["mk1"]),
                                           cu(inp["b_ids"]), cu(inp["kenc"]), cu(inp["rubbish"]))
    assert np.array_equal(ps.cpu().numpy(), g["p_s"]) and np.array_equal(pt.cpu().numpy(), g["p_t"])
    np.testing.assert_array_equal(o0.cpu().numpy()[:, ::4, :], g["out0"])
    np.testing.assert_array_equal(o1.cpu().numpy()[:, ::4, :], g["out1"])
    # the border ring of the cell grid (fixture from the reference's lines): wrapped windows, the next patch's dustbin feature
    inp = synth.third_maps_ring()
    g = golden("third_desc_ring.npz")
    for fmt in (cu, cl):
        o0, o1, ps, pt = ops.third_descriptors(fmt(inp["ff0"]), fmt(inp["ff1"]), cu(inp["mk0"]), cu(inp["mk1"]),
                                               cu(inp["b_ids"]), cu(inp["kenc"]), cu(inp["rubbish"]))
        assert np.array_equal(ps.cpu().numpy(), g["p_s"]) and np.array_equal(pt.cpu().numpy(), g["p_t"])
        np.testing.assert_array_equal(o0.cpu().numpy()[:, ::4, :], g["out0"])
        np.testing.assert_array_equal(o1.cpu().numpy()[:, ::4, :], g["out1"])
    # random maps, points on and beyond the borders (clamped / wrapped windows), every image incl. the first and the last
    rng = np.random.default_rng(5)
    B, P = 5, 203
    ff0, ff1 = rng.standard_normal((2, B, 128, 52, 52)).astype(np.float32)
    mk0 = (rng.random((P, 2)) * 96).astype(np.float32)
    mk1 = (rng.random((P, 2)) * 130 - 17).astype(np.float32)
    mk0[:8] = [[0, 0], [96, 96], [0, 96], [96, 0], [2, 2], [94, 94], [0, 50], [50, 0]]
    mk1[:8] = [[-5, -5], [200, 200], [0, 96], [96, 0], [1.9, 2.1], [94, 97], [0, 50], [50, 0]]
    b_ids = rng.integers(1, B - 1, P).astype(np.int64)      # first / last image: below (a window leaving them leaves the tensor)
    b_ids[:8] = [1, B - 2, 1, B - 2, 2, 3, 1, 2]      # windows that leave the map wrap into the neighbouring image (:127)
    kenc = rng.standard_normal((128, 64)).astype(np.float32)
    rub = rng.standard_normal((B, 128, 144)).astype(np.float32)
    rub[0, :7, :] = -0.0
    want = oracle.third_descriptors(ff0, ff1, mk0, mk1, b_ids, kenc, rub)
    got_n = ops.third_descriptors(cu(ff0), cu(ff1), cu(mk0), cu(mk1), cu(b_ids), cu(kenc), cu(rub))
    got_c = ops.third_descriptors(cl(ff0), cl(ff1), cu(mk0), cu(mk1), cu(b_ids), cu(kenc), cu(rub))
    for a, b_, w in zip(got_n, got_c, want):
        assert torch.equal(a, b_)
        np.testing.assert_array_equal(b_.cpu().numpy().view(np.uint32 if w.dtype == np.float32 else w.dtype),
                                      w.view(np.uint32 if w.dtype == np.float32 else w.dtype))
    # indices beyond the first / last image (torch.gather would raise, so would the oracle): both kernels clamp alike
    b_ids[:4] = [0, B - 1, 0, B - 1]
    b_ids[100:] = rng.integers(0, B, P - 100)
    got_n = ops.third_descriptors(cu(ff0), cu(ff1), cu(mk0), cu(mk1), cu(b_ids), cu(kenc), cu(rub))
    got_c = ops.third_descriptors(cl(ff0), cl(ff1), cu(mk0), cu(mk1), cu(b_ids), cu(kenc), cu(rub))
    assert all(torch.equal(a, b_) for a, b_ in zip(got_n, got_c))
    # device-side count: rows past it are not written
    cnt = torch.tensor([77], dtype=torch.int64, device="cuda")
    out = (torch.full((P, 128, 65), 7.0, device="cuda"), torch.full((P, 128, 65), 7.0, device="cuda"))
    c0, c1, _, _ = ops.third_descriptors(cl(ff0), cl(ff1), cu(mk0), cu(mk1), cu(b_ids), cu(kenc), cu(rub), count=cnt, out=out)
    assert torch.equal(c0[:77], got_c[0][:77]) and torch.equal(c1[:77], got_c[1][:77])
    assert bool((c0[77:] == 7.0).all()) and bool((c1[77:] == 7.0).all())


# ---- properties at the reference's full sizes (oracle would take minutes) ------------------------
def _check_marginals(Z, ns, ms):
    """Each sweep ends with the column update (modules.py:142), so after any number of sweeps the
    COLUMN marginals are exact: target j receives its area ns_j, the dustbin column receives ms.
    Row marginals (mass 1 per source patch) only hold at convergence - checked loosely."""
    e = torch.exp(Z.double())
    rows, cols = e.sum(2), e.sum(1)
    ns = ns.reshape(ns.shape[0], -1).double()
    assert ((cols[:, :-1] - ns).abs() / ns).max().item() <= 2e-5
    assert ((cols[:, -1] - ms).abs() / ms).max().item() <= 2e-5
    total = ms + ns.sum(1)
    assert ((e.sum((1, 2)) - total).abs() / total).max().item() <= 2e-5    # mass conservation
    assert (rows[:, :-1] - 1).abs().max().item() <= 0.15
    assert (rows[:, :-1] - 1).abs().mean().item() <= 2e-3


def test_full_size_third_level_properties(ops):
    P = 20000                                   # P ~ 60 * B per 640x480 pair
    inp = synth.third_inputs(seed=123, P=P)
    Z = ops.cost_ot(cu(inp["d0"]), cu(inp["d1"]), 2, 1.0, cu(inp["scale"]), 100)
    assert Z.shape == (P, 65, 65) and torch.isfinite(Z).all()
    _check_marginals(Z, cu(inp["scale"]), 64.0)
    # idempotence of the fixed point: more sweeps on the converged plan change nothing
    e = torch.exp(Z[:64])
    lm, ln = torch.log(e.sum(2)), torch.log(e.sum(1))
    Z2 = ops.log_sinkhorn_iterations(Z[:64].contiguous(), lm, ln, 5)
    assert (Z2 - Z[:64]).abs().max().item() <= 5e-4


def test_full_size_fine_level_properties(ops):
    B = 512                                     # the if_local=False cap (first_layer.py:134)
    inp = synth.fine_inputs(seed=124, B=B)
    ns = cu(inp["scale_x"] * inp["scale_y"])
    Z = ops.cost_ot(cu(inp["d0"]), cu(inp["d1"]), 2, 1.0, ns, 100)
    assert torch.isfinite(Z).all()
    _check_marginals(Z, ns, 144.0)


def _near_tie_flips(Z, got, want, axis):
    """Indices that differ from the reference's, split into near ties (the two candidates' log-plan values
    agree to fp32 noise, 4 ulp of |Z|: either solver's rounding can pick either) and real mismatches."""
    Zn = Z if axis == 1 else Z.T
    bad = np.nonzero(got != want)[0]
    real = 0
    for i in bad:
        a, b = Zn[i, got[i]], Zn[i, want[i]]
        if abs(float(a) - float(b)) > 4 * np.spacing(np.float32(max(abs(a), abs(b)))):
            real += 1
    return len(bad), real


def test_roofline_config_4096(ops):
    """BASELINE.json configs[4] against the REFERENCE's own 4097 x 4097, 200-iteration run
    (tests/golden/roofline_4097.npz: sampled scores and log-plan entries, both argmax vectors, marginals),
    plus the marginal properties."""
    g = golden("roofline_4097.npz")
    inp = synth.roofline_inputs()
    assert abs(synth.checksum(inp["d0"][:, :, :64], inp["ns"]) - float(g["in_checksum"])) < 1e-6 * abs(float(g["in_checksum"]))
    d0, d1 = cu(inp["d0"]), cu(inp["d1"])
    S = ops.cost(d0, d1)
    np.testing.assert_allclose(S.cpu().numpy().reshape(-1)[g["S_idx"]], g["S_val"], atol=2e-6, rtol=1e-5)
    Z = ops.cost_ot(d0, d1, 1, float(inp["alpha"]), cu(inp["ns"]), int(g["iters"]))
    assert Z.shape == (1, 4097, 4097) and torch.isfinite(Z).all()
    _check_marginals(Z, cu(inp["ns"]), 4096.0)
    Zc = Z.cpu().numpy()[0]
    zs = Zc.reshape(-1)[g["Z_idx"]]
    assert np.abs(np.exp(zs.astype(np.float64)) - np.exp(g["Z_val"].astype(np.float64))).max() <= MASS_TOL
    assert np.abs(zs - g["Z_val"]).max() <= 2e-4
    np.testing.assert_allclose(Zc[-1, ::8], g["Z_last_row"], atol=2e-4)
    np.testing.assert_allclose(Zc[::8, -1], g["Z_last_col"], atol=2e-4)
    e = np.exp(Zc.astype(np.float64))
    np.testing.assert_allclose(e.sum(1), g["row_mass"], atol=MASS_TOL, rtol=1e-5)
    np.testing.assert_allclose(e.sum(0), g["col_mass"], atol=MASS_TOL, rtol=1e-5)
    r, c = ops.argmax(Z)
    nr, real_r = _near_tie_flips(Zc, r[0].cpu().numpy(), g["max0"], 1)
    nc, real_c = _near_tie_flips(Zc, c[0].cpu().numpy(), g["max1"], 0)
    print("config 5 argmax: %d row / %d column indices differ from the reference's, all within 4 ulp ties" % (nr, nc))
    assert real_r == 0 and real_c == 0
    assert nr <= 4 and nc <= 4             # flat N(0, 0.01) scores: a handful of exact-noise ties at most


@pytest.mark.parametrize("B,M,N", [(1, 4700, 600),      # 294 row blocks: the column reduce's tail loop (more than 18 partials per wave)
                                   (1, 4097, 3600),     # 8 columns per thread, blocks of 17 rows (one round of workgroups)
                                   (2, 4097, 3600),     # the same as a batch: 2 x 241 workgroups are two rounds, 2 x 257 three
                                   (1, 4100, 3100),     # 7 columns per thread, blocks of 17 rows
                                   (1, 700, 4608),      # the widest problem the streaming solver takes, blocks of 16 rows
                                   (1, 4097, 4600),     # one problem, 241 blocks of 17 rows, 9 columns per thread: all sweeps in ONE launch
                                                        # (stream_resident_kernel: K in registers, grid barrier + granules), ragged last columns
                                   (1, 4100, 4608),     # the same kernel: a last block of 3 rows, the widest row (every ninth slot in use)
                                   (1, 4352, 4097),     # 256 blocks = every CU of the part, one column in the ninth slot
                                   (1, 769, 769),       # the narrow shape of the resident kernel (N <= 1024, two columns a thread): 25 blocks of 32 rows
                                   (16, 769, 769),      # ... batched as BASELINE config [3] runs it: 16 problems x 13 blocks of 64 rows, all resident
                                   (3, 1000, 1024),     # ... the widest narrow problem, 32-row blocks, a ragged last block
                                   (1, 2000, 520)])     # ... tall: 63 blocks, 9 reduce groups
def test_streaming_solver_block_shapes(ops, oracle, B, M, N):
    """csrc/sinkhorn_stream.hip picks rows per workgroup and columns per thread from the shape: every branch of that
    choice against the oracle (modules.py:137-143), four sweeps."""
    rng = np.random.default_rng(M * 7 + N + B)
    Z = rng.standard_normal((B, M, N)).astype(np.float32)
    mu = rng.uniform(0.5, 2.0, (B, M)); nu = rng.uniform(0.5, 2.0, (B, N))
    log_mu = np.log(mu / mu.sum(1, keepdims=True)).astype(np.float32)
    log_nu = np.log(nu / nu.sum(1, keepdims=True)).astype(np.float32)
    want = oracle.log_sinkhorn_iterations(Z, log_mu, log_nu, 4)
    got = ops.log_sinkhorn_iterations(cu(Z), cu(log_mu), cu(log_nu), 4).cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=5e-5, rtol=0)
    eg, ew = np.exp(got.astype(np.float64)), np.exp(want.astype(np.float64))
    np.testing.assert_allclose(eg.sum(-1), ew.sum(-1), atol=MASS_TOL, rtol=1e-5)
    np.testing.assert_allclose(eg.sum(-2), ew.sum(-2), atol=MASS_TOL, rtol=1e-5)


# ---- merge, third-level inputs, result scatter, get_result ---------------------
@pytest.mark.parametrize("name", ["merge_new.npz", "merge_old.npz", "merge_new_portrait.npz"])
def test_merge_patches_golden(ops, name):
    """second_layer.py:137-238 over three successive chunks with the scores_back hand-over; bit-exact."""
    g = golden(name)
    inp = synth.merge_inputs(seed=int(g["seed"]), h=int(g["h"]), w=int(g["w"]))
    new, h, w = bool(g["merge_new"]), inp["h"], inp["w"]
    fn = ops.merge_patches_new if new else ops.merge_patches_old
    sb = torch.zeros((1, h * w, 16, 9), dtype=torch.float64, device="cuda")
    for c, ch in enumerate(inp["chunks"]):
        tr, f2, l1 = cu(ch["trust"]), cu(ch["ifn2"]), cu(ch["ifn_L1"])
        sb_arg = sb
        out, sb = fn(tr.shape[0], tr, (h * 32, w * 32), l1, f2, sb_arg)
        assert np.array_equal(out.cpu().numpy(), g["out%d" % c])
        assert np.array_equal(tr.cpu().numpy(), g["trust%d" % c])            # in-place like the reference
        assert np.array_equal(f2.cpu().numpy(), g["ifn2_%d" % c])
        assert np.array_equal(sb_arg.cpu().numpy().astype(np.float32), g["sb_written%d" % c])
        assert bool((sb == 0).all().item()) == bool(g["sb_returned_zero%d" % c])


@pytest.mark.parametrize("new", [True, False])
def test_merge_patches_large_and_batched(ops, oracle, new):
    """YFCC-sized grid (24x32) and a batch of two images against the oracle."""
    fn = ops.merge_patches_new if new else ops.merge_patches_old
    for seed, h, w, bt in ((31, 24, 32, 1), (32, 6, 7, 2)):
        rng = np.random.default_rng(seed)
        l1 = rng.random((bt, h * w)) < 0.3
        B = int((~l1).sum())
        trust = rng.lognormal(-1.0, 0.9, (B, 144)).astype(np.float32)
        trust[::3] = np.round(trust[::3] * 4) / 4
        f2 = rng.random((B, 144)) < 0.3
        sb0 = np.where(rng.random((bt, h * w, 16, 9)) < 0.5, 0.0, np.round(rng.normal(0, 1, (bt, h * w, 16, 9)), 1) - 5000.0)
        want, wt, wf2, wsb = oracle.merge_patches(new, trust, (h * 32, w * 32), l1, f2, sb0)
        tr, ff, sb = cu(trust), cu(f2), cu(sb0)
        out, _ = fn(B, tr, (h * 32, w * 32), cu(l1), ff, sb)
        assert np.array_equal(out.cpu().numpy(), want)
        assert np.array_equal(tr.cpu().numpy(), wt) and np.array_equal(ff.cpu().numpy(), wf2)
        assert np.array_equal(sb.cpu().numpy(), wsb)
        assert 0 < (~want).sum()


def test_merge_patches_row_count_mismatch_raises(ops):
    inp = synth.merge_inputs()
    ch = inp["chunks"][0]
    sb = torch.zeros((1, 300, 16, 9), dtype=torch.float64, device="cuda")
    with pytest.raises(IndexError):
        ops.merge_patches_new(ch["trust"].shape[0] - 1, cu(ch["trust"][:-1]), (480, 640), cu(ch["ifn_L1"]),
                              cu(ch["ifn2"][:-1]), sb)
    with pytest.raises(RuntimeError):                       # CPU tensors are refused, no fallback
        ops.merge_patches_new(ch["trust"].shape[0], torch.from_numpy(ch["trust"]), (480, 640), cu(ch["ifn_L1"]),
                              cu(ch["ifn2"]), sb)


@pytest.mark.parametrize("name", ["result.npz", "result_mixed.npz"])
def test_third_inputs_scatter_get_result_golden(ops, name):
    """pats.py:53-78 + utils.py:189-213 against the reference's outputs; bit-exact."""
    g = golden(name)
    inp = synth.result_inputs(seed=int(g["seed"]), h=5, w=6, mixed_choice=bool(g["mixed"]))
    mk0, mk1, b_ids = ops.third_inputs(cu(inp["ifn2"]), cu(inp["pts"]))
    assert np.array_equal(mk0.cpu().numpy(), g["mk0"]) and np.array_equal(mk1.cpu().numpy(), g["mk1"])
    assert np.array_equal(b_ids.cpu().numpy(), g["b_ids"])
    label = torch.zeros((inp["label0"].shape[0], 2), device="cuda")
    label[:, 0] = cu(inp["label0"])
    f16, p16 = ops.refine_scatter(cu(inp["ifn2"]), cu(inp["pts"]), cu(inp["mkpts1"]), label)
    assert np.array_equal(f16.cpu().numpy(), g["ifn16"]) and np.array_equal(p16.cpu().numpy(), g["pts16"])
    ifn0, sc0 = cu(inp["ifn0"]), cu(inp["sc0"])
    sc_rows = sc0.reshape(-1, 30, 2)[torch.logical_not(ifn0)]                  # [K,2]
    args = (1, [ifn0, f16], [cu(inp["ap0"]), p16.flip(dims=[2]) / 2.0])
    tail = ([[32, 5, 6], [2, 48, 48]], [cu(inp["choice0"]), cu(inp["choice1"])])
    ml, mr = ops.get_result(*args, [sc0, sc_rows.reshape(-1, 1, 2).repeat(1, 2304, 1)], *tail)   # as pats.py:70 builds it
    assert np.array_equal(ml.cpu().numpy(), g["matches_l"]) and np.array_equal(mr.cpu().numpy(), g["matches_r"])
    ml2, mr2 = ops.get_result(*args, [sc0, sc_rows], *tail)                    # one scale per row, not materialised
    assert torch.equal(ml2, ml) and torch.equal(mr2, mr)


def test_result_chain_full_size(ops, oracle):
    """The 640x480 case (15x20 coarse cells, ~255 L2 rows, ~5e5 sub-cells) against the oracle, plus the
    size-independent property: every emitted match comes from a surviving sub-cell, in order."""
    inp = synth.result_inputs(seed=99, h=15, w=20, mixed_choice=True)
    mk0, mk1, b_ids = ops.third_inputs(cu(inp["ifn2"]), cu(inp["pts"]))
    w0, w1, wb = oracle.third_inputs(inp["ifn2"], inp["pts"])
    assert np.array_equal(mk0.cpu().numpy(), w0) and np.array_equal(mk1.cpu().numpy(), w1)
    assert np.array_equal(b_ids.cpu().numpy(), wb)
    f16, p16 = ops.refine_scatter(cu(inp["ifn2"]), cu(inp["pts"]), cu(inp["mkpts1"]), cu(inp["label0"]))
    wf, wp = oracle.refine_scatter(inp["ifn2"], inp["pts"], inp["mkpts1"], inp["label0"])
    assert np.array_equal(f16.cpu().numpy(), wf) and np.array_equal(p16.cpu().numpy(), wp)
    sc_rows = inp["sc0"][~inp["ifn0"]]
    ml, mr = ops.get_result(1, [cu(inp["ifn0"]), f16], [cu(inp["ap0"]), p16.flip(dims=[2]) / 2.0], [cu(inp["sc0"]), cu(sc_rows)],
                            [[32, 15, 20], [2, 48, 48]], [cu(inp["choice0"]), cu(inp["choice1"])])
    wl, wr = oracle.get_result(1, [inp["ifn0"], wf], [inp["ap0"], wp[:, :, ::-1] / np.float32(2.0)],
                               [inp["sc0"], np.repeat(sc_rows.reshape(-1, 1, 2), 2304, 1)], [[32, 15, 20], [2, 48, 48]],
                               [inp["choice0"], inp["choice1"]])
    assert np.array_equal(ml.cpu().numpy(), wl) and np.array_equal(mr.cpu().numpy(), wr)
    assert ml.shape[0] == int((~wf).sum()) > 50000
    # nothing survives -> empty outputs, like the reference's empty boolean selections
    none = torch.ones_like(f16)
    ml0, mr0 = ops.get_result(1, [cu(inp["ifn0"]), none], [cu(inp["ap0"]), p16], [cu(inp["sc0"]), cu(sc_rows)],
                              [[32, 15, 20], [2, 48, 48]], [cu(inp["choice0"]), cu(inp["choice1"])])
    assert ml0.shape == (0, 2) and mr0.shape == (0, 2)
    e0, e1, eb = ops.third_inputs(torch.ones((4, 144), dtype=torch.bool, device="cuda"), torch.zeros((4, 144, 2), device="cuda"))
    assert e0.shape == (0, 2) and eb.shape == (0,)


def test_third_level_rejects_unsupported_descriptor_dim(ops):
    """D = 48 is not a whole number of prefetch rings: refused (RuntimeError, like a TORCH_CHECK), not mis-computed."""
    P = 4
    z = lambda *s: torch.zeros(s, device="cuda")     # noqa: E731
    with pytest.raises(RuntimeError):
        ops.third_level(z(P, 48, 65), z(P, 48, 65), torch.ones((P, 1, 64), device="cuda"),
                        torch.zeros((P, 2), dtype=torch.int64, device="cuda"), torch.zeros((P, 2), dtype=torch.int64, device="cuda"))


# ---- randomised cross-check (a fixed-seed slice of tools/fuzz_parity.py) -------------------------------
@pytest.mark.parametrize("op", ["sinkhorn", "ot", "ot2", "cost", "expand", "resize", "merge", "result", "third", "attention"])
def test_fuzz_slice(ops, oracle, op):
    """Random shapes (ragged, tiny, resident sizes and their neighbours, tie-heavy data) against the
    oracle; tools/fuzz_parity.py runs the same generators for minutes (about 35 000 cases clean in round 1)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import fuzz_parity
    for case in range(6):
        fuzz_parity.OPS[op](np.random.default_rng(777000 + 31 * case + len(op)))


def test_gnn_deferred_overflow_protocol(ops):
    """ops.set_gnn_redo('deferred'): the fine level's packed stack and the third level's fused layer queue no gated redo chain; in-range
    inputs give the inline mode's bits and leave the device flag down, a spike beyond the fp16 range raises it (outputs void) and
    the inline mode's answer for the same inputs is finite."""
    for C, n, rows in ((264, 145, 64), (128, 65, 256)):
        layers = [ops.PropagationParams(synth.gnn_params(seed=900 + i, C=C)) for i in range(2)]
        names = ["self", "cross"]
        g = torch.Generator(device="cuda")
        g.manual_seed(33)
        d0, d1 = torch.randn((rows, C, n), device="cuda", generator=g), torch.randn((rows, C, n), device="cuda", generator=g)
        want = [t.clone() for t in ops.attentional_gnn(d0, d1, layers, names)]
        prev = ops.set_gnn_redo("deferred")
        try:
            assert prev == "inline"
            ops.gnn_overflows(reset=True)
            got = ops.attentional_gnn(d0, d1, layers, names)
            assert not ops.gnn_overflows(reset=True)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
            wild = d0.clone()
            wild[3, :, 7] *= 1e6                                   # far beyond the fp16 range of the split operands
            ops.attentional_gnn(wild, d1, layers, names)
            assert ops.gnn_overflows(reset=True)
            assert not ops.gnn_overflows(reset=False)              # the read above reset it
        finally:
            assert ops.set_gnn_redo(prev) == "deferred"
        ok = ops.attentional_gnn(wild, d1, layers, names)          # inline: the gated composition redoes the stack
        assert torch.isfinite(ok[0]).all() and torch.isfinite(ok[1]).all()


# ---- attention(query, key, value), modules.py:84-88 ----------------------
def test_attention_golden(ops):
    g = golden("attention.npz")
    cases = [dict(b=3, dim=32, heads=4, n=65), dict(b=2, dim=66, heads=4, n=145, amp=1.5),
             dict(b=1, dim=112, heads=4, n=300), dict(b=2, dim=6, heads=2, n=37, m=53, amp=2.0)]
    for c, kw in enumerate(cases):
        inp = synth.attention_inputs(seed=synth.SEED + 12 + c, **kw)
        x, prob = ops.attention(cu(inp["q"]), cu(inp["k"]), cu(inp["v"]))
        x, prob = x.cpu().numpy(), prob.cpu().numpy()
        np.testing.assert_allclose(x.reshape(-1)[g["x_idx%d" % c]], g["x_val%d" % c], atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(prob.reshape(-1)[g["p_idx%d" % c]], g["p_val%d" % c], atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(prob.sum(-1), g["p_rowsum%d" % c], atol=2e-6)
        # without prob the 65-token / dim-32 shape takes the one-wave-per-head kernel: same gates
        x2, none = ops.attention(cu(inp["q"]), cu(inp["k"]), cu(inp["v"]), return_prob=False)
        assert none is None
        np.testing.assert_allclose(x2.cpu().numpy().reshape(-1)[g["x_idx%d" % c]], g["x_val%d" % c], atol=1e-5, rtol=1e-5)


def test_attention_shapes_against_oracle(ops, oracle):
    """Ragged and degenerate shapes (one query, one key, odd dim, n != m), peaked softmax rows, the
    third-level batch size, and the properties rows-sum-to-1 / convex combination of the values."""
    for seed, kw in enumerate([dict(b=2, dim=7, heads=3, n=1, m=1), dict(b=1, dim=5, heads=1, n=33, m=2),
                               dict(b=2, dim=32, heads=4, n=64, m=96, amp=4.0), dict(b=1, dim=128, heads=2, n=200, m=640), dict(b=1, dim=112, heads=4, n=769, m=769), dict(b=1, dim=8, heads=1, n=40, m=1024),
                               # beyond 1 024 keys the workgroup takes 16 / 8 / 4 query rows: the reference's demo size (1 900 tokens,
                               # demo.py:36, dim 112 x 4 heads), a ragged 16-row case and an 8-row one
                               dict(b=1, dim=112, heads=4, n=1901, m=1901), dict(b=2, dim=24, heads=2, n=37, m=1100),
                               dict(b=1, dim=16, heads=1, n=19, m=4100),
                               dict(b=300, dim=32, heads=4, n=65)]):
        inp = synth.attention_inputs(seed=500 + seed, **kw)
        x, prob = ops.attention(cu(inp["q"]), cu(inp["k"]), cu(inp["v"]))
        wx, wp = oracle.attention(inp["q"], inp["k"], inp["v"])
        np.testing.assert_allclose(x.cpu().numpy(), wx, atol=2e-5, rtol=1e-5)
        x_np, _ = ops.attention(cu(inp["q"]), cu(inp["k"]), cu(inp["v"]), return_prob=False)
        np.testing.assert_allclose(x_np.cpu().numpy(), wx, atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(prob.cpu().numpy(), wp, atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(prob.sum(-1).cpu().numpy(), 1.0, atol=2e-6)
        vmin = cu(inp["v"]).amin(dim=3, keepdim=True)
        vmax = cu(inp["v"]).amax(dim=3, keepdim=True)
        assert bool(((x >= vmin - 1e-5) & (x <= vmax + 1e-5)).all())
    e, _ = ops.attention(torch.zeros((0, 32, 4, 65), device="cuda"), torch.zeros((0, 32, 4, 65), device="cuda"),
                         torch.zeros((0, 32, 4, 65), device="cuda"))
    assert e.shape == (0, 32, 4, 65)
    with pytest.raises(RuntimeError):          # even four query rows of 9 000 keys do not fit the LDS slab
        ops.attention(torch.zeros((1, 8, 2, 4), device="cuda"), torch.zeros((1, 8, 2, 9000), device="cuda"),
                      torch.zeros((1, 8, 2, 9000), device="cuda"))


# ---- the whole path chained: pats_amd.pipeline.forward_path vs the reference's functions in its own order ----
class _CudaNets:
    """synth.SynthNets (numpy) as the GPU callbacks of pats_amd.pipeline."""

    def __init__(self, nets):
        self.n = nets

    def coarse(self, left, right):
        c = self.n.coarse()
        return cu(c["d0"]), cu(c["d1"]), cu(c["ns"]), float(c["alpha"])

    def fine(self, num, new_left, new_right, mask, sizes=None):
        if num is None:                 # batched mode: the per-chunk tensors, concatenated
            parts = [self.n.fine(c, b) for c, b in enumerate(sizes)]
            return tuple(cu(np.concatenate([p[k] for p in parts])) for k in ("d0", "d1", "scale_x", "scale_y"))
        f = self.n.fine(num, new_left.shape[0])
        return cu(f["d0"]), cu(f["d1"]), cu(f["scale_x"]), cu(f["scale_y"])

    def third(self, num, mk0, mk1, b_ids, sizes=None, count=None):
        if count is not None:           # device-count walk: the tensors are a capacity, the first `count` rows exist
            P, cap = int(count.item()), mk0.shape[0]        # (a host read inside the TEST's network stand-in, not in the path)
            t = self.n.third(num, P) if P > 0 else None
            pad = lambda a, shape: cu(np.concatenate([a, np.ones((cap - P,) + shape, np.float32)]) if a is not None
                                      else np.ones((cap,) + shape, np.float32))
            return (pad(t["d0"] if t else None, (128, 65)), pad(t["d1"] if t else None, (128, 65)),
                    pad(t["scale"] if t else None, (1, 64)))
        if num is None:
            edges = np.cumsum([0] + list(sizes))
            per = np.histogram(b_ids.cpu().numpy(), bins=edges)[0]          # third-level problems per chunk
            parts = [self.n.third(c, int(p)) for c, p in enumerate(per) if p > 0]
            return tuple(cu(np.concatenate([p[k] for p in parts])) for k in ("d0", "d1", "scale"))
        t = self.n.third(num, mk0.shape[0])
        return cu(t["d0"]), cu(t["d1"]), cu(t["scale"])


@pytest.mark.parametrize("batched", [False, True, "device_counts", "device_counts_3_streams"])
@pytest.mark.parametrize("name", ["pipeline_outdoor.npz", "pipeline_indoor.npz", "pipeline_640x480_outdoor.npz",
                                  "pipeline_640x480_indoor.npz"])
def test_pipeline_chain(name, batched):
    """first_layer.py:110-157 -> second_layer.py:100-124 -> pats.py:32-78 -> third_layer.py:153-170 ->
    get_result, on synthetic network outputs: same chunk sizes, same third-level counts, same matches in
    the same order as the reference's own functions produce (tools/make_golden.py::gen_pipeline).
    Source-side coordinates are index arithmetic (exact); target-side ones carry the third-level
    expectation (3e-4 px at 1/2 resolution) times the area scale."""
    from pats_amd import pipeline
    g = golden(name)
    nets = synth.SynthNets(seed=int(g["seed"]), h=int(g["h"]), w=int(g["w"]))
    left, right = [cu(x) for x in nets.images()]
    mode = {}
    if isinstance(batched, str):        # the chunk walk with the counts on the device (pipeline.forward_chunks_device)
        mode, batched = dict(device_counts=True, streams=3 if batched.endswith("streams") else 1), False
    out = pipeline.forward_path(left, right, _CudaNets(nets), if_local=bool(g["if_local"]),
                                if_outdoor=bool(g["if_outdoor"]), merge_new=bool(g["merge_new"]), batch_chunks=batched, **mode)
    if batched:
        assert [c[0] for c in out["chunks"]] == g["chunks"][:, 0].tolist()
    else:
        assert [list(c) for c in out["chunks"]] == g["chunks"].tolist()
    ml, mr = out["matches_l"].cpu().numpy(), out["matches_r"].cpu().numpy()
    assert ml.shape == g["matches_l"].shape and ml.shape[0] > 500
    np.testing.assert_allclose(ml, g["matches_l"], atol=1e-4, rtol=1e-6)
    # target side: the third-level expectation differs from the reference's by <= 3e-4 px on the half-resolution
    # crop (test_third_level), get_result maps it to image pixels with x2 and the crop's scale
    # (<= (736 / 96) = 7.7 for a 640x480 pair): <= 5e-3 px; measured maxima are printed
    d = np.abs(mr - g["matches_r"])
    print("%s batched=%s: %d matches, max |d target| = %.2e px, max |d source| = %.2e px"
          % (name, batched, ml.shape[0], d.max(), np.abs(ml - g["matches_l"]).max()))
    np.testing.assert_allclose(mr, g["matches_r"], atol=6e-3, rtol=1e-6)


def test_chunk_walk_edge_cases_no_match_and_many_chunks(ops):
    """pipeline.forward_chunks_device at its edges: (1) a pair whose coarse level matches nothing returns empty tensors without
    touching the chunk loop (pats.py:27-31), in every mode; (2) a 24 x 32 grid (YFCC shapes: up to 12 chunks) gives the same
    matches in the same order on one, two and three streams as the host-read walk."""
    from pats_amd import pipeline

    class Nets(_CudaNets):
        def __init__(self, nets, kill):
            super().__init__(nets)
            self.kill = kill

        def coarse(self, left, right):
            d0, d1, ns, alpha = super().coarse(left, right)
            if self.kill:                                     # unrelated descriptors: every patch goes to the dustbin
                d1 = torch.randn(d1.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
                d0 = torch.randn(d0.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
            return d0, d1, ns, alpha
    nets = synth.SynthNets(seed=synth.SEED + 77, h=5, w=6)
    left, right = [cu(x) for x in nets.images()]
    for mode in (dict(), dict(batch_chunks=True), dict(device_counts=True), dict(device_counts=True, streams=3)):
        out = pipeline.forward_path(left, right, Nets(nets, True), **mode)
        assert out["matches_l"].shape == (0, 2) and out["matches_r"].shape == (0, 2) and out["chunks"] == [], mode
    big = synth.SynthNets(seed=synth.SEED + 78, h=24, w=32)
    left, right = [cu(x) for x in big.images()]
    want = pipeline.forward_path(left, right, _CudaNets(big))
    assert len(want["chunks"]) >= 8 and want["matches_l"].shape[0] > 5000
    for streams in (1, 2, 3):
        got = pipeline.forward_path(left, right, _CudaNets(big), device_counts=True, streams=streams)
        assert [tuple(c) for c in got["chunks"]] == [tuple(c) for c in want["chunks"]]
        assert torch.equal(got["matches_l"], want["matches_l"]) and torch.equal(got["matches_r"], want["matches_r"]), streams


# ---- index parity at bench-like volumes: thousands of problems, HIP vs the oracle -----------------------
def test_index_parity_4096_third_level_problems(ops, oracle):
    """4 096 third-level problems (the bench solves 25 920 per pair): label, if_matching1, source points
    identical; target points and transport mass within tolerance."""
    inp = synth.third_inputs(seed=synth.SEED + 60, P=4096)
    S = oracle.cost(inp["d0"], inp["d1"])
    Zr = oracle.log_optimal_transport2(S, 1.0, inp["scale"], 100)
    sq = np.sqrt(inp["scale"] + np.float32(1e-8)).astype(np.float32)
    r0, r1, rwl, rlabel, rifm = oracle.compute_result(np.exp(Zr), sq, sq, inp["p_s"], inp["p_t"], True)
    d0, d1, sc = cu(inp["d0"]), cu(inp["d1"]), cu(inp["scale"])
    g0, g1, glabel, gifm, Z = ops.third_level(d0, d1, sc, cu(inp["p_s"]), cu(inp["p_t"]), outdoor=True, return_plan=True)
    f0, f1, flabel, fifm = ops.third_level(d0, d1, sc, cu(inp["p_s"]), cu(inp["p_t"]), outdoor=True)
    for lab, ifm, m0, m1 in ((glabel, gifm, g0, g1), (flabel, fifm, f0, f1)):
        assert np.array_equal(lab.cpu().numpy(), rlabel)
        assert np.array_equal(ifm.cpu().numpy().astype(bool), rifm.astype(bool))
        assert np.array_equal(m0.cpu().numpy(), r0)
        assert np.abs(m1.cpu().numpy() - r1).max() <= 3e-4 * 8       # 3e-4 of the 8-px window
    e, er = np.exp(Z.cpu().numpy().astype(np.float64)), np.exp(Zr.astype(np.float64))
    assert np.abs(e[:, :-1, :-1] - er[:, :-1, :-1]).max() <= MASS_TOL
    np.testing.assert_allclose(e, er, atol=MASS_TOL, rtol=3e-6)
    r, c = ops.argmax(Z)
    wr, wc = oracle.argmax(Zr)
    assert np.array_equal(r.cpu().numpy(), wr) and np.array_equal(c.cpu().numpy(), wc)


def test_index_parity_128_fine_problems(ops, oracle):
    """128 fine-level problems: both argmax vectors, the no-match flags, the expansion rectangles identical."""
    f = synth.fine_inputs(seed=synth.SEED + 61, B=128)
    ns = f["scale_x"] * f["scale_y"]
    Zr = oracle.dustbin_bias(oracle.log_optimal_transport2(oracle.cost(f["d0"], f["d1"]), 1.0, ns, 100), 2.0)
    wr, wc = oracle.argmax(Zr)
    want = oracle.iterative_expand(np.exp(Zr), f["scale_x"], f["scale_y"], 12, 12, 12, 1e-3, 8)
    Z = ops.cost_ot(cu(f["d0"]), cu(f["d1"]), 2, 1.0, cu(ns), 100, bias_k=2.0)
    r, c = ops.argmax(Z)
    assert np.array_equal(r.cpu().numpy(), wr) and np.array_equal(c.cpu().numpy(), wc)
    e, er = np.exp(Z.cpu().numpy().astype(np.float64)), np.exp(Zr.astype(np.float64))
    assert np.abs(e[:, :-1, :-1] - er[:, :-1, :-1]).max() <= MASS_TOL
    pos, rng_ = ops.Compute_positions_and_ranges(12, 12, "cuda")
    got = ops.Iterative_expand_matrix(Z, cu(f["scale_x"]).reshape(128, -1, 1), cu(f["scale_y"]).reshape(128, -1, 1),
                                      [0, 12, 0, 12], rng_, pos, lower_bound=1e-3, iter_num=8, width=12, height=12,
                                      input_is_log=True)
    assert np.array_equal(got[5].cpu().numpy(), want[5])                     # bound: the grown rectangles
    np.testing.assert_allclose(got[0].cpu().numpy(), want[0], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(got[2].cpu().numpy(), want[2], atol=1e-4)
    out = ops.est_position_second(Z, cu(f["scale_x"]), cu(f["scale_y"]), [96, 96], 8)
    assert np.array_equal(out[4].cpu().numpy(), wr[:, :-1] == 144) and np.array_equal(out[5].cpu().numpy(), wc[:, :-1] == 144)


# ---- throughput-mode variants without host reads, error paths, multi-rank plumbing (round 2) ---------
def test_split_patches_device_matches_host(ops):
    """pats_split_patches_device (one thread per pair, no host read) against the host planner and the
    reference's golden plan (utils.py:152-181)."""
    g = golden("coarse_301.npz")
    rng = np.random.default_rng(5)
    flags = [g["ifn1"][0]] + [rng.random(300) < p for p in (0.0, 0.05, 0.5, 0.97, 1.0)]
    sc = torch.cumsum(torch.logical_not(cu(np.stack(flags))).int(), dim=1, dtype=torch.int32)
    for cap in (40, 100, 512):
        num, second, third = ops.split_patches_device(sc, 15, 20, cap)
        num, second, third = num.cpu().numpy(), second.cpu().numpy(), third.cpu().numpy()
        for i in range(len(flags)):
            n, s, t = ops.split_patches(sc[i], 15, 20, cap)
            assert n == num[i] and s == second[i, :n].tolist() and t == third[i, :n].tolist()
            assert not second[i, n:].any() and not third[i, n:].any()
    num, second, third = ops.split_patches_device(sc[:1], 15, 20, 40)
    assert int(num[0]) == int(g["split40_cycle"]) and second[0, :int(num[0])].cpu().tolist() == g["split40_second"].tolist()


def test_compute_imgs_device_counts(ops):
    """known_count="device": same crops in the first K_total rows, no host read, counts on the device."""
    g = golden("coarse_301.npz")
    left, right = synth.image_pair()
    L, R = cu(left), cu(right)
    xs, ys, pts, ifn = cu(g["x_scale"]), cu(g["y_scale"]), cu(g["average_point"]), cu(g["ifn1"])
    ifn_b, ifn_c = ifn.clone(), torch.ones_like(ifn)
    ifn_b[0, ::7] = True
    xs3, ys3, pts3 = torch.cat([xs, xs * 0.9, xs]), torch.cat([ys, ys * 1.1, ys]), torch.cat([pts, pts, pts])
    ifn3, L3, R3 = torch.cat([ifn, ifn_c, ifn_b]), torch.cat([L, L.flip(2), L]), torch.cat([R, R.flip(1), R])
    want = ops.Compute_imgs_ex(xs3, ys3, pts3, ifn3, L3, R3, width=20, height=15)
    got = ops.Compute_imgs_ex(xs3, ys3, pts3, ifn3, L3, R3, width=20, height=15, known_count="device")
    K = want[0].shape[0]
    assert got[0].shape[0] == 3 * 300 and int(got[7].item()) == K
    assert got[6].cpu().tolist() == [int((~ifn3[i]).sum()) for i in range(3)] and got[6][1] == 0
    for k in (0, 1, 5):
        assert torch.equal(got[k][:K], want[k])
    for k in (2, 3, 4):
        assert torch.equal(got[k], want[k])
    # caller-supplied counts are normalised (0-d numpy) and can be validated against the device's
    one = ops.Compute_imgs(xs, ys, pts, ifn, L, R, width=20, height=15, known_count=np.int64(int((~ifn).sum())), validate=True)
    assert torch.equal(one[1], want[1][:one[1].shape[0]])
    with pytest.raises(RuntimeError):
        ops.Compute_imgs(xs, ys, pts, ifn, L, R, width=20, height=15, known_count=3, validate=True)


def test_tensor_resize_raises_like_reference(ops):
    """library.cpp:56-60: narrow() outside the tensor or an empty crop is a c10::Error -> RuntimeError."""
    import tensor_resize
    g = golden("resize_small.npz")
    src = cu(g["src"])
    Hp, Wp = src.shape[2], src.shape[3]
    ok = cu(g["bound"])[:1]
    for bad in ([0, Hp + 1, 0, 10, 0], [5, 5, 0, 10, 0], [0, 10, 4, Wp, 0], [0, 10, 0, 10, 10000 * src.shape[0]],
                [-1, 10, 0, 10, 0]):
        b = torch.cat([ok, torch.tensor([bad], dtype=torch.int64).cuda()])
        with pytest.raises(RuntimeError):
            tensor_resize.tensor_resize(src, b)
        out = ops.tensor_resize(src, b, validate=False)           # no host read: the bad crop is zero-filled
        assert torch.equal(out[0], tensor_resize.tensor_resize(src, ok)[0]) and not out[1].any()


def test_get_result_without_host_read(ops):
    inp = synth.result_inputs(seed=99, h=15, w=20, mixed_choice=True)
    f16, p16 = ops.refine_scatter(cu(inp["ifn2"]), cu(inp["pts"]), cu(inp["mkpts1"]), cu(inp["label0"]))
    sc_rows = inp["sc0"][~inp["ifn0"]]
    args = (1, [cu(inp["ifn0"]), f16], [cu(inp["ap0"]), p16.flip(dims=[2]) / 2.0], [cu(inp["sc0"]), cu(sc_rows)],
            [[32, 15, 20], [2, 48, 48]], [cu(inp["choice0"]), cu(inp["choice1"])])
    ml, mr = ops.get_result(*args)
    mlc, mrc, M = ops.get_result(*args, sync=False)
    assert int(M.item()) == ml.shape[0] and mlc.shape[0] == f16.shape[0] * 2304
    assert torch.equal(mlc[:ml.shape[0]], ml) and torch.equal(mrc[:mr.shape[0]], mr)


def _run_bench(extra_env, gpus, extra_args=()):
    import json
    import subprocess
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "1",
           "--pairs", "2", "--no-cpu-baseline", "--no-secondary"] + list(extra_args)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_other_workloads(ops, sinkhorn_mode):
    """BASELINE configs[2] (ScanNet shapes: indoor rules, one L2 chunk) and configs[3] (YFCC shapes: 769 x 769 coarse problem)
    through the same step as the bench line."""
    if sinkhorn_mode != "kernel":
        pytest.skip("once is enough")
    sc = _run_bench({}, 1, ["--workload", "scannet"])
    assert "configs[2]" in sc["config"]["workload"] and "at most 1 chunks per pair" in sc["config"]["L2"] and sc["matches_per_pair"] > 1000
    yf = _run_bench({}, 1, ["--workload", "yfcc", "--pairs", "1"])
    assert "769x769" in yf["config"]["L1"] and yf["matches_per_pair"] > 1000 and yf["value"] > 0


def test_bench_spawns_its_own_ranks(ops, sinkhorn_mode):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself, shards the pairs and gathers the
    matches to rank 0.  On a 1-GPU box the two ranks share the device and talk over gloo (plumbing only);
    with >= 2 GPUs the same command runs one rank per GPU over RCCL."""
    if sinkhorn_mode != "kernel":
        pytest.skip("once is enough")
    one = _run_bench({}, 1)
    assert one["n_gpus"] == 1 and one["matches_per_pair"] > 1000 and one["roofline"]["frac"] > 0
    if torch.cuda.device_count() >= 2:
        two = _run_bench({}, 2)
    else:
        two = _run_bench({"PATS_BENCH_SHARE_DEVICE": "1", "PATS_BENCH_BACKEND": "gloo"}, 2)
    assert two["n_gpus"] == 2 and two["gather_bytes"] > 0
    assert two["matches_per_pair"] > 1000


def test_bench_under_a_launcher_over_rccl_with_one_rank(ops, sinkhorn_mode):
    """The driver's multi-GPU command line (`python -m torch.distributed.run ... bench.py --gpus N`) with N = 1 and
    PATS_BENCH_FORCE_DIST=1: the process group is initialised over RCCL (backend "nccl") on this one GPU and every collective of
    the N > 1 path runs on it - barrier, all_reduce(MAX), all_gather, shard.gather_matches (all_gather_into_tensor / gather).  A
    1-GPU box cannot host two RCCL ranks; this is as much of the real transport as it can exercise."""
    import json
    import subprocess
    if sinkhorn_mode != "kernel":
        pytest.skip("once is enough")
    env = dict(os.environ, PATS_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PATS_BENCH_BACKEND", "PATS_BENCH_SHARE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--pairs", "2",
           "--no-cpu-baseline", "--no-secondary"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["matches_per_pair"] > 1000 and d["gather_bytes"] > 0
    assert "nccl" in json.dumps(d), "the line should name the backend the gather ran on"


def test_flags_from_the_ot_epilogue_and_the_expansion(ops, oracle):
    """est_position's two flag vectors without an argmax pass (first_layer.py:162-167, second_layer.py:243-248):
    if_nomatching2 from the 145 x 145 Sinkhorn epilogue (or the colmass pass), if_nomatching1 from the expansion
    kernel - identical to the argmax of the returned log-plan, also for problems the guard re-solves."""
    f = synth.fine_inputs(seed=synth.SEED + 62, B=40)
    d0 = f["d0"].copy()
    d0[3] *= 60.0                                               # +-150-nat scores: this problem trips the guard
    ns = cu(f["scale_x"] * f["scale_y"])
    Z, cflag = ops.cost_ot(cu(d0), cu(f["d1"]), 2, 1.0, ns, 100, bias_k=2.0, return_flags=True)
    Zp = ops.cost_ot(cu(d0), cu(f["d1"]), 2, 1.0, ns, 100, bias_k=2.0)
    assert torch.equal(Z, Zp)
    r, c = ops.argmax(Z)
    assert torch.equal(cflag, c[:, :-1] == 144)
    out = ops.est_position_second(Z, cu(f["scale_x"]), cu(f["scale_y"]), [96, 96], 8, col_nomatch=cflag)
    assert torch.equal(out[4], r[:, :-1] == 144) and torch.equal(out[5], c[:, :-1] == 144)
    alone = ops.est_position_second(Z, cu(f["scale_x"]), cu(f["scale_y"]), [96, 96], 8)      # flags from one colmass-style pass
    assert torch.equal(alone[4], out[4]) and torch.equal(alone[5], out[5]) and torch.equal(alone[0], out[0])
    # coarse level: scales and the column flags in one pass
    ci = synth.coarse_inputs()
    Zc = ops.cost_ot(cu(ci["d0"]), cu(ci["d1"]), 1, float(ci["alpha"]), cu(ci["ns"]), 100)
    scales, cf = ops.colmass_sqrt(Zc, return_flags=True)
    assert torch.equal(scales, ops.colmass_sqrt(Zc))
    rc_, cc_ = ops.argmax(Zc)
    assert torch.equal(cf, cc_[:, :-1] == 300)
    o1 = ops.est_position_first(Zc, scales, (480, 640), 32, col_nomatch=cf)
    assert torch.equal(o1[4], rc_[:, :-1] == 300) and torch.equal(o1[5], cf)
    # 65 x 65 (generic pass after the kernel)
    t = synth.third_inputs(seed=5, P=16)
    Z3, c3 = ops.cost_ot(cu(t["d0"]), cu(t["d1"]), 2, 1.0, cu(t["scale"]), 100, return_flags=True)
    assert torch.equal(c3, ops.argmax(Z3)[1][:, :-1] == 64)


def test_third_level_guard_trips_are_resolved_by_the_scan_kernel(ops, oracle, sinkhorn_mode):
    """Fused third level with descriptors scaled so that some problems leave the linear-domain guard band: the
    third-generation kernel flags them (sentinel in if_matching1), the log-domain kernel re-solves exactly those in
    scan mode; tame problems in the same launch are untouched.  Results against the oracle as usual."""
    P = 70                                                      # two scan workgroups (64 + 6 problems)
    inp = synth.third_inputs(seed=synth.SEED + 63, P=P)
    d0, d1 = inp["d0"].copy(), inp["d1"].copy()
    wild = np.array([1, 17, 63, 64, 69])
    d0[wild] *= 9.0                                             # scores of +-100 nats and more
    d1[wild] *= 9.0
    ops.sinkhorn_fallbacks(reset=True)
    m0, m1, label, ifm = ops.third_level(cu(d0), cu(d1), cu(inp["scale"]), cu(inp["p_s"]), cu(inp["p_t"]), outdoor=True)
    trips = ops.sinkhorn_fallbacks(reset=True)
    assert (1 <= trips <= len(wild)) if sinkhorn_mode == "kernel" else trips == 0
    assert set(np.unique(ifm.cpu().numpy().astype(np.uint8))) <= {0, 1}          # no sentinel left behind
    Zr = oracle.log_optimal_transport2(oracle.cost(d0, d1), 1.0, inp["scale"], 100)
    sq = np.sqrt(inp["scale"] + np.float32(1e-8)).astype(np.float32)
    r0, r1, rwl, rlabel, rifm = oracle.compute_result(np.exp(Zr), sq, sq, inp["p_s"], inp["p_t"], True)
    tame = np.setdiff1d(np.arange(P), wild)
    assert np.array_equal(m0.cpu().numpy(), r0)
    assert np.array_equal(label.cpu().numpy().reshape(P, 16, 2)[tame], rlabel.reshape(P, 16, 2)[tame])
    assert np.array_equal(ifm.cpu().numpy().astype(bool)[tame], rifm.astype(bool)[tame])
    assert np.abs(m1.cpu().numpy()[tame] - r1[tame]).max() <= 3e-4 * 8
    # the wild problems: near-degenerate plans (one entry per row carries everything); flags and points where the
    # oracle's top two entries are clearly apart
    S = np.exp(Zr)[wild][:, :-1, :].reshape(len(wild), 8, 8, 65)[:, 2:6, 2:6, :].reshape(len(wild), 16, 65)
    top = np.sort(S, axis=2)[:, :, -2:]
    clear = (top[:, :, 1] - top[:, :, 0]) > 1e-3 * top[:, :, 1]
    assert np.array_equal(ifm.cpu().numpy().astype(bool)[wild][clear], rifm.astype(bool)[wild][clear])
    assert np.isfinite(m1.cpu().numpy()).all()


STAB_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(repo)r)
from pats_amd import ops, synth
P = 600
inp = synth.third_inputs(seed=synth.SEED + 64, P=P)
d0, d1 = inp["d0"].copy(), inp["d1"].copy()
rng = np.random.default_rng(5)
wild = rng.choice(P, 200, replace=False)
amp = rng.choice([3.0, 5.0, 9.0, 14.0], size=200).astype(np.float32)
d0[wild] *= amp[:, None, None]; d1[wild] *= amp[:, None, None]
ops.sinkhorn_fallbacks(reset=True)
cu = lambda a: torch.from_numpy(a).cuda()
m0, m1, label, ifm = ops.third_level(cu(d0), cu(d1), cu(inp["scale"]), cu(inp["p_s"]), cu(inp["p_t"]), outdoor=True)
np.savez(sys.argv[1], m0=m0.cpu().numpy(), m1=m1.cpu().numpy(), label=label.cpu().numpy(), ifm=ifm.cpu().numpy(), wild=wild,
         trips=np.int64(ops.sinkhorn_fallbacks(reset=True)))
"""


def test_stabilised_third_level_resolve_agrees_with_the_log_domain_one(tmp_path):
    """Problems the plain linear solve flags are re-solved by its stabilised instantiation (third_fused3_stab_kernel: scalings
    absorbed into the kernel matrix when they drift) and only what that cannot hold by the log-sum-exp kernel.  Two processes on
    the same inputs - PATS_THIRD_STAB=0 takes every flagged problem to the log-sum-exp kernel as before round 4: same flags and
    labels wherever the plan is not a near-tie, the same points within the parity gate, tame problems bit-identical."""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for stab in ("1", "0"):
        out = str(tmp_path / ("third_stab%s.npz" % stab))
        env = dict(os.environ, PATS_THIRD_STAB=stab)
        r = subprocess.run([sys.executable, "-c", STAB_CHILD % {"repo": repo}, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert int(a["trips"]) == int(b["trips"]) and int(a["trips"]) >= 20           # the same problems leave the band in both
    assert set(np.unique(a["ifm"].astype(np.uint8))) <= {0, 1}                    # no sentinel left behind
    P = a["m0"].shape[0] // 16 if a["m0"].ndim == 2 else 600
    tame = np.setdiff1d(np.arange(600), a["wild"])
    for k in ("m0", "m1", "label", "ifm"):
        x, y = a[k].reshape(600, 16, -1), b[k].reshape(600, 16, -1)
        assert np.array_equal(x[tame], y[tame]), k
    assert np.array_equal(a["m0"], b["m0"])
    same_flag = a["ifm"].reshape(600, 16) == b["ifm"].reshape(600, 16)
    assert same_flag.mean() > 0.995                                                # a flag may flip only on a near-tie
    both = same_flag & (a["ifm"].reshape(600, 16) == 0)
    d = np.abs(a["m1"].reshape(600, 16, 2) - b["m1"].reshape(600, 16, 2)).max(-1)
    assert d[both].max() <= 3e-4 * 8, d[both].max()
    assert np.isfinite(a["m1"]).all()


# ---- AttentionalPropagation / AttentionalGNN around the attention core (section 8f rank 4) -------------------
GNN_CASES = [dict(C=128, b=3, n=65, m=65), dict(C=64, b=2, n=145, m=145), dict(C=32, b=2, n=37, m=53),
             dict(C=264, b=5, n=145, m=145), dict(C=448, b=2, n=300, m=300)]      # 3, 4: the fine and the coarse level's production shapes


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_attentional_propagation_golden(ops, oracle, case):
    """modules.py:107-117 against the reference's own AttentionalPropagation (eval mode and train mode = BatchNorm on
    batch statistics, which is what the third layer runs under PATS.eval) and against the oracle."""
    g = golden("gnn_layer.npz")
    kw = GNN_CASES[case]
    params = synth.gnn_params(seed=synth.SEED + 70 + case, C=kw["C"])
    inp = synth.gnn_inputs(seed=synth.SEED + 80 + case, b=kw["b"], C=kw["C"], n=kw["n"], m=kw["m"])
    P = ops.PropagationParams(params)
    x, s = cu(inp["x"]), cu(inp["source"])
    for mode in ("eval", "train"):
        y = ops.attentional_propagation(x, s, P, bn_train=(mode == "train")).cpu().numpy()
        np.testing.assert_allclose(y.reshape(-1)[g["%s_idx%d" % (mode, case)]], g["%s_val%d" % (mode, case)], atol=5e-5, rtol=2e-4)
        np.testing.assert_allclose(y.astype(np.float64).sum((1, 2)), g["%s_sum%d" % (mode, case)], atol=3e-2, rtol=2e-4)
        want = oracle.attentional_propagation(inp["x"], inp["source"], params, bn_train=(mode == "train"))
        np.testing.assert_allclose(y, want, atol=5e-5, rtol=2e-4)
    assert torch.equal(x, cu(inp["x"]))                                   # inputs are borrowed


def test_attentional_gnn_and_edge_shapes(ops, oracle):
    g = golden("gnn_layer.npz")
    ps = [synth.gnn_params(seed=synth.SEED + 90 + i, C=128) for i in range(2)]
    a = synth.gnn_inputs(seed=synth.SEED + 95, b=4, C=128, n=65)
    d0, d1 = ops.attentional_gnn(cu(a["x"]), cu(a["source"]), [ops.PropagationParams(p) for p in ps], ["self", "cross"])
    np.testing.assert_allclose(d0.cpu().numpy().reshape(-1)[g["gnn_idx"]], g["gnn_d0"], atol=1e-4, rtol=2e-4)
    np.testing.assert_allclose(d1.cpu().numpy().reshape(-1)[g["gnn_idx"]], g["gnn_d1"], atol=1e-4, rtol=2e-4)
    # coarse-level width (448 channels, 300 tokens): three row tiles, ragged last tile, against the oracle
    p = synth.gnn_params(seed=5, C=448)
    i = synth.gnn_inputs(seed=6, b=1, C=448, n=300)
    y = ops.attentional_propagation(cu(i["x"]), cu(i["source"]), ops.PropagationParams(p), residual=cu(i["x"])).cpu().numpy()
    want = oracle.attentional_propagation(i["x"], i["source"], p, residual=i["x"])
    np.testing.assert_allclose(y, want, atol=1e-4, rtol=2e-4)
    # fine-level width 264 = 8 * 33, 145 tokens, many problems per column tile
    p = synth.gnn_params(seed=7, C=264)
    i = synth.gnn_inputs(seed=8, b=5, C=264, n=145)
    y = ops.attentional_propagation(cu(i["x"]), cu(i["source"]), ops.PropagationParams(p), bn_train=True).cpu().numpy()
    np.testing.assert_allclose(y, oracle.attentional_propagation(i["x"], i["source"], p, bn_train=True), atol=1e-4, rtol=2e-4)
    with pytest.raises(RuntimeError):
        ops.attentional_propagation(cu(i["x"]), cu(i["source"][:, :100]), ops.PropagationParams(p))


def test_attentional_gnn_three_layers_at_the_fine_level_shape(ops):
    """AttentionalGNN(264, [self, cross, self]) against the reference's own class (tests/golden/gnn_layer.npz, gnn264_*): the
    stack second_layer.py:89 runs, on the packed fine-level layer kernels."""
    g = golden("gnn_layer.npz")
    ps = [synth.gnn_params(seed=synth.SEED + 96 + i, C=264) for i in range(3)]
    a = synth.gnn_inputs(seed=synth.SEED + 99, b=3, C=264, n=145)
    d0, d1 = ops.attentional_gnn(cu(a["x"]), cu(a["source"]), [ops.PropagationParams(p) for p in ps], ["self", "cross", "self"])
    d0, d1 = d0.cpu().numpy(), d1.cpu().numpy()
    np.testing.assert_allclose(d0.reshape(-1)[g["gnn264_idx"]], g["gnn264_d0"], atol=1e-4, rtol=2e-4)
    np.testing.assert_allclose(d1.reshape(-1)[g["gnn264_idx"]], g["gnn264_d1"], atol=1e-4, rtol=2e-4)
    np.testing.assert_allclose(d0.astype(np.float64).sum((1, 2)), g["gnn264_sum0"], atol=5e-2, rtol=2e-4)
    np.testing.assert_allclose(d1.astype(np.float64).sum((1, 2)), g["gnn264_sum1"], atol=5e-2, rtol=2e-4)


def test_gnn_ten_layers_with_small_magnitude_weights(ops, oracle):
    """The fp16-split contraction carries an operand below 0.004 with a flushed lo half (absolute error <= 2^-20): trained
    Conv1d weights and post-ReLU activations live there.  Ten third-level layers (self / cross alternating, BatchNorm on
    batch statistics like PATS.eval leaves the third layer) with weights of std 1e-3 .. 3e-2 against the oracle, which
    accumulates every dot product in double: the error stays at the 1e-5 level of the descriptors' unit scale and does not
    compound over the layers (the residual path carries the descriptors; the deltas are what the split touches)."""
    rng = np.random.default_rng(77)
    C, b, n = 128, 6, 65
    layers = []
    for k in range(10):
        p = synth.gnn_params(seed=200 + k, C=C)
        scale = (1e-3, 3e-3, 1e-2, 3e-2)[k % 4] * np.sqrt(C)      # gnn_params draws U(-1/sqrt(fan_in), +): rescale to a std
        for name in list(p):
            if name.endswith("weight") and p[name].ndim == 3:
                p[name] = (p[name] * scale).astype(np.float32)
        layers.append(p)
    names = ["self", "cross"] * 5
    x0 = rng.standard_normal((b, C, n)).astype(np.float32)
    x1 = rng.standard_normal((b, C, n)).astype(np.float32)
    g0, g1 = ops.attentional_gnn(cu(x0), cu(x1), [ops.PropagationParams(p) for p in layers], names, bn_train=True)
    w0, w1 = x0, x1
    for p, name in zip(layers, names):
        s0, s1 = (w1, w0) if name == "cross" else (w0, w1)
        n0 = oracle.attentional_propagation(w0, s0, p, bn_train=True, residual=w0)
        n1 = oracle.attentional_propagation(w1, s1, p, bn_train=True, residual=w1)
        w0, w1 = n0, n1
    err = max(np.abs(g0.cpu().numpy() - w0).max(), np.abs(g1.cpu().numpy() - w1).max())
    moved = np.abs(w0 - x0).max()
    print("ten layers, small weights: max |error| = %.2e on descriptors of scale 1 (the layers moved them by up to %.2e)" % (err, moved))
    assert moved > 1e-3 and err <= 2e-5


# ---- the descriptor heads: KeypointEncoder (modules.py:70-82) and final_proj (first_layer.py:34-36,105) ----------------
@pytest.mark.parametrize("tag,dim,h,w,seed", [("third", 128, 8, 8, synth.SEED + 100), ("first", 448, 15, 20, synth.SEED + 101)])
def test_keypoint_encoder_golden(ops, oracle, tag, dim, h, w, seed):
    """Six Conv1d layers with BatchNorm + ReLU between them (eval: running statistics; train: batch statistics - the third
    layer's mode under PATS.eval when if_local is off), against the reference's own class and against the oracle."""
    g = golden("heads.npz")
    params = synth.kenc_params(seed=seed, feature_dim=dim)
    P = ops.MLPParams(params, prefix="encoder.")
    assert [tuple(l["weight"].shape[:2]) for l in P.layers] == [(32, 2), (64, 32), (128, 64), (256, 128), (512, 256), (dim, 512)]
    kpts = cu(synth.grid_kpts(h, w))
    for mode in ("eval", "train"):
        y = ops.keypoint_encoder(kpts, P, bn_train=(mode == "train")).cpu().numpy()
        assert y.shape == (1, dim, h * w)
        want = oracle.keypoint_encoder(synth.grid_kpts(h, w), params, bn_train=(mode == "train"))
        np.testing.assert_allclose(y, want, atol=3e-5, rtol=2e-4)
        if tag == "third":
            np.testing.assert_allclose(y, g["kenc_third_%s" % mode], atol=3e-5, rtol=2e-4)
        else:
            np.testing.assert_allclose(y.reshape(-1)[g["kenc_first_idx"]], g["kenc_first_%s" % mode], atol=3e-5, rtol=2e-4)


@pytest.mark.parametrize("tag,C,b,n,seed", [("first", 448, 1, 300, synth.SEED + 110), ("second", 264, 6, 145, synth.SEED + 111)])
def test_final_proj_golden(ops, oracle, tag, C, b, n, seed):
    g = golden("heads.npz")
    p = synth.final_proj_params(seed=seed, C=C)
    x = synth.gnn_inputs(seed=seed + 5, b=b, C=C, n=n)["x"]
    y = ops.conv1d(cu(x), cu(p["weight"]), cu(p["bias"])).cpu().numpy()
    np.testing.assert_allclose(y.reshape(-1)[g["proj_%s_idx" % tag]], g["proj_%s_val" % tag], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(y, oracle.conv1d(x, p["weight"], p["bias"]), atol=2e-5, rtol=1e-4)
    # what feeds the cost build: cost(final_proj(d0), final_proj(d1)) as first_layer.py:105-111 chains them
    S = ops.cost(cu(y), cu(y)).cpu().numpy()
    np.testing.assert_allclose(S, oracle.cost(y, y), atol=3e-5, rtol=1e-5)


SCALE_CASES = [("first", 448, 2, 15, 20, 1, False, synth.SEED + 120), ("second", 264, 6, 12, 12, 2, True, synth.SEED + 121),
               ("third", 128, 40, 8, 8, 1, True, synth.SEED + 122)]


@pytest.mark.parametrize("tag,C,b,h,w,heads,dust,seed", SCALE_CASES)
def test_scale_head_golden(ops, oracle, tag, C, b, h, w, heads, dust, seed):
    """The 3 x 3 stencil + sigmoid / exp that makes `ns` (first_layer.py:106-107, second_layer.py:92-98, third_layer.py:151-152),
    against the reference expression on nn.Conv2d and against the oracle; its result is what the OT takes."""
    g = golden("heads.npz")
    ws, bs = synth.scale_head_params(seed=seed, C=C, heads=heads)
    x = (4.0 * synth.gnn_inputs(seed=seed + 5, b=b, C=C, n=h * w + int(dust))["x"]).astype(np.float32)
    y = ops.scale_head(cu(x), h, w, [cu(v) for v in ws], [cu(v) for v in bs])
    assert y.shape == (b, 1, h * w)
    np.testing.assert_allclose(y.cpu().numpy(), g["scale_%s" % tag], rtol=3e-5)
    np.testing.assert_allclose(y.cpu().numpy(), oracle.scale_head(x, h, w, ws, bs), rtol=3e-5)
    y2, per = ops.scale_head(cu(x), h, w, [cu(v) for v in ws], [cu(v) for v in bs], return_heads=True)
    assert torch.equal(y2, y) and len(per) == heads
    for i in range(heads):      # scale_x, scale_y on their own (second_layer.py:92-97)
        np.testing.assert_allclose(per[i].cpu().numpy(), oracle.scale_head(x, h, w, ws[i:i + 1], bs[i:i + 1]), rtol=3e-5)
    if tag == "third":          # straight into the solver, as third_layer.py:157-158 does
        S = ops.cost(cu(x), cu(x))
        Z = ops.log_optimal_transport2(S, 1.0, y, 100)
        _check_marginals(Z, y, float(h * w))


def test_scale_head_edge_cases(ops, oracle):
    rng = np.random.default_rng(9)
    for (b, C, h, w, ld, heads) in [(3, 5, 1, 1, 1, 1), (2, 17, 1, 7, 9, 2), (1, 40, 16, 32, 512, 1), (5, 33, 3, 2, 6, 2)]:
        x = rng.standard_normal((b, C, ld)).astype(np.float32)
        ws = [rng.standard_normal((1, C, 3, 3)).astype(np.float32) * 0.2 for _ in range(heads)]
        bs = [rng.standard_normal(1).astype(np.float32) for _ in range(heads)]
        y = ops.scale_head(cu(x), h, w, [cu(v) for v in ws], [cu(v) for v in bs]).cpu().numpy()
        np.testing.assert_allclose(y, oracle.scale_head(x, h, w, ws, bs), rtol=3e-5)
    assert ops.scale_head(cu(x[:0]), 3, 2, [cu(v) for v in ws], [cu(v) for v in bs]).shape == (0, 1, 6)
    with pytest.raises(RuntimeError):
        ops.scale_head(cu(x), 3, 3, [cu(v) for v in ws], [cu(v) for v in bs])           # 9 cells > ld = 6
    with pytest.raises(RuntimeError):
        ops.scale_head(cu(np.zeros((1, 4, 600), np.float32)), 20, 30, cu(np.zeros((1, 4, 3, 3), np.float32)), cu(np.zeros(1, np.float32)))


def _oracle_gnn(oracle, d0, d1, layers, names, bn_train=False):
    for p, name in zip(layers, names):
        s0, s1 = (d1, d0) if name == "cross" else (d0, d1)
        d0, d1 = (oracle.attentional_propagation(d0, s0, p, bn_train=bn_train, residual=d0),
                  oracle.attentional_propagation(d1, s1, p, bn_train=bn_train, residual=d1))
    return d0, d1


def test_layer_heads_against_the_oracle_chain(ops, oracle):
    """pats_amd.heads: everything a layer computes between its backbone and its OT problem (first_layer.py:74-107,
    second_layer.py:71-97, third_layer.py:121-152), each against the same composition of oracle functions - and the coarse
    one on into the solver."""
    from pats_amd import heads
    names = ["self", "cross"]
    dev = lambda t: [cu(v) for v in t] if isinstance(t, (list, tuple)) else cu(t)
    # ---- coarse: 448 channels, 5 x 6 grid -------------------------------------------------------------------------------
    h, w, C = 5, 6, 448
    kp = synth.kenc_params(seed=1, feature_dim=C)
    gp = [synth.gnn_params(seed=2 + i, C=C) for i in range(2)]
    fp = synth.final_proj_params(seed=5, C=C)
    (sw,), (sb,) = synth.scale_head_params(seed=6, C=C)
    rng = np.random.default_rng(7)
    dl = rng.standard_normal((1, C, h, w)).astype(np.float32); dr = rng.standard_normal((1, C, h, w)).astype(np.float32)
    H1 = heads.CoarseHeads(ops.MLPParams(kp, prefix="encoder."), [ops.PropagationParams(p) for p in gp], names,
                           (cu(fp["weight"]), cu(fp["bias"])), (cu(sw), cu(sb)), bin_score=-0.25)
    m0, m1, scale, alpha = H1(cu(dl), cu(dr))
    k = oracle.keypoint_encoder(synth.grid_kpts(h, w), kp)
    assert np.array_equal(heads.grid_kpts(h, w, "cuda").cpu().numpy(), synth.grid_kpts(h, w))
    o0, o1 = _oracle_gnn(oracle, dl.reshape(1, C, -1) + k, dr.reshape(1, C, -1) + k, gp, names)
    w0, w1 = oracle.conv1d(o0, fp["weight"], fp["bias"]), oracle.conv1d(o1, fp["weight"], fp["bias"])
    np.testing.assert_allclose(m0.cpu().numpy(), w0, atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(m1.cpu().numpy(), w1, atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(scale.cpu().numpy(), oracle.scale_head(w1, h, w, [sw], [sb]), rtol=2e-3)
    assert alpha == 0.25
    Z = ops.cost_ot(m0, m1, 1, alpha, scale, 100)                       # first_layer.py:110-115 on the heads' outputs
    _check_marginals(Z, scale, float(h * w))
    # ---- fine: the sampled descriptors of three crops ---------------------------------------------------------------------
    fm = synth.fine_maps(seed=8, B=3)
    gp2 = [synth.gnn_params(seed=9 + i, C=264) for i in range(2)]
    fp2 = synth.final_proj_params(seed=12, C=264)
    (sxw, syw), (sxb, syb) = synth.scale_head_params(seed=13, C=264, heads=2)
    H2 = heads.FineHeads([ops.PropagationParams(p) for p in gp2], names, (cu(fp2["weight"]), cu(fp2["bias"])),
                         (cu(sxw), cu(sxb)), (cu(syw), cu(syb)))
    m0, m1, sx, sy = H2([cu(fm["f0"]), cu(fm["f1"]), cu(fm["f2"])], cu(fm["title"]), cu(fm["rubbish"]))
    d = oracle.fine_descriptors(fm["f0"], fm["f1"], fm["f2"], fm["title"], fm["rubbish"])
    o0, o1 = _oracle_gnn(oracle, d[0], d[1], gp2, names)
    w0, w1 = oracle.conv1d(o0, fp2["weight"], fp2["bias"]), oracle.conv1d(o1, fp2["weight"], fp2["bias"])
    np.testing.assert_allclose(m0.cpu().numpy(), w0, atol=3e-4, rtol=3e-4)
    np.testing.assert_allclose(m1.cpu().numpy(), w1, atol=3e-4, rtol=3e-4)
    np.testing.assert_allclose(sx.cpu().numpy(), oracle.scale_head(w1, 12, 12, [sxw], [sxb]), rtol=2e-3)
    np.testing.assert_allclose(sy.cpu().numpy(), oracle.scale_head(w1, 12, 12, [syw], [syb]), rtol=2e-3)
    # ---- third: window gather + KeypointEncoder, GNN on batch statistics, scale head ----------------------------------------
    tm = synth.third_maps(seed=14, B=3, P=40)
    kp3 = synth.kenc_params(seed=15, feature_dim=128)
    gp3 = [synth.gnn_params(seed=16 + i, C=128) for i in range(2)]
    (s3w,), (s3b,) = synth.scale_head_params(seed=19, C=128)
    for train in (False, True):
        H3 = heads.ThirdHeads(ops.MLPParams(kp3, prefix="encoder."), [ops.PropagationParams(p) for p in gp3], names,
                              (cu(s3w), cu(s3b)), bn_train=train)
        f0, f1, sc, ps, pt = H3(cu(tm["ff0"]), cu(tm["ff1"]), cu(tm["mk0"]), cu(tm["mk1"]), cu(tm["b_ids"]), cu(tm["rubbish"]))
        k3 = oracle.keypoint_encoder(synth.grid_kpts(8, 8), kp3, bn_train=train)
        u0, u1, ops_, opt = oracle.third_descriptors(tm["ff0"], tm["ff1"], tm["mk0"], tm["mk1"], tm["b_ids"], k3, tm["rubbish"])
        assert np.array_equal(ps.cpu().numpy(), ops_) and np.array_equal(pt.cpu().numpy(), opt)
        o0, o1 = _oracle_gnn(oracle, u0, u1, gp3, names, bn_train=train)
        np.testing.assert_allclose(f0.cpu().numpy(), o0, atol=3e-4, rtol=3e-4)
        np.testing.assert_allclose(f1.cpu().numpy(), o1, atol=3e-4, rtol=3e-4)
        np.testing.assert_allclose(sc.cpu().numpy(), oracle.scale_head(o1, 8, 8, [s3w], [s3b]), rtol=2e-3)


def test_weights_stationary_conv_matches_the_lean_tile(ops, oracle):
    """csrc/gnn.hip conv_ws_kernel (the third level's 128-channel products: weights split once per workgroup, activations
    streamed) against the oracle, and bit for bit against the lean tile it replaces there (PATS_CONV_WS=0) - subprocesses,
    the switch is read once per process."""
    import subprocess
    import tempfile
    code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
from pats_amd import ops, synth
import pats_oracle as oracle
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out = {}
# one whole layer at the third level's width, small batch (PATS_CONV_WS=2 takes the kernel at any size) against the oracle
p = synth.gnn_params(seed=3, C=128); i = synth.gnn_inputs(seed=4, b=7, C=128, n=65)
P = ops.PropagationParams(p)
for train in (False, True):
    y = ops.attentional_propagation(cu(i["x"]), cu(i["source"]), P, bn_train=train, residual=cu(i["x"])).cpu().numpy()
    want = oracle.attentional_propagation(i["x"], i["source"], p, bn_train=train, residual=i["x"])
    np.testing.assert_allclose(y, want, atol=1e-4, rtol=2e-4)
    out["layer%%d" %% train] = y
# a size over the kernel's own threshold with a ragged last tile, folded affine on the input, bias / residual variants, M < 128
rng = np.random.default_rng(5)
x = rng.standard_normal((4100, 128, 65)).astype(np.float32)            # 266 500 columns
w = (rng.standard_normal((128, 128, 1)) / 11).astype(np.float32); bias = rng.standard_normal(128).astype(np.float32)
sc = rng.uniform(0.5, 1.5, 128).astype(np.float32); sh = rng.standard_normal(128).astype(np.float32)
res = rng.standard_normal((4100, 128, 65)).astype(np.float32)
y = ops.conv1d(cu(x), cu(w), cu(bias), cu(sc), cu(sh), cu(res)).cpu().numpy()
xa = np.maximum(x[-40:] * sc[None, :, None] + sh[None, :, None], 0).astype(np.float32)
np.testing.assert_allclose(y[-40:], oracle.conv1d(xa, w, bias) + res[-40:], atol=5e-5, rtol=2e-4)
out["conv"] = y[::97]
y2 = ops.conv1d(cu(x), cu(w[:72]), None).cpu().numpy()                # 72 output rows: a partial row tile
np.testing.assert_allclose(y2[:40], oracle.conv1d(x[:40], w[:72]), atol=5e-5, rtol=2e-4)
np.testing.assert_allclose(y2[-40:], oracle.conv1d(x[-40:], w[:72]), atol=5e-5, rtol=2e-4)
out["conv72"] = y2[::97]
# operands beyond the fp16 range: the redo launch recomputes the product
xb = x[:300].copy(); xb[5, 17, 3] = 3.0e4
yb = ops.conv1d(cu(xb), cu(w), cu(bias)).cpu().numpy()
assert np.isfinite(yb).all()
np.testing.assert_allclose(yb[:8], oracle.conv1d(xb[:8], w, bias), atol=2e-2, rtol=2e-4)
np.savez(sys.argv[1], **out)
print("OK")
''' % (REPO, REPO)
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for mode in ("2", "0"):
            path = os.path.join(d, "o%s.npz" % mode)
            p = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, PATS_CONV_WS=mode), capture_output=True,
                               text=True, timeout=900)
            assert p.returncode == 0 and "OK" in p.stdout, mode + ": " + p.stdout[-500:] + p.stderr[-2000:]
            res[mode] = dict(np.load(path))
    for k in res["2"]:
        assert np.array_equal(res["2"][k], res["0"][k]), k          # same split, same MFMA order: the same bits


def test_merge_folded_into_mlp0_against_the_two_product_form(ops, oracle):
    """Round 4: pats_propagation_pack_f32 folds the merge Conv1d into mlp[0] (W1m Wm and W1m bm + b1 formed once per layer in
    double, csrc/gnn_fused.hip gnn_fold_kernel); PATS_GNN_FOLD=0 keeps modules.py:104,116 as two products.  Both forms against
    the oracle (which runs the two products), and against each other at fp32 rounding level - third-level shape (fused kernel,
    eval and batch statistics), fine-level shape (conv_pk path) and a small ragged one.  Subprocesses: the switch is read once."""
    import subprocess
    import tempfile
    code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
from pats_amd import ops, synth
import pats_oracle as oracle
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out = {}
for tag, C, b, n in (("third", 128, 9, 65), ("fine", 264, 5, 145), ("small", 64, 3, 37)):
    p = synth.gnn_params(seed=11 + C, C=C); i = synth.gnn_inputs(seed=12 + C, b=b, C=C, n=n)
    P = ops.PropagationParams(p)
    for train in (False, True):
        y = ops.attentional_propagation(cu(i["x"]), cu(i["source"]), P, bn_train=train, residual=cu(i["x"])).cpu().numpy()
        want = oracle.attentional_propagation(i["x"], i["source"], p, bn_train=train, residual=i["x"])
        np.testing.assert_allclose(y, want, atol=1e-4, rtol=2e-4)
        out["%%s%%d" %% (tag, train)] = y
np.savez(sys.argv[1], **out)
print("OK")
''' % (REPO, REPO)
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for mode in ("1", "0"):
            path = os.path.join(d, "o%s.npz" % mode)
            p = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, PATS_GNN_FOLD=mode), capture_output=True,
                               text=True, timeout=900)
            assert p.returncode == 0 and "OK" in p.stdout, mode + ": " + p.stdout[-500:] + p.stderr[-2000:]
            res[mode] = dict(np.load(path))
    for k in res["1"]:
        assert not np.array_equal(res["1"][k], res["0"][k]), k         # the switch took effect: a different rounding somewhere
        np.testing.assert_allclose(res["1"][k], res["0"][k], atol=2e-5, rtol=2e-5, err_msg=k)


W2_CHILD = r"""
import sys, os, numpy as np, torch
sys.path.insert(0, %(repo)r)
from pats_amd import ops
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(91)
B = 256
out = {}
# (a) log_optimal_transport2: tame scores, three biases, 1 / 3 / 100 sweeps; a few problems far outside the guard band and one
#     with -inf entries (both leave through the fail flag and the re-solve)
S = (2.0 * rng.standard_normal((B, 145, 145))).astype(np.float32)
S[5] *= 70.0
S[77] *= 90.0
S[130, 3, 40:60] = -np.inf
ns = np.exp(0.4 * rng.standard_normal((B, 1, 144))).astype(np.float32)
ops.sinkhorn_fallbacks(reset=True)
for bias in (0.0, 2.0, 3.0):
    for iters in (1, 3, 100):
        out["ot2_b%%g_i%%d" %% (bias, iters)] = ops.log_optimal_transport2(cu(S), 1.0, cu(ns), iters, bias_k=bias).cpu().numpy()
out["trips"] = np.array([ops.sinkhorn_fallbacks(reset=True)])
# (b) given marginals (log_sinkhorn_iterations): random positive marginals of equal mass
mu = rng.uniform(0.2, 2.0, (B, 145)); nu = rng.uniform(0.2, 2.0, (B, 145))
mu /= mu.sum(1, keepdims=True); nu /= nu.sum(1, keepdims=True)
out["given"] = ops.log_sinkhorn_iterations(cu(S[:64]), cu(np.log(mu[:64]).astype(np.float32)), cu(np.log(nu[:64]).astype(np.float32)), 100).cpu().numpy()
# (c) cost + OT with the column flags, over a capacity with the count on the device (padding rows untouched)
base = rng.standard_normal((B, 264, 145)).astype(np.float32)
d0 = (3.0 * (base + 0.3 * rng.standard_normal((B, 264, 145)))).astype(np.float32)
d1 = (3.0 * (base + 0.3 * rng.standard_normal((B, 264, 145)))).astype(np.float32)
Z, fl = ops.cost_ot(cu(d0), cu(d1), 2, 1.0, cu(ns), 100, bias_k=2.0, return_flags=True)
out["cost_ot"] = Z.cpu().numpy(); out["flags"] = fl.cpu().numpy()
cnt = torch.tensor([200], device="cuda", dtype=torch.int64)
Zc, fc = ops.cost_ot(cu(d0), cu(d1), 2, 1.0, cu(ns), 100, bias_k=2.0, return_flags=True, count=cnt)
assert torch.equal(Zc[:200], Z[:200]) and torch.equal(fc[:200], fl[:200])
assert torch.equal(fl, Z[:, -1, :-1] > Z[:, :-1, :-1].max(1).values)          # second_layer.py:243,248 on the plan's own values
np.savez(sys.argv[1], **out)
print("OK")
"""


def test_two_wave_fine_solver_against_the_four_wave_one(ops, oracle, sinkhorn_mode):
    """csrc/sinkhorn_blk2w.hip (two waves per 145 x 145 problem, the default) against csrc/sinkhorn_blk.hip (four waves,
    PATS_FINE_W2=0) in two child processes: log_optimal_transport2 at three biases and 1 / 3 / 100 sweeps (after ONE sweep a
    wrong hand-over shows at once: the bug this kernel once had was invisible after 100), given marginals, guard trips and -inf
    entries, cost + OT with column flags over a counted capacity.  Plans under the mass gate of the parity tests, the same
    problems re-solved, flags equal up to near-ties."""
    import subprocess
    import tempfile
    if sinkhorn_mode != "kernel":
        pytest.skip("the forced log domain runs neither kernel")
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for w2 in ("0", "1"):
            path = os.path.join(d, "w%s.npz" % w2)
            p = subprocess.run([sys.executable, "-c", W2_CHILD % {"repo": REPO}, path], env=dict(os.environ, PATS_FINE_W2=w2),
                               capture_output=True, text=True, timeout=900)
            assert p.returncode == 0 and "OK" in p.stdout, "PATS_FINE_W2=" + w2 + ": " + p.stdout[-500:] + p.stderr[-2500:]
            res[w2] = dict(np.load(path))
    assert int(res["0"]["trips"][0]) == int(res["1"]["trips"][0]) >= 6              # the same problems leave the guard band (two, at 100 sweeps, per bias)
    for k in res["0"]:
        if k in ("trips", "flags"):
            continue
        a, b = res["0"][k].astype(np.float64), res["1"][k].astype(np.float64)
        assert np.array_equal(np.isneginf(a), np.isneginf(b)), k
        fin = np.isfinite(a)
        np.testing.assert_allclose(np.exp(b[fin]), np.exp(a[fin]), atol=1e-4, rtol=5e-6, err_msg=k)
    assert float((res["0"]["flags"] != res["1"]["flags"]).mean()) < 1e-4             # a flag may flip only on a near-tie
    # and one slice against the oracle
    rng = np.random.default_rng(91)
    S = (2.0 * rng.standard_normal((256, 145, 145))).astype(np.float32)
    ns = np.exp(0.4 * rng.standard_normal((256, 1, 144))).astype(np.float32)
    want = oracle.log_optimal_transport2(S[:4], 1.0, ns[:4], 3)
    np.testing.assert_allclose(np.exp(res["1"]["ot2_b0_i3"][:4].astype(np.float64)), np.exp(want.astype(np.float64)), atol=1e-4, rtol=5e-6)


def test_conv1d_edge_cases(ops, oracle):
    rng = np.random.default_rng(4)
    # no bias, ragged channel counts (K = 5 is padded to 8 inside), residual, folded input affine + ReLU
    x = rng.standard_normal((3, 5, 77)).astype(np.float32)
    w = rng.standard_normal((13, 5, 1)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 5).astype(np.float32); sh = rng.standard_normal(5).astype(np.float32)
    res = rng.standard_normal((3, 13, 77)).astype(np.float32)
    y = ops.conv1d(cu(x), cu(w), None, cu(sc), cu(sh), cu(res)).cpu().numpy()
    xa = np.maximum(x * sc[None, :, None] + sh[None, :, None], 0).astype(np.float32)
    np.testing.assert_allclose(y, oracle.conv1d(xa, w) + res, atol=2e-5, rtol=1e-4)
    assert ops.conv1d(cu(x[:0]), cu(w)).shape == (0, 13, 77)
    with pytest.raises(RuntimeError):
        ops.conv1d(cu(x), cu(w[:, :4]))
    with pytest.raises(RuntimeError):
        ops.conv1d(torch.from_numpy(x), cu(w))            # CPU tensor: no fallback
    # batch statistics, folded: against numpy
    h = rng.standard_normal((7, 24, 50)).astype(np.float32) * 3 + 1
    gam = rng.uniform(0.5, 1.5, 24).astype(np.float32); bet = rng.standard_normal(24).astype(np.float32)
    s, t = ops.bn_fold(cu(h), cu(gam), cu(bet), 1e-5)
    mean = h.astype(np.float64).mean((0, 2)); var = h.astype(np.float64).var((0, 2))
    np.testing.assert_allclose(s.cpu().numpy(), gam / np.sqrt(var + 1e-5), rtol=1e-5)
    np.testing.assert_allclose(t.cpu().numpy(), bet - mean * gam / np.sqrt(var + 1e-5), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("C,b,n,m", [(8, 37, 3, 5), (24, 9, 70, 33), (72, 4, 161, 20), (136, 3, 65, 65), (160, 3, 100, 160), (320, 2, 160, 97),
                                     (296, 2, 129, 145)])
def test_attentional_propagation_small_widths_against_oracle(ops, oracle, C, b, n, m):
    """Single-chunk and ragged reductions (K = 8 .. 272, the two-pass reduction of the MLP's first product with a
    zero-filled tail in each pass), column tiles that cover many problems, train and eval BatchNorm, residual.  The last three
    are attention145_kernel's corners (40 / 80 / 74 channels per head, 97 .. 160 tokens, n != m) and, for the packed-weights
    convolutions, 10 / 20 / 19 row tiles and reductions of 160 .. 640 channels in one to three passes."""
    params = synth.gnn_params(seed=100 + C, C=C)
    inp = synth.gnn_inputs(seed=200 + C, b=b, C=C, n=n, m=m)
    P = ops.PropagationParams(params)
    x, s = cu(inp["x"]), cu(inp["source"])
    for train in (False, True):
        y = ops.attentional_propagation(x, s, P, bn_train=train, residual=x).cpu().numpy()
        want = oracle.attentional_propagation(inp["x"], inp["source"], params, bn_train=train, residual=inp["x"])
        np.testing.assert_allclose(y, want, atol=1e-4, rtol=2e-4)


def test_fine_level_layer_operands_beyond_the_fp16_range(ops, oracle):
    """The fine level's shape runs conv_pk_kernel and attention145_kernel (fp16-split operands, no fp32 path inside): spikes in the
    descriptors make their outputs non-finite, they raise their flags, and the fp32-capable kernels queued behind redo the launch."""
    params = synth.gnn_params(seed=21, C=264)
    inp = synth.gnn_inputs(seed=22, b=3, C=264, n=145)
    inp["x"][1, 100, 77] = 70000.0                 # beyond fp16 even before the 2^6 prescale: q overflows, then the attention's operands
    inp["source"][2, 5, 144] = -2500.0
    y = ops.attentional_propagation(cu(inp["x"]), cu(inp["source"]), ops.PropagationParams(params), residual=cu(inp["x"])).cpu().numpy()
    want = oracle.attentional_propagation(inp["x"], inp["source"], params, residual=inp["x"])
    assert np.isfinite(y).all()
    np.testing.assert_allclose(y, want, atol=2e-2, rtol=5e-4)          # outputs reach 1e4 next to the spikes


def test_attentional_propagation_operands_beyond_the_fp16_range(ops, oracle):
    """The 1x1 convolutions use the fp16-split contraction; an activation beyond +-1023 sends the workgroups that read it
    through the fp32 redo, and the layer still matches the oracle."""
    params = synth.gnn_params(seed=11, C=128)
    inp = synth.gnn_inputs(seed=12, b=6, C=128, n=65)
    inp["x"][2, 17, 40] = 3000.0
    inp["source"][4, 90, 3] = -2500.0
    y = ops.attentional_propagation(cu(inp["x"]), cu(inp["source"]), ops.PropagationParams(params)).cpu().numpy()
    want = oracle.attentional_propagation(inp["x"], inp["source"], params)
    assert np.isfinite(y).all()
    np.testing.assert_allclose(y, want, atol=5e-3, rtol=5e-4)          # outputs reach 1e3 next to the spikes


def test_dropin_runs_a_gnn_module_on_the_hip_kernels(ops):
    """pats_amd.dropin.install() on a module tree SHAPED like the reference's models/modules.py classes: containers built from
    torch.nn primitives with the reference's attribute names and NO forward of their own (the reference does not travel to the
    GPU box, and restating its forward here would be a copy) - after install() the instances run AttentionalGNN /
    AttentionalPropagation / KeypointEncoder on the HIP kernels.  Expected values: tests/golden/dropin_gnn.npz, produced by
    the reference's own classes (tools/make_golden.py::gen_dropin) from the same pats_amd.synth parameters - eval mode, train
    mode (BatchNorm on batch statistics), a checkpoint loaded after the first forward, an in-place weight update."""
    import types
    import torch.nn as nn
    from pats_amd import dropin
    g = golden("dropin_gnn.npz")
    C, names = 128, ["self", "cross", "self"]

    class ShapeOnly(nn.Module):
        def forward(self, *a, **k):
            raise NotImplementedError("shape-only test double: dropin.install() supplies the forward")

    class MultiHeadedAttention(ShapeOnly):
        def __init__(self, num_heads, d_model):
            super().__init__()
            self.dim, self.num_heads = d_model // num_heads, num_heads
            self.merge = nn.Conv1d(d_model, d_model, kernel_size=1)
            self.proj = nn.ModuleList([nn.Conv1d(d_model, d_model, kernel_size=1) for _ in range(3)])

    class AttentionalPropagation(ShapeOnly):
        def __init__(self, feature_dim, num_heads):
            super().__init__()
            self.attn = MultiHeadedAttention(num_heads, feature_dim)
            self.mlp = nn.Sequential(nn.Conv1d(feature_dim * 2, feature_dim * 2, 1), nn.BatchNorm1d(feature_dim * 2), nn.ReLU(),
                                     nn.Conv1d(feature_dim * 2, feature_dim, 1))

    class AttentionalGNN(ShapeOnly):
        def __init__(self, feature_dim, layer_names):
            super().__init__()
            self.layers = nn.ModuleList([AttentionalPropagation(feature_dim, 4) for _ in layer_names])
            self.names = layer_names

    class KeypointEncoder(ShapeOnly):
        def __init__(self, feature_dim, layers):
            super().__init__()
            ch, seq = [2] + layers + [feature_dim], []
            for i in range(1, len(ch)):
                seq.append(nn.Conv1d(ch[i - 1], ch[i], kernel_size=1, bias=True))
                if i < len(ch) - 1:
                    seq += [nn.BatchNorm1d(ch[i]), nn.ReLU()]
            self.encoder = nn.Sequential(*seq)

    def load(gnn, plist):
        for lyr, p_ in zip(gnn.layers, plist):
            lyr.load_state_dict({k: torch.from_numpy(v) for k, v in p_.items()}, strict=False)

    def check(tag, got, atol=2e-4):
        idx = torch.from_numpy(g["idx"]).cuda()
        for k, y in zip(("_d0", "_d1"), got):
            np.testing.assert_allclose(y.reshape(-1)[idx].cpu().numpy(), g[tag + k], atol=atol, rtol=2e-4)
        np.testing.assert_allclose([float(got[0].double().sum()), float(got[1].double().sum())], g[tag + "_sum"], atol=0.5, rtol=2e-4)

    ps = [synth.gnn_params(seed=synth.SEED + 120 + i, C=C) for i in range(3)]
    other = [synth.gnn_params(seed=synth.SEED + 130 + i, C=C) for i in range(3)]
    a = synth.gnn_inputs(seed=synth.SEED + 125, b=6, C=C, n=65)
    kp = synth.kenc_params(seed=synth.SEED + 140, feature_dim=C)
    assert synth.checksum(a["x"], a["source"], ps[0]["mlp.0.weight"], other[2]["mlp.3.weight"], kp["encoder.0.weight"]) == \
        pytest.approx(float(g["in_checksum"]), rel=1e-9)
    saved = {n: sys.modules.get(n) for n in ("models", "models.modules")}
    mod = types.ModuleType("models.modules")
    mod.AttentionalPropagation, mod.AttentionalGNN, mod.KeypointEncoder = AttentionalPropagation, AttentionalGNN, KeypointEncoder
    sys.modules["models"], sys.modules["models.modules"] = types.ModuleType("models"), mod
    try:
        gnn = AttentionalGNN(C, names).cuda()
        load(gnn, ps)
        kenc = KeypointEncoder(C, [32, 64, 128, 256, 512]).cuda()
        kenc.load_state_dict({k: torch.from_numpy(v) for k, v in kp.items()}, strict=False)
        d0, d1, kpts = cu(a["x"]), cu(a["source"]), cu(synth.grid_kpts(8, 8))
        with pytest.raises(NotImplementedError):
            gnn(d0, d1)                                                   # nothing installed yet: the double has no forward
        with torch.no_grad():
            touched = dropin.install()
            assert "models.modules.AttentionalGNN.forward" in touched and "models.modules.KeypointEncoder.forward" in touched
            check("eval", gnn.eval()(d0, d1))
            one = gnn.layers[0].eval()(d0, d1)
            idx = torch.from_numpy(g["idx"]).cuda()
            np.testing.assert_allclose(one.reshape(-1)[idx].cpu().numpy(), g["one"], atol=1e-4, rtol=2e-4)
            stats = [(l.mlp[1].running_mean.clone(), l.mlp[1].running_var.clone()) for l in gnn.layers]
            check("train", gnn.train()(d0, d1))
            for l, (mu, var) in zip(gnn.layers, stats):                   # the HIP path leaves the running statistics alone
                assert torch.equal(l.mlp[1].running_mean, mu) and torch.equal(l.mlp[1].running_var, var)
            np.testing.assert_allclose(kenc.eval()(kpts).cpu().numpy(), g["kenc_eval"], atol=5e-5, rtol=2e-4)
            np.testing.assert_allclose(kenc.train()(kpts).cpu().numpy(), g["kenc_train"], atol=5e-5, rtol=2e-4)
            # the parameter caches follow the weights (round-2 advice): a checkpoint loaded AFTER the first forward, then an
            # in-place optimizer-style update
            gnn.eval()
            load(gnn, other)
            check("load", gnn(d0, d1))
            gnn.layers[1].attn.merge.weight.mul_(1.5)
            check("step", gnn(d0, d1))
        # with autograd on and parameters that require grad the module's ORIGINAL forward runs (the HIP path is inference
        # only): for these doubles that is the NotImplementedError above
        gnn.train()
        with pytest.raises(NotImplementedError):
            gnn(d0.clone().requires_grad_(True), d1)
        dropin.uninstall()
        assert not hasattr(gnn.layers[0], "_pats_params")                # uninstall() dropped the caches
        with pytest.raises(NotImplementedError):
            gnn(d0, d1)
    finally:
        dropin.uninstall()
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def test_cost_fp16_split_path_against_float64_and_range_fallback(ops):
    """csrc/cost.hip: the default contraction (fp16 hi + lo, three MFMA passes) against float64 at the coarse / fine
    shapes, ragged channel counts and edge tiles; operands beyond the fp16 range (|x| > 1023) make the workgroup redo
    its tile on the fp32 path in-kernel; inf / NaN inputs come out as the fp32 chain has them."""
    rng = np.random.default_rng(21)

    def run(b, D, n, m, scale=1.0, spike=None):
        d0 = (rng.standard_normal((b, D, n)) * scale).astype(np.float32)
        d1 = (rng.standard_normal((b, D, m)) * scale).astype(np.float32)
        if spike is not None:
            d0[0, D // 2, n // 3] = spike
        with np.errstate(invalid="ignore", over="ignore"):
            truth = np.einsum("bdn,bdm->bnm", d0.astype(np.float64), d1.astype(np.float64)) / np.sqrt(float(D)) * 0.1
        return ops.cost(cu(d0), cu(d1)).cpu().numpy(), truth

    for shape in ((2, 448, 300, 300), (3, 264, 145, 145), (2, 128, 40, 77), (2, 100, 161, 33), (2, 7, 16, 500),
                  (1, 24, 321, 163), (2, 17, 3, 2)):
        S, truth = run(*shape)
        assert np.abs(S - truth).max() < 6e-7, shape                 # |S| <= 0.6: a few fp32 ulps
    S, truth = run(2, 264, 145, 145, scale=30.0)
    assert np.abs(S - truth).max() < 4e-4 and np.abs(truth).max() > 100
    S, truth = run(2, 264, 145, 145, scale=1e-3)                      # far below the range where `lo` stays normal
    assert np.abs(S - truth).max() < 1e-9
    S, truth = run(2, 264, 145, 145, spike=5000.0)                    # hi overflows -> fp32 redo of that workgroup
    assert np.isfinite(S).all() and np.abs(S - truth).max() < 1e-4
    S, truth = run(2, 264, 145, 145, spike=float("inf"))
    bad = ~np.isfinite(truth)
    assert bad.any() and (~np.isfinite(S) == bad).all() and np.abs(S[~bad] - truth[~bad]).max() < 6e-7


def test_cost_fp32_path_still_selectable():
    """PATS_COST_F32=1 routes pats_cost_f32 through the fp32-MFMA contraction (the in-kernel fallback of the split path)."""
    import subprocess
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from pats_amd import ops, synth; "
            "inp = synth.fine_inputs(seed=3, B=4); "
            "S = ops.cost(torch.from_numpy(inp['d0']).cuda(), torch.from_numpy(inp['d1']).cuda()).cpu().numpy(); "
            "np.save(sys.argv[1], S)") % REPO
    import tempfile
    outs = []
    base = {k: v for k, v in os.environ.items() if k != "PATS_COST_F32"}      # whatever the ambient setting is
    for env in ({"PATS_COST_F32": "1"}, {}):
        with tempfile.NamedTemporaryFile(suffix=".npy") as f:
            subprocess.run([sys.executable, "-c", code, f.name], env=dict(base, **env), check=True)
            outs.append(np.load(f.name))
    assert not np.array_equal(outs[0], outs[1])                       # two different contractions ...
    np.testing.assert_allclose(outs[0], outs[1], rtol=3e-6, atol=4e-6)   # ... a few ulps of the largest terms apart (|S| up to 16)


def test_third_level_fp16_split_cost_build_matches_the_fp32_build(ops, oracle):
    """The fused third level builds its scores from fp16-split operands (three exact-product MFMA passes, fp32
    accumulation); PATS_THIRD_VARIANT=300 is the same kernel with the fp32 MFMA.  Both must agree with the oracle to
    the same gates - and descriptors beyond the fp16 range (|x| > 1023 after the 2^6 pre-scale) must come out right
    through the guard (inf -> NaN scores -> re-solve by the fp32 log-domain kernel)."""
    import subprocess
    code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
import pats_oracle as oracle
from pats_amd import ops, synth
inp = synth.third_inputs(seed=synth.SEED + 64, P=512)
d0, d1 = inp["d0"].copy(), inp["d1"].copy()
d0[7] *= 100.0; d1[7] *= 100.0                      # elements up to ~1500: beyond the split's range
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ops.sinkhorn_fallbacks(reset=True)
m0, m1, label, ifm = ops.third_level(cu(d0), cu(d1), cu(inp["scale"]), cu(inp["p_s"]), cu(inp["p_t"]), outdoor=True)
trips = ops.sinkhorn_fallbacks(reset=True)
Zr = oracle.log_optimal_transport2(oracle.cost(d0, d1), 1.0, inp["scale"], 100)
sq = np.sqrt(inp["scale"] + np.float32(1e-8)).astype(np.float32)
r0, r1, rwl, rlabel, rifm = oracle.compute_result(np.exp(Zr), sq, sq, inp["p_s"], inp["p_t"], True)
tame = np.setdiff1d(np.arange(512), [7])
assert np.array_equal(label.cpu().numpy().reshape(512, 16, 2)[tame], rlabel.reshape(512, 16, 2)[tame])
assert np.array_equal(ifm.cpu().numpy().astype(bool)[tame], rifm.astype(bool)[tame])
assert np.array_equal(m0.cpu().numpy(), r0)
d = np.abs(m1.cpu().numpy()[tame] - r1[tame]).max()
assert d <= 3e-4 * 8, d
assert np.isfinite(m1.cpu().numpy()).all() and trips >= 1
print("OK", d, trips)
''' % (REPO, REPO)
    for variant in (None, "300"):        # the default = the one instantiation the production library holds (fp32-MFMA cost build)
        env = dict(os.environ) if variant is None else dict(os.environ, PATS_THIRD_VARIANT=variant)
        env.pop("PATS_THIRD_VARIANT", None) if variant is None else None
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "OK" in p.stdout, str(variant) + ": " + p.stdout[-500:] + p.stderr[-1500:]
    # every other variant (the fp16-split cost build - not bit-reproducible on its first launch -, sweep-loop experiments,
    # timing ablations that are wrong by design) exists in libpats_amd_diag.so only (csrc/third_fused3.hip, -DPATS_DIAG).  Since
    # round 5 the production library does not even READ the variant switch (diag_env() is a constant there): setting it changes
    # nothing, the one production instantiation runs and passes the same gates
    for variant in ("1350", "1308"):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PATS_THIRD_VARIANT=variant, PATS_THIRD_ABLATION="1"),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "OK" in p.stdout, variant + ": " + p.stdout[-500:] + p.stderr[-1500:]


FUSED_FINE_CHILD = r"""
import sys, os, numpy as np, torch
sys.path.insert(0, %(repo)r)
from pats_amd import ops
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
forced_log = os.environ.get("PATS_TEST_SINKHORN_MODE") == "log"
if forced_log:
    ops.set_sinkhorn_mode("log")
bitwise = os.environ.get("PATS_FINE_W2") == "0" or forced_log
rng = np.random.default_rng(31)
B, D = 300, 264
base = rng.standard_normal((B, D, 145)).astype(np.float32)
d0 = (3.0 * (base + 0.3 * rng.standard_normal((B, D, 145)))).astype(np.float32)
d1 = (3.0 * (base + 0.3 * rng.standard_normal((B, D, 145)))).astype(np.float32)
d0[:, :, -1] *= 0.5
d1[:, :, -1] *= 0.5
wild = [7, 150, 299]
d0[wild] *= 12.0                                          # scores x 144: far outside the linear-domain guard band
d1[wild] *= 12.0
ns = np.exp(0.3 * rng.standard_normal((B, 1, 144))).astype(np.float32)
ops.sinkhorn_fallbacks(reset=True)
prev = ops.set_fine_fused(True)

def same(a, b, what):
    if bitwise:
        assert torch.equal(a, b), what + ": differ in %%d problems" %% int((a != b).flatten(1).any(1).sum())
    else:       # the two-wave kernel sums in another order: the mass gate of the parity tests
        ea, eb = torch.exp(a.double()), torch.exp(b.double())
        assert torch.allclose(ea, eb, atol=1e-4, rtol=5e-6), what + ": %%g" %% float((ea - eb).abs().max())

for bias in (2.0, 0.0):
    Zf, flags = ops.cost_ot(cu(d0), cu(d1), 2, 1.0, cu(ns), 100, bias_k=bias, return_flags=True)
    trips = ops.sinkhorn_fallbacks(reset=True)
    S = ops.cost(cu(d0), cu(d1))
    Zu = ops.log_optimal_transport2(S, 1.0, cu(ns), 100, bias_k=bias)
    ops.sinkhorn_fallbacks(reset=True)
    if not forced_log:
        assert trips >= len(wild)                       # (the forced log domain has no guard to trip)
    same(Zf, Zu, "fused and two-kernel log-plans")
    want_flags = Zf[:, -1, :-1] > Zf[:, :-1, :-1].max(1).values          # second_layer.py:243,248
    assert torch.equal(flags, want_flags)
    assert torch.equal(ops.cost_ot(cu(d0), cu(d1), 2, 1.0, cu(ns), 100, bias_k=bias), Zf)      # without the flags
    ops.set_fine_fused(False)
    Z2, flags2 = ops.cost_ot(cu(d0), cu(d1), 2, 1.0, cu(ns), 100, bias_k=bias, return_flags=True)      # the default path
    ops.set_fine_fused(True)
    same(Z2, Zf, "default path and fused kernel")
    assert torch.equal(flags2, Z2[:, -1, :-1] > Z2[:, :-1, :-1].max(1).values)               # every plan's flags are its own
    if bitwise:
        assert torch.equal(flags2, flags)
    else:
        assert float((flags2 != flags).float().mean()) < 1e-4                                 # a flag may flip only on a near-tie
ops.set_fine_fused(prev)
sys.path.insert(0, os.path.join(%(repo)r, "oracle"))
import pats_oracle as oracle
n = 6
want = oracle.log_optimal_transport2(oracle.cost(d0[:n], d1[:n]), 1.0, ns[:n], 100)
np.testing.assert_allclose(np.exp(Zf[:n].cpu().numpy().astype(np.float64)), np.exp(want.astype(np.float64)), atol=1e-4, rtol=5e-6)
np.testing.assert_allclose(np.exp(Z2[:n].cpu().numpy().astype(np.float64)), np.exp(want.astype(np.float64)), atol=1e-4, rtol=5e-6)
print("OK")
"""


def test_fused_fine_level_cost_ot_is_the_two_kernel_path_bit_for_bit(ops, oracle, sinkhorn_mode):
    """ops.cost_ot at 145 x 145 (the fused kernel: MFMA cost tile -> register blocks through an LDS band buffer -> sweeps ->
    epilogue, scores never in HBM) against ops.cost + ops.log_optimal_transport2 on the same descriptors, with problems that leave
    the guard band (the fused kernel hands their raw scores to the log-domain kernel in place).  Every bit of the log-plan and the
    column flags against the FOUR-wave kernel it shares its sweep loop with (PATS_FINE_W2=0: a child process, the switch is read
    once), and under the mass gate against the two-wave kernel that is the default since round 4 (another summation order)."""
    import subprocess
    env = dict(os.environ)
    if sinkhorn_mode != "kernel":
        env["PATS_TEST_SINKHORN_MODE"] = "log"
    for w2 in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", FUSED_FINE_CHILD % {"repo": REPO}], env=dict(env, PATS_FINE_W2=w2), capture_output=True,
                           text=True, timeout=900)
        assert p.returncode == 0 and "OK" in p.stdout, "PATS_FINE_W2=" + w2 + ": " + p.stdout[-500:] + p.stderr[-2500:]


# ---- the fine level's one-kernel layer and stack (csrc/gnn_fine.hip, round 5) -----------------------------------------------------
def test_fine_level_gnn_stack_matches_the_layer_by_layer_path(ops, oracle):
    """pats_attentional_gnn_packed_f32 (descriptors kept as (blocked fp32, fragment image) between the layers) against the same
    layers run one by one through pats_attentional_propagation_packed_f32, and against the oracle: four layers, self / cross, more
    problems than workgroups (every scratch block and the LDS slot are reused), odd batch."""
    C, n, b = 264, 145, 301
    ps = [synth.gnn_params(seed=300 + i, C=C) for i in range(4)]
    names = ["self", "cross", "cross", "self"]
    P = [ops.PropagationParams(p) for p in ps]
    a = synth.gnn_inputs(seed=310, b=b, C=C, n=n)
    d0, d1 = cu(a["x"]), cu(a["source"])
    s0, s1 = ops.attentional_gnn(d0, d1, P, names)
    assert torch.equal(d0, cu(a["x"])) and torch.equal(d1, cu(a["source"]))                    # inputs are borrowed
    l0, l1 = d0, d1
    for p, name in zip(P, names):
        x0, x1 = (l1, l0) if name == "cross" else (l0, l1)
        l0, l1 = ops.attentional_propagation(l0, x0, p, residual=l0), ops.attentional_propagation(l1, x1, p, residual=l1)
    # the stack rounds nothing the single layer does not; the only difference is the order of the two halves of mlp[3]'s sum
    np.testing.assert_allclose(s0.cpu().numpy(), l0.cpu().numpy(), atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(s1.cpu().numpy(), l1.cpu().numpy(), atol=2e-5, rtol=2e-5)
    # the first 3 problems of each set against the oracle's four layers
    r0, r1 = a["x"][:3], a["source"][:3]
    full0, full1 = a["x"], a["source"]
    # (cross layers couple row i of one set with row i of the other only: three rows suffice)
    for p, name in zip(ps, names):
        y0, y1 = (r1, r0) if name == "cross" else (r0, r1)
        r0, r1 = (oracle.attentional_propagation(r0, y0, p, residual=r0), oracle.attentional_propagation(r1, y1, p, residual=r1))
    np.testing.assert_allclose(s0[:3].cpu().numpy(), r0, atol=1e-4, rtol=2e-4)
    np.testing.assert_allclose(s1[:3].cpu().numpy(), r1, atol=1e-4, rtol=2e-4)
    # launch-to-launch identical
    t0, t1 = ops.attentional_gnn(d0, d1, P, names)
    assert torch.equal(s0, t0) and torch.equal(s1, t1)


def test_fine_level_gnn_stack_beyond_the_fp16_range(ops, oracle):
    """An activation beyond +-1023 overflows the hi half of the split operands: the one-kernel layer raises its flag and the gated
    per-layer compositions queued behind the stack redo it from the inputs (fp32 MFMA inside conv1x1_kernel) - no host read."""
    C, n, b = 264, 145, 4
    ps = [synth.gnn_params(seed=320 + i, C=C) for i in range(2)]
    a = synth.gnn_inputs(seed=330, b=b, C=C, n=n)
    x, s = a["x"].copy(), a["source"].copy()
    x[1, 7, 33] = 3.0e4                                                                           # one wild descriptor entry
    d0, d1 = ops.attentional_gnn(cu(x), cu(s), [ops.PropagationParams(p) for p in ps], ["self", "cross"])
    r0, r1 = x, s
    for p, name in zip(ps, ["self", "cross"]):
        y0, y1 = (r1, r0) if name == "cross" else (r0, r1)
        r0, r1 = (oracle.attentional_propagation(r0, y0, p, residual=r0), oracle.attentional_propagation(r1, y1, p, residual=r1))
    assert np.isfinite(d0.cpu().numpy()).all()
    np.testing.assert_allclose(d0.cpu().numpy(), r0, atol=2e-2, rtol=2e-4)                        # (outputs of magnitude 3e4 in row 1)
    np.testing.assert_allclose(d1.cpu().numpy(), r1, atol=2e-2, rtol=2e-4)
    np.testing.assert_allclose(d0.cpu().numpy()[[0, 2, 3]], r0[[0, 2, 3]], atol=1e-4, rtol=2e-4)
