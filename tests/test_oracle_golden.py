"""Pins the CPU oracle (oracle/pats_oracle.c) against fixtures produced by the REFERENCE's own
code (tools/make_golden.py).  CPU only.  Gates: indices exact, transport
mass |d exp(Z)| <= 1e-4 element-wise and on marginals, crops <= 1e-4 abs on 0-255 data."""
import numpy as np
import pytest
import torch

from conftest import golden
from pats_amd import synth

MASS_TOL = 1e-4


def assert_mass(Zo, Zr):
    eo, er = np.exp(Zo.astype(np.float64)), np.exp(Zr.astype(np.float64))
    np.testing.assert_allclose(eo, er, atol=MASS_TOL, rtol=2e-6)
    # marginals: 1e-4 absolute, plus fp32 resolution on the dustbin marginals (mass ~ sum(ns) >> 1)
    np.testing.assert_allclose(eo.sum(-1), er.sum(-1), atol=MASS_TOL, rtol=3e-6)
    np.testing.assert_allclose(eo.sum(-2), er.sum(-2), atol=MASS_TOL, rtol=3e-6)
    # log-plan itself agrees tightly where mass is non-negligible
    big = er > 1e-6
    assert np.abs(Zo[big] - Zr[big]).max() <= 2e-4


def test_kat(oracle):
    g = golden("ot_kat.npz")
    z1 = oracle.log_optimal_transport(g["s1"], g["a1"], g["n1"], 100)
    assert_mass(z1, g["z1"])
    # the SURVEY 8c known-answer values
    np.testing.assert_allclose(np.exp(z1[0, 0]), [.3251170, .2107056, .0291975, .4349800], atol=2e-6)
    np.testing.assert_allclose(np.exp(z1).sum(2)[0], [1, 1, 3.5], atol=1e-5)
    np.testing.assert_allclose(np.exp(z1).sum(1)[0], [1, 2, .5, 2], atol=1e-5)
    z2 = oracle.log_optimal_transport2(g["s2"], 1.0, g["n2"], 100)
    assert_mass(z2, g["z2"])
    np.testing.assert_allclose(np.exp(z2[0, 2]), [1.1363239, .1686142, 1.1950618], atol=2e-6)


@pytest.mark.parametrize("iters", [1, 3, 100])
def test_sinkhorn_raw_ragged(oracle, iters):
    g = golden("sinkhorn_raw.npz")
    out = oracle.log_sinkhorn_iterations(g["Z"], g["log_mu"], g["log_nu"], iters)
    np.testing.assert_allclose(out, g["out%d" % iters], atol=2e-5, rtol=0)


def test_exact_ties_first_index(oracle):
    g = golden("ot_ties.npz")
    z = oracle.log_optimal_transport(g["s"], g["alpha"], g["ns"], 100)
    assert_mass(z, g["z"])
    r, c = oracle.argmax(z)
    assert np.array_equal(r, g["max0"]) and np.array_equal(c, g["max1"])
    # the duplicated columns/rows really are exact ties in the oracle too
    assert np.array_equal(z[0, :, 3], z[0, :, 7]) and np.array_equal(z[0, 2, :], z[0, 9, :])
    assert r[0, 5] == 3 and c[0, 4] == 2


def _coarse(oracle, name, seed, h, w):
    g = golden(name)
    inp = synth.coarse_inputs(seed=seed, h=h, w=w)
    assert synth.checksum(inp["d0"], inp["d1"], inp["ns"]) == pytest.approx(float(g["in_checksum"]), rel=1e-12)
    S = oracle.cost(inp["d0"], inp["d1"])
    np.testing.assert_allclose(S.reshape(-1)[g["S_idx"]], g["S_val"], atol=2e-5, rtol=1e-5)
    Z = oracle.log_optimal_transport(S, inp["alpha"], inp["ns"], 100)
    return g, inp, S, Z


@pytest.mark.parametrize("name,seed,h,w", [("coarse_301.npz", synth.SEED, 15, 20),
                                           ("coarse_portrait.npz", synth.SEED + 20, 20, 15)])
def test_coarse_ot_and_expand(oracle, name, seed, h, w):
    g, inp, S, Z = _coarse(oracle, name, seed, h, w)
    assert_mass(Z, g["Z"])
    np.testing.assert_allclose(np.exp(Z).sum(2), g["row_mass"], atol=MASS_TOL, rtol=1e-6)
    np.testing.assert_allclose(np.exp(Z).sum(1), g["col_mass"], atol=MASS_TOL, rtol=1e-6)
    r, c = oracle.argmax(Z)
    assert np.array_equal(r[:, :-1], g["max0"]) and np.array_equal(c[:, :-1], g["max1"])
    n = h * w
    assert np.array_equal(r[:, :-1] == n, g["ifn1"]) and np.array_equal(c[:, :-1] == n, g["ifn2"])
    scales = oracle.colmass_sqrt(Z)
    np.testing.assert_allclose(scales, g["scales"], atol=2e-5)
    # expansion on the reference's own Z (so decisions see identical inputs)
    P = np.exp(g["Z"])
    whole, core, avg, xs, ys, bound = oracle.iterative_expand(P, g["scales"], g["scales"], w, h, w, 1e-5, 15)
    assert np.array_equal(bound, g["bound"])
    np.testing.assert_allclose(whole, g["whole_cost"], atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(core, g["core_cost"], atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(avg, g["average_point"], atol=1e-4)
    np.testing.assert_allclose(xs, g["x_scale"], rtol=2e-5)
    np.testing.assert_allclose(ys, g["y_scale"], rtol=2e-5)
    # ... and end to end on the oracle's own Z: integer outputs still identical
    _, _, _, _, _, bound2 = oracle.iterative_expand(np.exp(Z), scales, scales, w, h, w, 1e-5, 15)
    assert np.array_equal(bound2, g["bound"])


@pytest.mark.parametrize("name,seed,h,w", [("coarse_769.npz", synth.SEED + 21, 24, 32),          # YFCC, BASELINE configs[3]
                                           ("coarse_1901.npz", synth.SEED + 22, 38, 50)])        # the demo's size (demo.py:36)
def test_coarse_769_sampled(oracle, name, seed, h, w):
    g, inp, S, Z = _coarse(oracle, name, seed, h, w)
    zs = Z.reshape(-1)[g["Z_idx"]]
    e0, e1 = np.exp(zs.astype(np.float64)), np.exp(g["Z_val"].astype(np.float64))
    assert np.abs(e0 - e1).max() <= MASS_TOL
    r, c = oracle.argmax(Z)
    assert np.array_equal(r[:, :-1], g["max0"]) and np.array_equal(c[:, :-1], g["max1"])
    np.testing.assert_allclose(np.exp(Z).sum(2), g["row_mass"], atol=MASS_TOL, rtol=1e-5)
    np.testing.assert_allclose(np.exp(Z).sum(1), g["col_mass"], atol=MASS_TOL, rtol=1e-5)
    whole, core, avg, xs, ys, bound = oracle.iterative_expand(np.exp(Z), oracle.colmass_sqrt(Z),
                                                              oracle.colmass_sqrt(Z), w, h, w, 1e-5, 15)
    assert np.array_equal(bound, g["bound"])
    np.testing.assert_allclose(avg, g["average_point"], atol=1e-4)


@pytest.mark.parametrize("cap", [40, 100, 512])
def test_split_patches(oracle, cap):
    g = golden("coarse_301.npz")
    sum_cycle = np.cumsum(~g["ifn1"][0]).astype(np.int32)
    n, second, third = oracle.split_patches(sum_cycle, 15, 20, cap)
    assert n == int(g["split%d_cycle" % cap])
    assert np.array_equal(second, g["split%d_second" % cap])
    assert np.array_equal(third, g["split%d_third" % cap])


def test_split_patches_first_row_overflow(oracle):
    # i == 0 branch: Python's sum_cycle[-1] wrap (utils.py:165) is reproduced
    sc = np.cumsum(np.ones(12, np.int32)).astype(np.int32)
    n, second, third = oracle.split_patches(sc, 3, 4, 3)
    assert n == 4
    assert second.tolist() == [[0, 4], [0, 8], [4, 12], [8, 12]]
    assert third.tolist() == [[1, 4 - 12], [5, 4], [5, 4], [0, 0]]


def test_compute_imgs_and_resize(oracle):
    g = golden("coarse_301.npz")
    left, right = synth.image_pair()
    assert synth.checksum(left, right) == pytest.approx(float(g["img_checksum"]), rel=1e-12)
    bound5, xsn, ysn, avn = oracle.compute_imgs_bounds(g["x_scale"], g["y_scale"], g["average_point"],
                                                       g["ifn1"], 15, 20)
    assert np.array_equal(bound5, g["resize_bound"])
    np.testing.assert_allclose(xsn[None], g["x_scale_new"], rtol=1e-6)
    np.testing.assert_allclose(ysn[None], g["y_scale_new"], rtol=1e-6)
    np.testing.assert_allclose(avn[None], g["average_new"], atol=1e-5)
    # right crops through the subdivision gather
    src = np.zeros((1, 3, 480 + 256, 640 + 256), np.float32)
    src[0, :, 128:-128, 128:-128] = right[0].transpose(2, 0, 1)
    assert list(src.shape) == g["resize_src_shape"].tolist()
    crops = oracle.tensor_resize(src, bound5).transpose(0, 2, 3, 1)
    assert crops.shape[0] == int(g["K"])
    np.testing.assert_allclose(crops[g["crop_pick"]], g["right_pick"], atol=1e-4)
    np.testing.assert_allclose(crops.astype(np.float64).sum((1, 2, 3)), g["right_sum"], rtol=1e-6)
    wts = np.arange(96 * 96 * 3, dtype=np.float64).reshape(96, 96, 3)
    np.testing.assert_allclose((crops.astype(np.float64) * wts).sum((1, 2, 3)), g["right_wsum"], rtol=1e-6)
    # left crops (origin_extract)
    lc = oracle.left_crops(left[0], g["ifn1"][0], 15, 20)
    np.testing.assert_array_equal(lc[g["crop_pick"]][:, ::4, ::4], g["left_pick"])
    np.testing.assert_allclose(lc.astype(np.float64).sum((1, 2, 3)), g["left_sum"], rtol=1e-9)


def test_resize_small_edges(oracle):
    g = golden("resize_small.npz")
    out = oracle.tensor_resize(g["src"], g["bound"])
    np.testing.assert_allclose(out, g["out"], atol=1e-4)
    with pytest.raises(RuntimeError):
        oracle.tensor_resize(g["src"], np.array([[5, 5, 0, 3, 0]]))   # empty crop: torch raises too
    assert oracle.tensor_resize(g["src"], np.zeros((0, 5), np.int64)).shape == (0, 3, 96, 96)


@pytest.mark.parametrize("name,k", [("fine_145.npz", 2.0), ("fine_145_indoor.npz", 3.0)])
def test_fine_layer(oracle, name, k):
    g = golden(name)
    B = int(g["B"])
    inp = synth.fine_inputs(seed=int(g["seed"]), B=B)
    assert synth.checksum(inp["d0"], inp["d1"], inp["scale_x"], inp["scale_y"]) == \
        pytest.approx(float(g["in_checksum"]), rel=1e-12)
    S = oracle.cost(inp["d0"], inp["d1"])
    np.testing.assert_allclose(S.reshape(-1)[g["S_idx"]], g["S_val"], atol=2e-5, rtol=1e-5)
    Z0 = oracle.log_optimal_transport2(S, 1.0, inp["scale_x"] * inp["scale_y"], 100)
    np.testing.assert_allclose(np.exp(Z0).sum(2), g["row_mass"], atol=MASS_TOL, rtol=1e-5)
    np.testing.assert_allclose(np.exp(Z0).sum(1), g["col_mass"], atol=MASS_TOL, rtol=1e-5)
    Z = oracle.dustbin_bias(Z0, k)
    assert_mass(Z, g["Z"])
    r, c = oracle.argmax(Z)
    assert np.array_equal(r[:, :-1], g["max0"]) and np.array_equal(c[:, :-1], g["max1"])
    assert np.array_equal(r[:, :-1] == 144, g["ifn1"])
    out = oracle.iterative_expand(np.exp(g["Z"]), inp["scale_x"], inp["scale_y"], 12, 12, 12, 1e-3, 8)
    whole, core, avg, xs, ys, bound = out
    assert np.array_equal(bound, g["bound"])
    np.testing.assert_allclose(whole, g["whole_cost"], atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(core, g["core_cost"], atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(avg, g["average_point"], atol=1e-4)
    np.testing.assert_allclose(xs, g["x_scale"], rtol=2e-5)
    np.testing.assert_allclose(ys, g["y_scale"], rtol=2e-5)
    bound2 = oracle.iterative_expand(np.exp(Z), inp["scale_x"], inp["scale_y"], 12, 12, 12, 1e-3, 8)[5]
    assert np.array_equal(bound2, g["bound"])


@pytest.mark.parametrize("name", ["third_65.npz", "third_65_indoor.npz"])
def test_third_layer(oracle, name):
    g = golden(name)
    P, outdoor = int(g["P"]), bool(g["outdoor"])
    inp = synth.third_inputs(seed=int(g["seed"]), P=P)
    assert synth.checksum(inp["d0"], inp["d1"], inp["scale"]) == pytest.approx(float(g["in_checksum"]), rel=1e-12)
    S = oracle.cost(inp["d0"], inp["d1"])
    np.testing.assert_allclose(S.reshape(-1)[g["S_idx"]], g["S_val"], atol=2e-5, rtol=1e-5)
    Z = oracle.log_optimal_transport2(S, 1.0, inp["scale"], 100)
    assert_mass(Z, g["Z"])
    sxy = np.sqrt(inp["scale"] + np.float32(1e-8)).astype(np.float32)
    for Zin in (g["Z"], Z):
        m0, m1, wl, label, ifm = oracle.compute_result(np.exp(Zin), sxy, sxy, inp["p_s"], inp["p_t"], outdoor)
        np.testing.assert_array_equal(m0, g["mkpts0_f"])
        np.testing.assert_allclose(m1, g["mkpts1_f"], atol=2e-4)
        np.testing.assert_allclose(wl, g["whole_loss"], atol=1e-6)
        assert np.array_equal(ifm, g["if_matching1"])
        np.testing.assert_array_equal(label, g["label"])


def test_fine_descriptors(oracle):
    g = golden("fine_desc.npz")
    inp = synth.fine_maps()
    assert synth.checksum(inp["f0"], inp["f1"], inp["f2"], inp["title"], inp["rubbish"]) == \
        pytest.approx(float(g["in_checksum"]), rel=1e-12)
    desc = oracle.fine_descriptors(inp["f0"], inp["f1"], inp["f2"], inp["title"], inp["rubbish"])
    assert desc.shape == (2, 3, 264, 145)
    np.testing.assert_array_equal(desc.reshape(-1)[g["idx"]], g["val"])          # pure gathers + exact /4
    np.testing.assert_allclose(desc.astype(np.float64).sum((2, 3)), g["sum_per_block"], rtol=1e-12)
    np.testing.assert_array_equal(desc[:, 0][:, ::7, ::5], g["first"])


def test_third_descriptors(oracle):
    g = golden("third_desc.npz")
    inp = synth.third_maps()
    o0, o1, ps, pt = oracle.third_descriptors(inp["ff0"], inp["ff1"], inp["mk0"], inp["mk1"], inp["b_ids"],
                                              inp["kenc"], inp["rubbish"])
    assert np.array_equal(ps, g["p_s"]) and np.array_equal(pt, g["p_t"])
    np.testing.assert_array_equal(o0[:, ::4, :], g["out0"])
    np.testing.assert_array_equal(o1[:, ::4, :], g["out1"])
    np.testing.assert_allclose(o0.astype(np.float64).sum((1, 2)), g["sum0"], rtol=1e-12)
    np.testing.assert_allclose(o1.astype(np.float64).sum((1, 2)), g["sum1"], rtol=1e-12)
    bad, bid = inp["mk1"].copy(), inp["b_ids"].copy()
    bad[0], bid[0] = [0.0, 0.0], 0                # flattened index goes negative: torch.gather raises
    with pytest.raises(IndexError):
        oracle.third_descriptors(inp["ff0"], inp["ff1"], inp["mk0"], bad, bid, inp["kenc"], inp["rubbish"])


def test_third_descriptors_on_the_border_ring(oracle):
    """Source points on cells 0 / 11 of the 12x12 grid (third_desc_ring.npz, made from the reference's own lines): windows
    that wrap in the flattened NHWC view, and the dustbin index round(92 / 8) = 12 that reads the NEXT patch's feature
    (third_layer.py:127,141-144) - rows of the flattened views, not clamped per map."""
    g = golden("third_desc_ring.npz")
    inp = synth.third_maps_ring()
    o0, o1, ps, pt = oracle.third_descriptors(inp["ff0"], inp["ff1"], inp["mk0"], inp["mk1"], inp["b_ids"],
                                              inp["kenc"], inp["rubbish"])
    assert np.array_equal(ps, g["p_s"]) and np.array_equal(pt, g["p_t"])
    np.testing.assert_array_equal(o0[:, ::4, :], g["out0"])
    np.testing.assert_array_equal(o1[:, ::4, :], g["out1"])
    np.testing.assert_allclose(o0.astype(np.float64).sum((1, 2)), g["sum0"], rtol=1e-12)
    np.testing.assert_allclose(o1.astype(np.float64).sum((1, 2)), g["sum1"], rtol=1e-12)
    assert (inp["mk0"] == 92).any(1).sum() > 8 and (ps == 92).all(1).any()
    last = inp["b_ids"].copy()
    last[0] = inp["ff0"].shape[0] - 1               # cell (11, 11) of the LAST patch: the dustbin row leaves the tensor
    with pytest.raises(IndexError):
        oracle.third_descriptors(inp["ff0"], inp["ff1"], inp["mk0"], inp["mk1"], last, inp["kenc"], inp["rubbish"])


# ---- SURVEY.md section 8(f) rows -------------------------------------------------------------------
def _merge_case(name):
    g = golden(name)
    inp = synth.merge_inputs(seed=int(g["seed"]), h=int(g["h"]), w=int(g["w"]))
    assert synth.checksum(*[ch["trust"] for ch in inp["chunks"]]) == float(g["in_checksum"])
    return g, inp


@pytest.mark.parametrize("name", ["merge_new.npz", "merge_old.npz", "merge_new_portrait.npz"])
def test_merge_patches(oracle, name):
    """second_layer.py:137-238 over three successive chunks; bit-exact flags, trust and scores_back."""
    g, inp = _merge_case(name)
    new, h, w = bool(g["merge_new"]), inp["h"], inp["w"]
    sb = np.zeros((1, h * w, 16, 9))
    for c, ch in enumerate(inp["chunks"]):
        out, t, f2, sb_w = oracle.merge_patches(new, ch["trust"], (h * 32, w * 32), ch["ifn_L1"], ch["ifn2"], sb)
        assert np.array_equal(out, g["out%d" % c])
        assert np.array_equal(t, g["trust%d" % c])
        assert np.array_equal(f2, g["ifn2_%d" % c])
        assert np.array_equal(sb_w.astype(np.float32), g["sb_written%d" % c])
        assert int(g["sb_returned_zero%d" % c]) == (0 if new else 1)
        sb = sb_w if new else np.zeros_like(sb_w)          # what the reference hands back (:191 / :240)
    assert 0 < (~out).sum() < out.size


def test_merge_patches_wrong_row_count_raises(oracle):
    _, inp = _merge_case("merge_new.npz")
    ch = inp["chunks"][0]
    with pytest.raises(IndexError):
        oracle.merge_patches(True, ch["trust"][:-1], (480, 640), ch["ifn_L1"], ch["ifn2"][:-1], np.zeros((1, 300, 16, 9)))


def _result_case(name):
    g = golden(name)
    inp = synth.result_inputs(seed=int(g["seed"]), h=5, w=6, mixed_choice=bool(g["mixed"]))
    assert synth.checksum(inp["ap0"], inp["sc0"], inp["pts"], inp["mkpts1"], inp["label0"]) == float(g["in_checksum"])
    return g, inp


@pytest.mark.parametrize("name", ["result.npz", "result_mixed.npz"])
def test_third_inputs_scatter_and_get_result(oracle, name):
    """pats.py:53-78 + utils.py:189-213: bit-exact (index work and a fixed fp32 operation order)."""
    g, inp = _result_case(name)
    mk0, mk1, b_ids = oracle.third_inputs(inp["ifn2"], inp["pts"])
    assert np.array_equal(mk0, g["mk0"]) and np.array_equal(mk1, g["mk1"]) and np.array_equal(b_ids, g["b_ids"])
    f16, p16 = oracle.refine_scatter(inp["ifn2"], inp["pts"], inp["mkpts1"], inp["label0"])
    assert np.array_equal(f16, g["ifn16"]) and np.array_equal(p16, g["pts16"])
    sc1 = np.repeat(inp["sc0"][~inp["ifn0"]].reshape(-1, 1, 2), 2304, 1)
    ml, mr = oracle.get_result(1, [inp["ifn0"], f16], [inp["ap0"], p16[:, :, ::-1] / np.float32(2.0)],
                               [inp["sc0"], sc1], [[32, 5, 6], [2, 48, 48]], [inp["choice0"], inp["choice1"]])
    assert np.array_equal(ml, g["matches_l"]) and np.array_equal(mr, g["matches_r"])
    assert ml.shape[0] > 1000


def test_attention(oracle):
    """modules.py:84-88 at the path's three token counts and a ragged case (fp32: 1e-5 on the output,
    2e-6 on probabilities, rows of prob sum to 1)."""
    import sys as _s, os as _o
    _s.path.insert(0, _o.path.join(_o.path.dirname(_o.path.dirname(_o.path.abspath(__file__))), "tools"))
    g = golden("attention.npz")
    cases = [dict(b=3, dim=32, heads=4, n=65), dict(b=2, dim=66, heads=4, n=145, amp=1.5),
             dict(b=1, dim=112, heads=4, n=300), dict(b=2, dim=6, heads=2, n=37, m=53, amp=2.0)]
    for c, kw in enumerate(cases):
        inp = synth.attention_inputs(seed=synth.SEED + 12 + c, **kw)
        assert synth.checksum(inp["q"], inp["k"], inp["v"]) == float(g["in_checksum%d" % c])
        x, prob = oracle.attention(inp["q"], inp["k"], inp["v"])
        np.testing.assert_allclose(x.reshape(-1)[g["x_idx%d" % c]], g["x_val%d" % c], atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(prob.reshape(-1)[g["p_idx%d" % c]], g["p_val%d" % c], atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(prob.sum(-1), g["p_rowsum%d" % c], atol=2e-6)


# ---- AttentionalPropagation / AttentionalGNN (modules.py:91-134), fixtures from the reference's own classes ----
GNN_CASES = [dict(C=128, b=3, n=65, m=65), dict(C=64, b=2, n=145, m=145), dict(C=32, b=2, n=37, m=53),
             dict(C=264, b=5, n=145, m=145), dict(C=448, b=2, n=300, m=300)]      # 3, 4: the fine and the coarse level's production shapes


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_attentional_propagation_against_the_reference_class(oracle, case):
    g = golden("gnn_layer.npz")
    kw = GNN_CASES[case]
    params = synth.gnn_params(seed=synth.SEED + 70 + case, C=kw["C"])
    inp = synth.gnn_inputs(seed=synth.SEED + 80 + case, b=kw["b"], C=kw["C"], n=kw["n"], m=kw["m"])
    assert abs(synth.checksum(inp["x"], inp["source"], params["mlp.0.weight"]) - float(g["in_checksum%d" % case])) < 1e-6
    for mode in ("eval", "train"):
        y = oracle.attentional_propagation(inp["x"], inp["source"], params, bn_train=(mode == "train"))
        np.testing.assert_allclose(y.reshape(-1)[g["%s_idx%d" % (mode, case)]], g["%s_val%d" % (mode, case)], atol=3e-5, rtol=1e-4)
        np.testing.assert_allclose(y.astype(np.float64).sum((1, 2)), g["%s_sum%d" % (mode, case)], atol=2e-2, rtol=1e-4)


def test_merge_folded_into_mlp0_is_the_same_layer(oracle):
    """The identity behind pats_propagation_pack_f32's fold (csrc/gnn_fused.hip gnn_fold_kernel): modules.py:104,116
    mlp[0](cat([x, merge(att)])) = W1x x + (W1m Wm) att + (W1m bm + b1).  A layer whose merge is the identity and whose mlp[0]
    carries the folded matrix and bias must reproduce the oracle's layer to double-rounding level - on the CPU, no kernel involved."""
    C = 64
    p = synth.gnn_params(seed=synth.SEED + 71, C=C)
    inp = synth.gnn_inputs(seed=synth.SEED + 81, b=2, C=C, n=37, m=53)
    W1 = p["mlp.0.weight"][:, :, 0].astype(np.float64)
    Wm, bm = p["attn.merge.weight"][:, :, 0].astype(np.float64), p["attn.merge.bias"].astype(np.float64)
    q = dict(p)
    q["attn.merge.weight"] = np.eye(C, dtype=np.float32)[:, :, None]
    q["attn.merge.bias"] = np.zeros(C, np.float32)
    q["mlp.0.weight"] = np.concatenate([W1[:, :C], W1[:, C:] @ Wm], 1).astype(np.float32)[:, :, None]
    q["mlp.0.bias"] = (p["mlp.0.bias"].astype(np.float64) + W1[:, C:] @ bm).astype(np.float32)
    for train in (False, True):
        want = oracle.attentional_propagation(inp["x"], inp["source"], p, bn_train=train)
        got = oracle.attentional_propagation(inp["x"], inp["source"], q, bn_train=train)
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=2e-5)


def test_attentional_gnn_two_layers(oracle):
    g = golden("gnn_layer.npz")
    ps = [synth.gnn_params(seed=synth.SEED + 90 + i, C=128) for i in range(2)]
    a = synth.gnn_inputs(seed=synth.SEED + 95, b=4, C=128, n=65)
    d0, d1 = a["x"], a["source"]
    for p, name in zip(ps, ["self", "cross"]):
        s0, s1 = (d1, d0) if name == "cross" else (d0, d1)
        d0, d1 = (oracle.attentional_propagation(d0, s0, p, residual=d0), oracle.attentional_propagation(d1, s1, p, residual=d1))
    np.testing.assert_allclose(d0.reshape(-1)[g["gnn_idx"]], g["gnn_d0"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(d1.reshape(-1)[g["gnn_idx"]], g["gnn_d1"], atol=5e-5, rtol=1e-4)


def test_attentional_gnn_three_layers_at_the_fine_level_shape(oracle):
    """AttentionalGNN(264, [self, cross, self]) from the reference's class (second_layer.py:44,89 runs 18 such layers)."""
    g = golden("gnn_layer.npz")
    ps = [synth.gnn_params(seed=synth.SEED + 96 + i, C=264) for i in range(3)]
    a = synth.gnn_inputs(seed=synth.SEED + 99, b=3, C=264, n=145)
    d0, d1 = a["x"], a["source"]
    for p, name in zip(ps, ["self", "cross", "self"]):
        s0, s1 = (d1, d0) if name == "cross" else (d0, d1)
        d0, d1 = (oracle.attentional_propagation(d0, s0, p, residual=d0), oracle.attentional_propagation(d1, s1, p, residual=d1))
    np.testing.assert_allclose(d0.reshape(-1)[g["gnn264_idx"]], g["gnn264_d0"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(d1.reshape(-1)[g["gnn264_idx"]], g["gnn264_d1"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(d0.astype(np.float64).sum((1, 2)), g["gnn264_sum0"], atol=3e-2, rtol=1e-4)
    np.testing.assert_allclose(d1.astype(np.float64).sum((1, 2)), g["gnn264_sum1"], atol=3e-2, rtol=1e-4)


# ---- the descriptor heads: KeypointEncoder (modules.py:70-82) and final_proj (Conv1d, first_layer.py:34-36,105) ----
@pytest.mark.parametrize("tag,dim,h,w,seed", [("third", 128, 8, 8, synth.SEED + 100), ("first", 448, 15, 20, synth.SEED + 101)])
def test_keypoint_encoder_against_the_reference_class(oracle, tag, dim, h, w, seed):
    g = golden("heads.npz")
    params = synth.kenc_params(seed=seed, feature_dim=dim)
    assert abs(synth.checksum(params["encoder.0.weight"], params["encoder.15.weight"]) - float(g["kenc_%s_checksum" % tag])) < 1e-9
    kpts = synth.grid_kpts(h, w)
    for mode in ("eval", "train"):
        y = oracle.keypoint_encoder(kpts, params, bn_train=(mode == "train"))
        assert y.shape == (1, dim, h * w)
        if tag == "third":
            np.testing.assert_allclose(y, g["kenc_third_%s" % mode], atol=2e-5, rtol=1e-4)
        else:
            np.testing.assert_allclose(y.reshape(-1)[g["kenc_first_idx"]], g["kenc_first_%s" % mode], atol=2e-5, rtol=1e-4)
            assert abs(float(y.astype(np.float64).sum()) - float(g["kenc_first_%s_sum" % mode])) < 2e-2


@pytest.mark.parametrize("tag,C,b,n,seed", [("first", 448, 1, 300, synth.SEED + 110), ("second", 264, 6, 145, synth.SEED + 111)])
def test_final_proj_against_torch_conv1d(oracle, tag, C, b, n, seed):
    g = golden("heads.npz")
    p = synth.final_proj_params(seed=seed, C=C)
    x = synth.gnn_inputs(seed=seed + 5, b=b, C=C, n=n)["x"]
    y = oracle.conv1d(x, p["weight"], p["bias"])
    np.testing.assert_allclose(y.reshape(-1)[g["proj_%s_idx" % tag]], g["proj_%s_val" % tag], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(y.astype(np.float64).sum((1, 2)), g["proj_%s_sum" % tag], atol=2e-2, rtol=1e-4)


SCALE_CASES = [("first", 448, 2, 15, 20, 1, False, synth.SEED + 120), ("second", 264, 6, 12, 12, 2, True, synth.SEED + 121),
               ("third", 128, 40, 8, 8, 1, True, synth.SEED + 122)]


@pytest.mark.parametrize("tag,C,b,h,w,heads,dust,seed", SCALE_CASES)
def test_scale_head_against_torch_conv2d(oracle, tag, C, b, h, w, heads, dust, seed):
    """first_layer.py:106-107 / second_layer.py:92-98 / third_layer.py:151-152 re-executed on nn.Conv2d (tools/make_golden.py)."""
    g = golden("heads.npz")
    ws, bs = synth.scale_head_params(seed=seed, C=C, heads=heads)
    x = (4.0 * synth.gnn_inputs(seed=seed + 5, b=b, C=C, n=h * w + int(dust))["x"]).astype(np.float32)
    assert abs(synth.checksum(x, ws[0]) - float(g["scale_%s_checksum" % tag])) < 1e-6 * max(1.0, abs(float(g["scale_%s_checksum" % tag])))
    y = oracle.scale_head(x, h, w, ws, bs)
    want = g["scale_%s" % tag]
    assert y.shape == want.shape == (b, 1, h * w)
    assert want.min() >= (1 / 16) ** heads * (1 - 1e-5) and want.max() <= 16.0 ** heads * (1 + 1e-5) and want.std() > 0.05    # a non-trivial fixture
    np.testing.assert_allclose(y, want, rtol=2e-5)


@pytest.mark.parametrize("h,w", [(15, 20), (20, 15), (12, 12), (24, 32)])
def test_positions_and_ranges_tables(oracle, h, w):
    """a9 pinned directly (utils/utils.py:1527-1537): the oracle's tables AND the product's host function against the values the
    reference's Compute_positions_and_ranges returned (tools/make_golden.py::gen_positions_ranges)."""
    g = golden("positions_ranges.npz")
    pos, rng = oracle.compute_positions_and_ranges(h, w)
    np.testing.assert_array_equal(pos, g["positions_%dx%d" % (h, w)])
    np.testing.assert_array_equal(rng, g["ranges_%dx%d" % (h, w)])
    from pats_amd import ops
    p2, r2 = ops.Compute_positions_and_ranges(h, w, "cpu")
    assert p2.dtype == torch.float32 and r2.dtype == torch.float32
    np.testing.assert_array_equal(p2.numpy(), g["positions_%dx%d" % (h, w)])
    np.testing.assert_array_equal(r2.numpy(), g["ranges_%dx%d" % (h, w)])


def test_reference_honours_a_different_ranges_table():
    """The fixture's control: handed a shifted `ranges`, the reference's own expansion returns different rectangles - so a
    replacement must either read the tensor or refuse anything but the canonical table (ops._grid_of refuses; GPU test)."""
    g = golden("positions_ranges.npz")
    assert bool(g["wrong_changes_bound"]) and (g["bound_canonical"] != g["bound_wrong_ranges"]).sum() > 100
