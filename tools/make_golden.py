#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code (this container only).

TEST INFRASTRUCTURE.  Imports /root/reference through tools/ref_import.py (torch CPU), feeds it
the deterministic inputs of pats_amd/synth.py and stores inputs' checksums + the reference's
outputs as small fixtures.  The fixtures are data; no reference source is copied.
Re-run:  python tools/make_golden.py   (needs /root/reference and oracle/_ref built).

What each fixture pins (reference file:line):
  ot_kat.npz       log_optimal_transport / log_optimal_transport2 known answers
                   (models/modules.py:145-182; values quoted in SURVEY.md section 8c)
  sinkhorn_raw.npz log_sinkhorn_iterations on a ragged 2x21x23 problem (models/modules.py:137-143)
  ot_ties.npz      exact column duplicates -> argmax first-index tie-break (first_layer.py:162)
  coarse_301.npz   L1: cost einsum, OT, column mass, est_position, Iterative_expand_matrix,
                   split_patches, Compute_imgs/tensor_resize (first_layer.py:110-146,159-178;
                   utils/utils.py:152-181,1179-1393; setup/library.cpp:47-66)
  coarse_portrait.npz  same expansion on a 20x15 grid (the height/width swap quirk, utils.py:1181)
  coarse_769.npz   YFCC-sized L1 (24x32 grid): argmax, marginals, sampled Z
  coarse_1901.npz  the demo's L1 (demo.py:36: long side 1 600 -> 38x50 grid, 1901x1901 OT): the same, between 769^2 and 4097^2
  fine_145.npz     L2: cost, log_optimal_transport2, dustbin bias ln2, est_position(8 iters)
                   (second_layer.py:100-116,240-259)
  fine_145_indoor.npz  L2 with ln3 bias
  third_65.npz     L3: cost, OT2, Compute_result, outdoor label (third_layer.py:156-170,184-217)
  third_65_indoor.npz  indoor label rule
  resize_small.npz tensor_resize edge cases (1-pixel crops, borders) (setup/library.cpp:47-66)
  merge_new.npz / merge_old.npz / merge_new_portrait.npz   merge_patches_new / _old over three
                   successive chunks incl. the scores_back hand-over (second_layer.py:137-238)
  pipeline_outdoor.npz / pipeline_indoor.npz   the whole path chained in the reference's order on synthetic
                   network outputs (MegaDepth-style: local chunks, outdoor, merge_new; ScanNet-style: one
                   chunk, indoor, merge_old) -> final matches_l / matches_r
  attention.npz    attention(query, key, value) of the GNN layers (models/modules.py:84-88)
  third_desc_ring.npz  the a16 gather for source points on the border ring of the cell grid: windows that wrap
                   in the flattened NHWC view, the dustbin index that reads the next patch (third_layer.py:127,141-144)
  dropin_gnn.npz   AttentionalGNN (3 layers) / AttentionalPropagation / KeypointEncoder instances of the reference in eval and
                   train mode and after two parameter changes (models/modules.py:70-134): the drop-in test's expected values
  positions_ranges.npz  Compute_positions_and_ranges' two tables for 15x20, 20x15, 12x12, 24x32 (utils/utils.py:1527-1537) and the
                   reference's expansion on a shifted `ranges` (the tensor is an input it honours)
  result.npz / result_mixed.npz   third-level inputs, result scatter and get_result
                   (pats.py:53-78, utils/utils.py:189-213); _mixed flips left_choice per row
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from pats_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
T = torch.from_numpy


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print("%-24s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


def sample_idx(rng, shape, k=1024):
    flat = rng.choice(int(np.prod(shape)), size=min(k, int(np.prod(shape))), replace=False)
    return np.sort(flat).astype(np.int64)


def cost(d0, d1, D):
    """The forward-body expression of first_layer.py:110-111,114 / second_layer.py:100-101,104 /
    third_layer.py:156-158, re-executed verbatim on synthetic tensors."""
    scores = torch.einsum('bdn,bdm->bnm', d0, d1)
    scores = scores / D ** .5
    return 0.1 * scores


def gen_kat(R):
    s1 = torch.tensor([[[1., 0., -1.], [0., 2., .5]]])
    a1 = torch.tensor(.5)
    n1 = torch.tensor([[[1., 2., .5]]])
    z1 = R.M.log_optimal_transport(s1, a1, n1, 100)
    s2 = torch.tensor([[[1., 0., .2], [0., 2., .1], [.3, .3, 0.]]])
    n2 = torch.tensor([[[2., .5]]])
    z2 = R.M.log_optimal_transport2(s2, torch.tensor(1.0), n2, 100)
    save("ot_kat.npz", s1=s1, a1=a1, n1=n1, z1=z1, s2=s2, n2=n2, z2=z2)


def gen_sinkhorn_raw(R):
    rng = np.random.default_rng(synth.SEED + 10)
    Z = (2.0 * rng.standard_normal((2, 21, 23), dtype=np.float32))
    mu = rng.uniform(0.5, 2.0, (2, 21)).astype(np.float32)
    nu = rng.uniform(0.5, 2.0, (2, 23)).astype(np.float32)
    nu *= (mu.sum(1, keepdims=True) / nu.sum(1, keepdims=True))
    log_mu, log_nu = np.log(mu), np.log(nu)
    outs = {}
    for it in (1, 3, 100):
        outs["out%d" % it] = R.M.log_sinkhorn_iterations(T(Z), T(log_mu), T(log_nu), it)
    save("sinkhorn_raw.npz", Z=Z, log_mu=log_mu, log_nu=log_nu, **outs)


def gen_ties(R):
    rng = np.random.default_rng(synth.SEED + 11)
    s = rng.standard_normal((1, 12, 12), dtype=np.float32)
    s[0, :, 7] = s[0, :, 3]          # duplicate columns -> exactly tied Z columns
    s[0, 9, :] = s[0, 2, :]          # duplicate rows    -> exactly tied Z rows
    s[0, 5, 3] = s[0, 5, 7] = 6.0    # make the tie the row maximum
    s[0, 2, 4] = s[0, 9, 4] = 6.0    # make the tie the column maximum
    ns = np.ones((1, 1, 12), np.float32)
    ns[0, 0, 7] = ns[0, 0, 3] = 1.5
    z = R.M.log_optimal_transport(T(s), torch.tensor(0.25), T(ns), 100)
    save("ot_ties.npz", s=s, ns=ns, alpha=np.float32(0.25), z=z,
         max0=z.max(2).indices, max1=z.max(1).indices)


def run_coarse(R, inp, H, W, name, full=True, with_imgs=True):
    h, w = inp["h"], inp["w"]
    d0, d1, ns = T(inp["d0"]), T(inp["d1"]), T(inp["ns"])
    D = d0.shape[1]
    S = cost(d0, d1, D)
    Z = R.M.log_optimal_transport(S, torch.tensor(float(inp["alpha"])), ns, 100)
    scales = torch.sqrt(Z[:, :-1, :-1].exp().sum(1) + 1e-8)          # first_layer.py:117-118
    trust, pts, xs, ys, ifn1, ifn2 = R.L1.FirstLayer.est_position(None, Z, scales, (H, W), 32)
    # the same call est_position makes (first_layer.py:173-175), to also pin core_cost and bound
    positions, ranges = R.U.Compute_positions_and_ranges(H // 32, W // 32, 'cpu')
    lim = torch.tensor([0, H // 32, 0, W // 32])
    sc = scales.reshape(scales.shape[0], -1, 1)
    whole, core, avg, xs2, ys2, bound = R.U.Iterative_expand_matrix(
        Z.exp(), sc, sc, lim, ranges, positions, height=H // 32, width=W // 32,
        iter_num=15, lower_bound=1e-5)
    assert torch.equal(whole, trust) and torch.equal(avg, pts)
    rng = np.random.default_rng(1)
    sidx = sample_idx(rng, S.shape)
    out = dict(in_checksum=synth.checksum(inp["d0"], inp["d1"], inp["ns"]),
               H=H, W=W, S_idx=sidx, S_val=S.reshape(-1)[T(sidx)],
               scales=scales, max0=Z.max(2).indices[:, :-1], max1=Z.max(1).indices[:, :-1],
               ifn1=ifn1, ifn2=ifn2, whole_cost=whole, core_cost=core, average_point=avg,
               x_scale=xs, y_scale=ys, bound=bound,
               row_mass=Z.exp().sum(2), col_mass=Z.exp().sum(1))
    if full:
        out["Z"] = Z
    else:
        zi = sample_idx(rng, Z.shape, 4096)
        out["Z_idx"], out["Z_val"] = zi, Z.reshape(-1)[T(zi)]
    # split_patches (utils.py:152-181) on the cumsum the caller builds (first_layer.py:130)
    sum_cycle = torch.cumsum(torch.logical_not(ifn1).int(), dim=1)
    for cap in (w * 2, 512, 100):
        cyc, second, third = R.U.split_patches(sum_cycle[0], h, w, cap)
        out["split%d_cycle" % cap] = np.int64(cyc)
        out["split%d_second" % cap] = np.array([[int(a), int(b)] for a, b in second], np.int64)
        out["split%d_third" % cap] = np.array([[int(a), int(b)] for a, b in third], np.int64)
    if with_imgs:
        left, right = synth.image_pair(H=H, W=W)
        out["img_checksum"] = synth.checksum(left, right)
        nl, nr, xsn, ysn, avgn = R.U.Compute_imgs(xs, ys, pts, ifn1, T(left), T(right),
                                                  width=w, height=h)
        K = nr.shape[0]
        pick = np.unique(np.linspace(0, K - 1, 8).astype(np.int64))
        out.update(K=np.int64(K), x_scale_new=xsn, y_scale_new=ysn, average_new=avgn,
                   crop_pick=pick, right_pick=nr[T(pick)], left_pick=nl[T(pick)][:, ::4, ::4],
                   right_sum=nr.double().sum((1, 2, 3)), left_sum=nl.double().sum((1, 2, 3)),
                   right_wsum=(nr.double() * torch.arange(96 * 96 * 3).reshape(96, 96, 3)).sum((1, 2, 3)))
        # the exact bound tensor handed to the native module (utils.py:1380-1381), recomputed by
        # the same expressions so the gather kernel can be pinned in isolation
        captured = {}
        orig = R.U.tensor_resize.tensor_resize

        def spy(src, bnd):
            captured["bound"] = bnd.clone()
            captured["src_shape"] = np.array(src.shape, np.int64)
            return orig(src, bnd)
        R.U.tensor_resize = types.SimpleNamespace(tensor_resize=spy)
        try:
            R.U.Compute_imgs(xs, ys, pts, ifn1, T(left), T(right), width=w, height=h)
        finally:
            R.U.tensor_resize = types.SimpleNamespace(tensor_resize=orig)
        out["resize_bound"] = captured["bound"]
        out["resize_src_shape"] = captured["src_shape"]
    save(name, **out)
    return Z


def gen_fine(R, name, B, outdoor, seed):
    inp = synth.fine_inputs(seed=seed, B=B)
    d0, d1 = T(inp["d0"]), T(inp["d1"])
    sx, sy = T(inp["scale_x"]), T(inp["scale_y"])
    S = cost(d0, d1, 264)
    one = torch.tensor(1.0)
    Z0 = R.M.log_optimal_transport2(S, one, sx * sy, 100)
    Z = Z0.clone()
    bias = torch.log(one * 2) if outdoor else torch.log(one * 3)   # second_layer.py:107-112
    Z[:, :, -1] += bias
    Z[:, -1, :] += bias
    trust, pts, xs, ys, ifn1, ifn2 = R.L2.SecondLayer.est_position(None, Z, sx, sy, [96, 96], 8)
    positions, ranges = R.U.Compute_positions_and_ranges(12, 12, 'cpu')
    whole, core, avg, xs2, ys2, bound = R.U.Iterative_expand_matrix(
        Z.exp(), sx.reshape(B, -1, 1), sy.reshape(B, -1, 1), torch.tensor([0, 12, 0, 12]),
        ranges, positions, height=12, width=12, iter_num=8, lower_bound=1e-3)
    assert torch.equal(whole, trust)
    rng = np.random.default_rng(2)
    sidx = sample_idx(rng, S.shape)
    save(name, in_checksum=synth.checksum(inp["d0"], inp["d1"], inp["scale_x"], inp["scale_y"]),
         B=np.int64(B), outdoor=np.int64(outdoor), seed=np.int64(seed),
         S_idx=sidx, S_val=S.reshape(-1)[T(sidx)], Z=Z,
         max0=Z.max(2).indices[:, :-1], max1=Z.max(1).indices[:, :-1], ifn1=ifn1, ifn2=ifn2,
         whole_cost=whole, core_cost=core, average_point=avg, x_scale=xs, y_scale=ys, bound=bound,
         row_mass=Z0.exp().sum(2), col_mass=Z0.exp().sum(1))


def gen_third(R, name, P, outdoor, seed):
    inp = synth.third_inputs(seed=seed, P=P)
    d0, d1, scale = T(inp["d0"]), T(inp["d1"]), T(inp["scale"])
    p_s, p_t = T(inp["p_s"]), T(inp["p_t"])
    S = cost(d0, d1, 128)
    one = torch.tensor(1.0)
    Zo = R.M.log_optimal_transport2(S, one, scale, 100)           # third_layer.py:158
    scores = torch.exp(Zo)
    scale_x = (scale + 1e-8).sqrt()                               # third_layer.py:153-154
    scale_y = (scale + 1e-8).sqrt()
    ns = types.SimpleNamespace(pad=torch.nn.ZeroPad2d(2), pad_1=torch.nn.ConstantPad2d(2, 1e-2))
    m0, m1, wl = R.L3.ThirdLayer.Compute_result(ns, scores, 8, 5, scale_x, scale_y, p_s, p_t, 'cpu')
    # label rule, third_layer.py:161-170, re-executed verbatim
    Wd = 8
    label = ((torch.zeros_like(p_t[:, None, :].expand(-1, 16, -1).float())) + 1e8).reshape(-1, 2)
    if not outdoor:
        ar = torch.arange(label.shape[0])
        select1 = torch.logical_or(ar % 16 == 5, ar % 16 == 15)
        select2 = torch.logical_or(ar % 16 == 7, ar % 16 == 13)
        select = torch.logical_or(select1, select2)
        label[:, 0] = torch.where(select, label[:, 0], torch.tensor(-10.0))
    scores_used = scores[:, :-1, :].reshape(scores.shape[0], Wd, Wd, -1)[:, 2:6, 2:6, :] \
        .reshape(scores.shape[0], 16, -1) + 1e-8
    if_matching1 = (scores_used.max(2)[1] != Wd ** 2)
    if outdoor:
        label[:, 0] = torch.where(if_matching1.reshape(-1), label[:, 0], torch.tensor(-10.0))
    rng = np.random.default_rng(3)
    sidx = sample_idx(rng, S.shape)
    save(name, in_checksum=synth.checksum(inp["d0"], inp["d1"], inp["scale"]),
         P=np.int64(P), outdoor=np.int64(outdoor), seed=np.int64(seed),
         S_idx=sidx, S_val=S.reshape(-1)[T(sidx)], Z=Zo,
         mkpts0_f=m0, mkpts1_f=m1, whole_loss=wl, label=label, if_matching1=if_matching1,
         max0=scores[:, :-1, :-1].max(2)[1])


def gen_fine_desc(R):
    """second_layer.py:71-86 re-executed verbatim (self.* replaced by the constructor's values, :23,45-53)."""
    inp = synth.fine_maps()
    B = inp["f0"].shape[0] // 2
    desc0_ = [T(inp["f0"]), T(inp["f1"]), T(inp["f2"])]
    left = torch.zeros(B, 3, 96, 96)                       # only left.shape[0] / .device are used
    row_num, point_num, descriptor_dim = 12, 144, 264
    cols = torch.arange(0, row_num).reshape(row_num, 1).repeat(1, row_num).reshape(point_num)
    rows = torch.arange(0, row_num).reshape(1, row_num).repeat(row_num, 1).reshape(point_num)
    positions = torch.zeros((point_num, 2))
    positions[:, 0] = cols
    positions[:, 1] = rows
    avgpool = torch.nn.AvgPool2d(2, stride=1, padding=1)
    compress_1_out = T(inp["title"])                       # stands for self.compress_1(desc_l.unsqueeze(2))
    compress_2_out = T(inp["rubbish"])                     # stands for self.compress_2(desc_l.unsqueeze(2))
    desc = []
    for i, feat in enumerate(desc0_):
        stride = int(8.0 / torch.pow(torch.tensor(2.0, device=left.device), i + 1))
        if i <= 1:
            feat = avgpool(feat)
        index = ((positions.reshape(row_num, row_num, 2) + 0.5) * stride).long()
        index = (index[:, :, 0] * feat.shape[3] + index[:, :, 1]).reshape(1, 1, -1).\
            repeat(feat.shape[0], feat.shape[1], 1)
        desc.append(torch.gather(feat.reshape(feat.shape[0], feat.shape[1], -1), 2, index))
    desc = torch.cat(desc, dim=1).reshape(2, left.shape[0], 256, -1)
    title = compress_1_out.repeat(2, 1, point_num).reshape(2, left.shape[0], 8, -1)
    rubbish = compress_2_out.repeat(2, 1, 1).reshape(2, left.shape[0], descriptor_dim, 1)
    desc = torch.cat([title, desc], dim=2)
    desc = torch.cat([desc, rubbish], dim=3)
    rng = np.random.default_rng(4)
    idx = sample_idx(rng, desc.shape, 8192)
    save("fine_desc.npz", in_checksum=synth.checksum(inp["f0"], inp["f1"], inp["f2"], inp["title"], inp["rubbish"]),
         idx=idx, val=desc.reshape(-1)[T(idx)], sum_per_block=desc.double().sum((2, 3)),
         first=desc[:, 0, :, :].clone()[:, ::7, ::5])


def gen_third_desc(R, name="third_desc.npz", inp=None):
    """third_layer.py:121-146 re-executed verbatim (self.W = 8, self.M = 52, :108-110)."""
    inp = synth.third_maps() if inp is None else inp
    feat_f0, feat_f1 = T(inp["ff0"]), T(inp["ff1"])
    mkpts0_c, mkpts1_c, b_ids = T(inp["mk0"]), T(inp["mk1"]), T(inp["b_ids"])
    kenc_out = T(inp["kenc"])                               # stands for self.kenc(kpts)
    rubbish = T(inp["rubbish"])
    W_, M_ = 8, 52
    b = b_ids.reshape(-1, 1).repeat(1, W_ * W_)
    mkpts0_c = torch.round(mkpts0_c / 4.0).long() * 4
    x0 = (mkpts0_c[:, 0] // 2).reshape(-1, 1).expand(-1, W_ * W_) + torch.arange(W_).reshape(1, 1, W_).repeat(b_ids.shape[0], W_, 1).reshape(-1, W_ * W_) - W_ / 2 + 2
    y0 = (mkpts0_c[:, 1] // 2).reshape(-1, 1).expand(-1, W_ * W_) + torch.arange(W_).reshape(1, W_, 1).repeat(b_ids.shape[0], 1, W_).reshape(-1, W_ * W_) - W_ / 2 + 2
    index0 = (b * M_ * M_ + y0 * M_ + x0).long().reshape(-1, 1).expand(-1, 128)
    mkpts1_c = torch.where(mkpts1_c >= 96, torch.tensor(96).float(), mkpts1_c)
    mkpts1_c = torch.where(mkpts1_c <= 0, torch.tensor(0).float(), mkpts1_c)
    mkpts1_c = torch.round(mkpts1_c / 4.0).long() * 4
    x1 = (mkpts1_c[:, 0] // 2).reshape(-1, 1).expand(-1, W_ * W_) + torch.arange(W_).reshape(1, 1, W_).repeat(b_ids.shape[0], W_, 1).reshape(-1, W_ * W_) - W_ / 2 + 2
    y1 = (mkpts1_c[:, 1] // 2).reshape(-1, 1).expand(-1, W_ * W_) + torch.arange(W_).reshape(1, W_, 1).repeat(b_ids.shape[0], 1, W_).reshape(-1, W_ * W_) - W_ / 2 + 2
    index1 = (b * M_ * M_ + y1 * M_ + x1).long().reshape(-1, 1).expand(-1, 128)
    feat_f0_unfold = torch.gather(feat_f0.permute(0, 2, 3, 1).reshape(-1, feat_f0.shape[1]), 0, index0).reshape(-1, W_ * W_, 128).permute(0, 2, 1) + kenc_out
    feat_f1_unfold = torch.gather(feat_f1.permute(0, 2, 3, 1).reshape(-1, feat_f1.shape[1]), 0, index1).reshape(-1, W_ * W_, 128).permute(0, 2, 1) + kenc_out
    x2 = torch.round(mkpts0_c[:, 0] / 8.0).long()
    y2 = torch.round(mkpts0_c[:, 1] / 8.0).long()
    index2 = (b_ids * 12 * 12 + y2 * 12 + x2).long().reshape(-1, 1).expand(-1, 128)
    rubbish_unfold = torch.gather(rubbish.permute(0, 2, 1).reshape(-1, 128), 0, index2).reshape(-1, 128, 1)
    feat_f0_unfold = torch.cat([feat_f0_unfold, rubbish_unfold], dim=2)
    feat_f1_unfold = torch.cat([feat_f1_unfold, rubbish_unfold], dim=2)
    save(name, in_checksum=synth.checksum(inp["ff0"], inp["ff1"], inp["mk0"], inp["mk1"], inp["kenc"], inp["rubbish"]),
         p_s=mkpts0_c, p_t=mkpts1_c, out0=feat_f0_unfold[:, ::4, :], out1=feat_f1_unfold[:, ::4, :],
         sum0=feat_f0_unfold.double().sum((1, 2)), sum1=feat_f1_unfold.double().sum((1, 2)))


def gen_resize_small(R):
    rng = np.random.default_rng(synth.SEED + 12)
    src = rng.uniform(0, 255, (2, 3, 40, 50)).astype(np.float32)
    bound = np.array([
        [0, 40, 0, 49, 0],          # whole image 0   (rows [0,40), cols [0,49])
        [5, 6, 7, 7, 3],            # 1x1 crop
        [10, 11, 0, 49, 10007],     # 1 pixel high, image 1
        [0, 40, 20, 20, 10001],     # 1 pixel wide
        [3, 30, 4, 44, 20],         # generic downscale... 27 rows x 41 cols -> upsample
        [38, 40, 47, 49, 19999],    # bottom-right corner, image 1
        [12, 19, 13, 21, 5],        # small crop (7x9), strong upsample
    ], np.int64)
    out = R.tensor_resize.tensor_resize(T(src), T(bound))
    save("resize_small.npz", src=src, bound=bound, out=out)


def gen_merge(R, name, merge_new, seed, h=15, w=20):
    """second_layer.py:137-238 run chunk after chunk with the caller's scores_back hand-over (pats.py:32,37)."""
    inp = synth.merge_inputs(seed=seed, h=h, w=w)
    fn = R.L2.SecondLayer.merge_patches_new if merge_new else R.L2.SecondLayer.merge_patches_old
    sb = torch.zeros([1, h * w, 16, 9]).double()
    arrs = {}
    for c, ch in enumerate(inp["chunks"]):
        tr, f2, l1 = T(ch["trust"].copy()), T(ch["ifn2"].copy()), T(ch["ifn_L1"].copy())
        sb_arg = sb
        out, sb = fn(None, tr.shape[0], tr, (h * 32, w * 32), l1, f2, sb_arg)
        arrs["out%d" % c], arrs["trust%d" % c], arrs["ifn2_%d" % c] = out, tr, f2
        arrs["sb_written%d" % c] = sb_arg.to(torch.float32)        # the argument after the in-place write (fp32 values)
        arrs["sb_returned_zero%d" % c] = np.int64(int((sb == 0).all()))
    save(name, merge_new=np.int64(merge_new), seed=np.int64(seed), h=np.int64(h), w=np.int64(w),
         in_checksum=synth.checksum(*[ch["trust"] for ch in inp["chunks"]]), **arrs)


def gen_result(R, name, seed, mixed):
    """pats.py:53-78 re-executed verbatim on synthetic L1/L2/L3 outputs, then utils.get_result (utils.py:189-213)."""
    inp = synth.result_inputs(seed=seed, h=5, w=6, mixed_choice=mixed)     # small grid: the fixture holds pts16 in full
    h, w = inp["h"], inp["w"]
    ifn2, pts = T(inp["ifn2"]), T(inp["pts"])
    K = ifn2.shape[0]
    # pats.py:53-58
    sequence = torch.arange(0, 144).reshape(-1, 144, 1).repeat(K, 1, 1)
    third_input = torch.cat([sequence % 12 * 4 + 2, sequence // 12 * 4 + 2, torch.round(pts * 4)[:, :, [1, 0]],
                             torch.arange(K).reshape(-1, 1, 1).repeat(1, 144, 1)], dim=2)
    third_input = third_input[torch.logical_not(ifn2)]
    mk0, mk1, b_ids = third_input[:, :2] * 2, third_input[:, 2:4] * 2, third_input[:, -1]
    # pats.py:59-67
    mkpts1 = T(inp["mkpts1"]).reshape(-1, 2)
    pts16 = pts.reshape(-1, 144, 1, 2).repeat(1, 1, 16, 1)
    f16 = ifn2.reshape(-1, 144, 1).repeat(1, 1, 16)
    pts16[torch.logical_not(f16)] = mkpts1.float()
    label = torch.zeros_like(f16).float()
    label[torch.logical_not(f16)] = T(inp["label0"])
    f16 = torch.logical_or(f16, label < -9.9)
    f16 = f16.reshape(-1, 12, 12, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 144 * 16)
    pts16 = pts16.reshape(-1, 12, 12, 4, 4, 2).permute(0, 1, 3, 2, 4, 5).reshape(-1, 144 * 16, 2)
    # pats.py:68-76
    ifn0 = T(inp["ifn0"])
    if_nomatching = [ifn0, f16]
    patch_size = [[32, h, w], [2, 48, 48]]
    sc0 = T(inp["sc0"])
    scale = [sc0, sc0.reshape(-1, w * h, 2)[torch.logical_not(ifn0)].reshape(-1, 1, 2).repeat(1, 144 * 16, 1)]
    average_point = [T(inp["ap0"]), pts16.flip(dims=[2]) / 2.0]
    left_choice = [T(inp["choice0"]), T(inp["choice1"])]
    ml, mr = R.U.get_result(ifn0.shape[0], if_nomatching, average_point, scale, patch_size, left_choice)
    save(name, seed=np.int64(seed), mixed=np.int64(mixed), mk0=mk0, mk1=mk1, b_ids=b_ids, ifn16=f16, pts16=pts16,
         matches_l=ml, matches_r=mr,
         in_checksum=synth.checksum(inp["ap0"], inp["sc0"], inp["pts"], inp["mkpts1"], inp["label0"]))


ATTENTION_CASES = [dict(b=3, dim=32, heads=4, n=65), dict(b=2, dim=66, heads=4, n=145, amp=1.5),
                   dict(b=1, dim=112, heads=4, n=300), dict(b=2, dim=6, heads=2, n=37, m=53, amp=2.0)]


def gen_attention(R):
    """modules.py:84-88 at the three token counts of the path (65 / 145 / 300) and a ragged n != m case;
    outputs sampled (4096 entries each) plus the softmax row sums."""
    arrs = {}
    for c, kw in enumerate(ATTENTION_CASES):
        inp = synth.attention_inputs(seed=synth.SEED + 12 + c, **kw)
        x, prob = R.M.attention(T(inp["q"]), T(inp["k"]), T(inp["v"]))
        rng = np.random.default_rng(40 + c)
        xi, pi = sample_idx(rng, x.shape, 4096), sample_idx(rng, prob.shape, 4096)
        arrs.update({"x_idx%d" % c: xi, "x_val%d" % c: x.reshape(-1)[T(xi)], "p_idx%d" % c: pi,
                     "p_val%d" % c: prob.reshape(-1)[T(pi)], "p_rowsum%d" % c: prob.sum(-1),
                     "in_checksum%d" % c: synth.checksum(inp["q"], inp["k"], inp["v"])})
    save("attention.npz", **arrs)


def gen_pipeline(R, name, seed, h, w, if_local, if_outdoor, merge_new):
    """The whole path in the reference's own order: first_layer.py:110-157, second_layer.py:100-124,
    pats.py:32-78, third_layer.py:122,126-128,153-170 re-executed verbatim on the SynthNets stand-ins,
    every function call going to the reference's code.  Pins pats_amd.pipeline.forward_path."""
    nets = synth.SynthNets(seed=seed, h=h, w=w)
    left, right = [T(x) for x in nets.images()]
    H, W = h * 32, w * 32
    c = nets.coarse()
    one = torch.tensor(1.0)
    # first_layer.py:110-146
    scores = R.M.log_optimal_transport(cost(T(c["d0"]), T(c["d1"]), 448), torch.tensor(float(c["alpha"])), T(c["ns"]), iters=100)
    scales = torch.sqrt(scores[:, :-1, :-1].exp().sum(1) + 1e-8)
    trust, pts, xr, yr, if_nomatching1, if_nomatching2 = R.L1.FirstLayer.est_position(None, scores, scales, (H, W), 32)
    sum_cycle = torch.cumsum(torch.logical_not(if_nomatching1).int(), dim=1)
    max_cycle = w * 2 if if_local else 512
    cycle_num, second_layer_set, third_layer_set = R.U.split_patches(sum_cycle[0], h, w, max_cycle)
    matches_l, matches_r = torch.zeros([0, 2]), torch.zeros([0, 2])
    scores_refine_iter = torch.zeros([1, h * w, 16, 9]).double()
    info = []
    for num in range(cycle_num):
        mask = torch.where(torch.logical_and(if_nomatching1 == False,  # noqa: E712
                                             torch.logical_and(sum_cycle > second_layer_set[num][0],
                                                               sum_cycle <= second_layer_set[num][1])), False, True)
        new_left, new_right, x_scale_new, y_scale_new, average_new = R.U.Compute_imgs(xr, yr, pts, mask, left, right,
                                                                                        width=w, height=h)
        B = new_left.shape[0]
        # second_layer.py:100-124
        f = nets.fine(num, B)
        sx, sy = T(f["scale_x"]), T(f["scale_y"])
        Z2 = R.M.log_optimal_transport2(cost(T(f["d0"]), T(f["d1"]), 264), one, sx * sy, iters=100)
        bias = torch.log(one * 2) if if_outdoor else torch.log(one * 3)
        Z2[:, :, -1] += bias
        Z2[:, -1, :] += bias
        trust2, pts2, _, _, ifn_L2, _ = R.L2.SecondLayer.est_position(None, Z2, sx, sy, [96, 96], 8)
        merge = R.L2.SecondLayer.merge_patches_new if merge_new else R.L2.SecondLayer.merge_patches_old
        ifn_L2, scores_refine_iter = merge(None, B, trust2, (H, W), mask, ifn_L2, scores_refine_iter)
        # pats.py:38-78
        if third_layer_set[num][1] != 0:
            ifn_L2[-third_layer_set[num][1]:, :] = True
        if_ndelete = torch.logical_not(ifn_L2).int().sum(1).bool()
        ifn_L2 = ifn_L2[if_ndelete]
        if torch.logical_not(ifn_L2).float().sum() < 0.5:
            info.append((B, 0, 0))
            continue
        pts2 = pts2[if_ndelete]
        mask[torch.logical_not(mask)] = torch.logical_not(if_ndelete)
        nb = int(if_ndelete.sum())
        sequence = torch.arange(0, 144).reshape(-1, 144, 1).repeat(nb, 1, 1)
        third_input = torch.cat([sequence % 12 * 4 + 2, sequence // 12 * 4 + 2, torch.round(pts2 * 4)[:, :, [1, 0]],
                                 torch.arange(nb).reshape(-1, 1, 1).repeat(1, 144, 1)], dim=2)
        third_input = third_input[torch.logical_not(ifn_L2)]
        mkpts0_c, mkpts1_c = third_input[:, :2] * 2, third_input[:, 2:4] * 2
        P = mkpts0_c.shape[0]
        # third_layer.py:122,126-128,153-170
        t = nets.third(num, P)
        mkpts0_c = torch.round(mkpts0_c / 4.0).long() * 4
        mkpts1_c = torch.where(mkpts1_c >= 96, torch.tensor(96).float(), mkpts1_c)
        mkpts1_c = torch.where(mkpts1_c <= 0, torch.tensor(0).float(), mkpts1_c)
        mkpts1_c = torch.round(mkpts1_c / 4.0).long() * 4
        scale = T(t["scale"])
        scale_x, scale_y = (scale + 1e-8).sqrt(), (scale + 1e-8).sqrt()
        scores_origin = R.M.log_optimal_transport2(cost(T(t["d0"]), T(t["d1"]), 128), one, scale, iters=100)
        sc3 = torch.exp(scores_origin)
        ns_ = types.SimpleNamespace(pad=torch.nn.ZeroPad2d(2), pad_1=torch.nn.ConstantPad2d(2, 1e-2))
        mkpts0_f, mkpts1_f, _ = R.L3.ThirdLayer.Compute_result(ns_, sc3, 8, 5, scale_x, scale_y, mkpts0_c, mkpts1_c, 'cpu')
        label = ((torch.zeros_like(mkpts1_c[:, None, :].expand(-1, 16, -1).float())) + 1e8).reshape(-1, 2)
        if not if_outdoor:
            ar = torch.arange(label.shape[0])
            select = torch.logical_or(torch.logical_or(ar % 16 == 5, ar % 16 == 15), torch.logical_or(ar % 16 == 7, ar % 16 == 13))
            label[:, 0] = torch.where(select, label[:, 0], torch.tensor(-10.0))
        scores_used = sc3[:, :-1, :].reshape(P, 8, 8, -1)[:, 2:6, 2:6, :].reshape(P, 16, -1) + 1e-8
        if_matching1 = (scores_used.max(2)[1] != 64)
        if if_outdoor:
            label[:, 0] = torch.where(if_matching1.reshape(-1), label[:, 0], torch.tensor(-10.0))
        # pats.py:59-78
        mkpts1 = mkpts1_f.reshape(-1, 2)
        pts16 = pts2.reshape(-1, 144, 1, 2).repeat(1, 1, 16, 1)
        f16 = ifn_L2.reshape(-1, 144, 1).repeat(1, 1, 16)
        pts16[torch.logical_not(f16)] = mkpts1.float()
        lab = torch.zeros_like(f16).float()
        lab[torch.logical_not(f16)] = label[:, 0]
        f16 = torch.logical_or(f16, lab < -9.9)
        f16 = f16.reshape(-1, 12, 12, 4, 4).permute(0, 1, 3, 2, 4).reshape(-1, 144 * 16)
        pts16 = pts16.reshape(-1, 12, 12, 4, 4, 2).permute(0, 1, 3, 2, 4, 5).reshape(-1, 144 * 16, 2)
        if_nomatching = [mask, f16]
        patch_size = [[32, h, w], [2, 48, 48]]
        scale_l = [x_scale_new, x_scale_new.reshape(-1, w * h, 2)[torch.logical_not(mask)].reshape(-1, 1, 2).repeat(1, 144 * 16, 1)]
        average_point = [average_new.flip(dims=[2]) / 32.0, pts16.flip(dims=[2]) / 2.0]
        left_choice = [torch.ones([1]).bool(), torch.ones([f16.shape[0]]).bool()]
        ml, mr = R.U.get_result(1, if_nomatching, average_point, scale_l, patch_size, left_choice)
        matches_l, matches_r = torch.cat([matches_l, ml], dim=0), torch.cat([matches_r, mr], dim=0)
        info.append((B, P, ml.shape[0]))
    print("   %s: chunks (B, P, M) = %s" % (name, info))
    save(name, seed=np.int64(seed), h=np.int64(h), w=np.int64(w), if_local=np.int64(if_local), if_outdoor=np.int64(if_outdoor),
         merge_new=np.int64(merge_new), chunks=np.asarray(info, dtype=np.int64).reshape(-1, 3), matches_l=matches_l,
         matches_r=matches_r)


# cases 3, 4 (round 5): the two PRODUCTION shapes besides the third level's - the fine level (second_layer.py:44,89: 264 channels,
# 144 cells + dustbin) and the coarse level (first_layer.py:32,101: 448 channels, 15 x 20 cells)
GNN_CASES = [dict(C=128, b=3, n=65, m=65), dict(C=64, b=2, n=145, m=145), dict(C=32, b=2, n=37, m=53),
             dict(C=264, b=5, n=145, m=145), dict(C=448, b=2, n=300, m=300)]


def gen_gnn(R):
    """AttentionalPropagation (modules.py:107-117) instantiated from the reference itself, its parameters loaded
    from pats_amd.synth.gnn_params, in eval mode and in train mode (BatchNorm on batch statistics), plus two layers
    of AttentionalGNN.forward (self, cross; :127-134).  Outputs sampled (8192 entries) with their sums."""
    arrs = {}
    for c, kw in enumerate(GNN_CASES):
        C = kw["C"]
        params = synth.gnn_params(seed=synth.SEED + 70 + c, C=C)
        inp = synth.gnn_inputs(seed=synth.SEED + 80 + c, b=kw["b"], C=C, n=kw["n"], m=kw["m"])
        layer = R.M.AttentionalPropagation(C, 4)
        layer.load_state_dict({k: T(v) for k, v in params.items()}, strict=False)
        rng = np.random.default_rng(60 + c)
        for mode in ("eval", "train"):
            layer.train(mode == "train")
            with torch.no_grad():
                y = layer(T(inp["x"]), T(inp["source"]))
            idx = sample_idx(rng, y.shape, 8192)
            arrs.update({"%s_idx%d" % (mode, c): idx, "%s_val%d" % (mode, c): y.reshape(-1)[T(idx)],
                         "%s_sum%d" % (mode, c): y.double().sum((1, 2))})
            # train-mode forward updates the running statistics: restore them for the next use
            layer.load_state_dict({k: T(v) for k, v in params.items()}, strict=False)
        arrs["in_checksum%d" % c] = synth.checksum(inp["x"], inp["source"], params["mlp.0.weight"])
    # two GNN layers sharing the shape of the third level: desc <- desc + delta, self then cross
    C = 128
    gnn = R.M.AttentionalGNN(C, ["self", "cross"])
    ps = [synth.gnn_params(seed=synth.SEED + 90 + i, C=C) for i in range(2)]
    for lyr, p in zip(gnn.layers, ps):
        lyr.load_state_dict({k: T(v) for k, v in p.items()}, strict=False)
    gnn.eval()
    a = synth.gnn_inputs(seed=synth.SEED + 95, b=4, C=C, n=65)
    with torch.no_grad():
        d0, d1 = gnn(T(a["x"]), T(a["source"]))
    rng = np.random.default_rng(77)
    i0 = sample_idx(rng, d0.shape, 8192)
    arrs.update(gnn_idx=i0, gnn_d0=d0.reshape(-1)[T(i0)], gnn_d1=d1.reshape(-1)[T(i0)])
    # three layers at the fine level's shape (self, cross, self), eval mode: the stack second_layer.py:89 runs 18 layers of
    C = 264
    gnn = R.M.AttentionalGNN(C, ["self", "cross", "self"])
    ps = [synth.gnn_params(seed=synth.SEED + 96 + i, C=C) for i in range(3)]
    for lyr, p in zip(gnn.layers, ps):
        lyr.load_state_dict({k: T(v) for k, v in p.items()}, strict=False)
    gnn.eval()
    a = synth.gnn_inputs(seed=synth.SEED + 99, b=3, C=C, n=145)
    with torch.no_grad():
        d0, d1 = gnn(T(a["x"]), T(a["source"]))
    rng = np.random.default_rng(78)
    i0 = sample_idx(rng, d0.shape, 8192)
    arrs.update(gnn264_idx=i0, gnn264_d0=d0.reshape(-1)[T(i0)], gnn264_d1=d1.reshape(-1)[T(i0)],
                gnn264_sum0=d0.double().sum((1, 2)), gnn264_sum1=d1.double().sum((1, 2)))
    save("gnn_layer.npz", **arrs)


def gen_heads(R):
    """The descriptor heads either side of the GNN, from the reference's own modules: KeypointEncoder (modules.py:70-82) as
    the third layer builds it (third_layer.py:96-97; 8 x 8 grid :132-136) in eval and in train mode, as the first layer
    builds it (first_layer.py:30-31; 15 x 20 grid :74-79), and final_proj = nn.Conv1d(k=1) (first_layer.py:34-36,105;
    second_layer.py:40-42,91) at both levels' shapes."""
    arrs = {}
    for tag, dim, (h, w), seed in (("third", 128, (8, 8), synth.SEED + 100), ("first", 448, (15, 20), synth.SEED + 101)):
        params = synth.kenc_params(seed=seed, feature_dim=dim)
        kenc = R.M.KeypointEncoder(dim, [32, 64, 128, 256, 512])
        kenc.load_state_dict({k: T(v) for k, v in params.items()}, strict=False)
        # the grid exactly as the layer writes it (float division of an integer arange)
        cols = torch.arange(0, h).reshape(h, 1).repeat(1, w).reshape(-1) / float(h)
        rows = torch.arange(0, w).reshape(1, w).repeat(h, 1).reshape(-1) / float(w)
        kpts = torch.zeros((h * w), 2)
        kpts[:, 0] = cols
        kpts[:, 1] = rows
        assert np.array_equal(npy(kpts), synth.grid_kpts(h, w))
        for mode in ("eval", "train"):
            kenc.train(mode == "train")
            with torch.no_grad():
                y = kenc(kpts)
            if y.numel() > 16384:                       # the coarse level's [1,448,300]: 8192 sampled entries + the sum
                idx = sample_idx(np.random.default_rng(92), y.shape, 8192)
                arrs.update({"kenc_%s_idx" % tag: idx, "kenc_%s_%s" % (tag, mode): y.reshape(-1)[T(idx)],
                             "kenc_%s_%s_sum" % (tag, mode): y.double().sum()})
            else:
                arrs["kenc_%s_%s" % (tag, mode)] = y
            kenc.load_state_dict({k: T(v) for k, v in params.items()}, strict=False)     # train mode moved the running statistics
        arrs["kenc_%s_checksum" % tag] = synth.checksum(params["encoder.0.weight"], params["encoder.15.weight"])
    rng = np.random.default_rng(91)
    for tag, C, b, n, seed in (("first", 448, 1, 300, synth.SEED + 110), ("second", 264, 6, 145, synth.SEED + 111)):
        p = synth.final_proj_params(seed=seed, C=C)
        conv = torch.nn.Conv1d(C, C, kernel_size=1, bias=True)
        conv.load_state_dict({k: T(v) for k, v in p.items()})
        x = synth.gnn_inputs(seed=seed + 5, b=b, C=C, n=n)["x"]
        with torch.no_grad():
            y = conv(T(x))
        idx = sample_idx(rng, y.shape, 8192)
        arrs.update({"proj_%s_idx" % tag: idx, "proj_%s_val" % tag: y.reshape(-1)[T(idx)], "proj_%s_sum" % tag: y.double().sum((1, 2))})
    # the scale heads: the expressions of first_layer.py:106-107, second_layer.py:92-98 and third_layer.py:151-152, re-executed
    # on nn.Conv2d(C, 1, kernel_size=3, padding=1) modules as the layers build them
    sigmoid = torch.nn.Sigmoid()
    for tag, C, b, (h, w), heads, dust, seed in (("first", 448, 2, (15, 20), 1, False, synth.SEED + 120),
                                                 ("second", 264, 6, (12, 12), 2, True, synth.SEED + 121),
                                                 ("third", 128, 40, (8, 8), 1, True, synth.SEED + 122)):
        ws, bs = synth.scale_head_params(seed=seed, C=C, heads=heads)
        mdesc1 = T((4.0 * synth.gnn_inputs(seed=seed + 5, b=b, C=C, n=h * w + int(dust))["x"]).astype(np.float32))
        grid = (mdesc1[:, :, :-1] if dust else mdesc1[:, :, :]).reshape(mdesc1.shape[0], -1, h, w)
        scale = None
        for wv, bv in zip(ws, bs):
            proj = torch.nn.Conv2d(in_channels=C, out_channels=1, kernel_size=3, padding=1, stride=1, bias=True)
            proj.load_state_dict({"weight": T(wv), "bias": T(bv)})
            with torch.no_grad():
                sc = proj(grid).reshape(mdesc1.shape[0], -1, h * w)
                sc = torch.exp(sigmoid(sc) * math.log(256.0) - math.log(256.0) / 2)
            scale = sc if scale is None else scale * sc
        arrs["scale_%s" % tag] = scale
        arrs["scale_%s_checksum" % tag] = synth.checksum(npy(mdesc1), ws[0])
    save("heads.npz", **arrs)


def gen_dropin(R):
    """What tests/test_gpu_parity.py::test_dropin_runs_a_gnn_module_on_the_hip_kernels compares against: the REFERENCE's own
    AttentionalGNN / AttentionalPropagation / KeypointEncoder instances (models/modules.py:70-134), parameters from
    pats_amd.synth, forward in eval and train mode, and after the two parameter changes the drop-in's caches must follow
    (a checkpoint loaded later; an in-place update).  Outputs sampled (8192 entries) with their sums: data only."""
    C, names = 128, ["self", "cross", "self"]
    gnn = R.M.AttentionalGNN(C, names)
    ps = [synth.gnn_params(seed=synth.SEED + 120 + i, C=C) for i in range(3)]
    other = [synth.gnn_params(seed=synth.SEED + 130 + i, C=C) for i in range(3)]

    def load(plist):
        for lyr, p_ in zip(gnn.layers, plist):
            lyr.load_state_dict({k: T(v) for k, v in p_.items()}, strict=False)
    a = synth.gnn_inputs(seed=synth.SEED + 125, b=6, C=C, n=65)
    d0, d1 = T(a["x"]), T(a["source"])
    rng = np.random.default_rng(93)
    idx = sample_idx(rng, d0.shape, 8192)
    arrs = {"idx": idx}

    def put(tag, y0, y1):
        arrs.update({tag + "_d0": y0.reshape(-1)[T(idx)], tag + "_d1": y1.reshape(-1)[T(idx)],
                     tag + "_sum": torch.stack([y0.double().sum(), y1.double().sum()])})
    with torch.no_grad():
        load(ps)
        put("eval", *gnn.eval()(d0, d1))
        one = gnn.layers[0].eval()(d0, d1)
        arrs.update(one=one.reshape(-1)[T(idx)], one_sum=one.double().sum())
        put("train", *gnn.train()(d0, d1))
        load(other)                                                   # "a checkpoint loaded after the first forward"
        put("load", *gnn.eval()(d0, d1))
        gnn.layers[1].attn.merge.weight.mul_(1.5)                     # "an in-place optimiser-style update"
        put("step", *gnn.eval()(d0, d1))
    kp = synth.kenc_params(seed=synth.SEED + 140, feature_dim=C)
    kenc = R.M.KeypointEncoder(C, [32, 64, 128, 256, 512])
    kpts = T(synth.grid_kpts(8, 8))
    for mode in ("eval", "train"):
        kenc.load_state_dict({k: T(v) for k, v in kp.items()}, strict=False)
        kenc.train(mode == "train")
        with torch.no_grad():
            arrs["kenc_" + mode] = kenc(kpts)
    arrs["in_checksum"] = synth.checksum(a["x"], a["source"], ps[0]["mlp.0.weight"], other[2]["mlp.3.weight"], kp["encoder.0.weight"])
    save("dropin_gnn.npz", **arrs)


def gen_positions_ranges(R):
    """a9, utils/utils.py:1527-1537: the two index tables themselves for the four grids of the path (15x20 and its portrait twin,
    the fine level's 12x12, YFCC's 24x32), and what the reference's expansion returns when it is handed a DIFFERENT `ranges`
    (every row shifted by one: [1 .. i + 1, 1e7 ...]) - the proof that the tensor is an input the reference honours."""
    out = {}
    for h, w in ((15, 20), (20, 15), (12, 12), (24, 32)):
        pos, rng = R.U.Compute_positions_and_ranges(h, w, 'cpu')
        out["positions_%dx%d" % (h, w)], out["ranges_%dx%d" % (h, w)] = pos, rng
    f = synth.fine_inputs(seed=synth.SEED + 1, B=2)
    Z = R.M.log_optimal_transport2(cost(T(f["d0"]), T(f["d1"]), 264), torch.tensor(1.0), T(f["scale_x"] * f["scale_y"]), 100)
    pos, rng = R.U.Compute_positions_and_ranges(12, 12, 'cpu')
    wrong = torch.where(rng < 1e6, rng + 1.0, rng)
    lim = torch.tensor([0, 12, 0, 12])
    sx, sy = T(f["scale_x"]).reshape(2, -1, 1), T(f["scale_y"]).reshape(2, -1, 1)
    good = R.U.Iterative_expand_matrix(Z.exp(), sx, sy, lim, rng, pos, height=12, width=12, iter_num=8, lower_bound=1e-3)
    bad = R.U.Iterative_expand_matrix(Z.exp(), sx, sy, lim, wrong, pos, height=12, width=12, iter_num=8, lower_bound=1e-3)
    out.update(wrong_ranges_12x12=wrong, bound_canonical=good[5], bound_wrong_ranges=bad[5],
               wrong_changes_bound=np.bool_(not torch.equal(good[5], bad[5])))
    save("positions_ranges.npz", **out)


def gen_roofline(R):
    """BASELINE.json configs[4] through the reference itself: cost einsum at [1,448,4096]^2, then
    log_optimal_transport on 4097x4097 with 200 iterations (modules.py:145-162; ~10 s on 8 cores).  Stored:
    sampled scores and log-plan entries, both argmax vectors, the marginals of exp(Z)."""
    inp = synth.roofline_inputs()
    d0, d1, ns = T(inp["d0"]), T(inp["d1"]), T(inp["ns"])
    S = cost(d0, d1, d0.shape[1])
    Z = R.M.log_optimal_transport(S, torch.tensor(float(inp["alpha"])), ns, 200)
    rng = np.random.default_rng(3)
    si, zi = sample_idx(rng, S.shape, 8192), sample_idx(rng, Z.shape, 16384)
    E = Z.double().exp()
    save("roofline_4097.npz", in_checksum=synth.checksum(inp["d0"][:, :, :64], inp["ns"]), iters=np.int64(200),
         S_idx=si, S_val=S.reshape(-1)[T(si)], Z_idx=zi, Z_val=Z.reshape(-1)[T(zi)],
         max0=Z.max(2).indices[0], max1=Z.max(1).indices[0], row_mass=E.sum(2)[0], col_mass=E.sum(1)[0],
         Z_last_row=Z[0, -1, ::8], Z_last_col=Z[0, ::8, -1])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = ref_import.load()
    only = [a for a in sys.argv[1:] if not a.startswith("-")]     # e.g. `make_golden.py big` regenerates only the
    if only == ["big"]:                                           # bench-size fixtures added in round 2
        gen_pipeline(R, "pipeline_640x480_outdoor.npz", synth.SEED + 50, 15, 20, True, True, True)
        gen_pipeline(R, "pipeline_640x480_indoor.npz", synth.SEED + 51, 15, 20, False, False, False)
        gen_roofline(R)
        return
    if only == ["demo"]:                                          # round 6: the demo's coarse problem (demo.py:36: long side 1 600 -> 38x50 grid)
        run_coarse(R, synth.coarse_inputs(seed=synth.SEED + 22, h=38, w=50), 1216, 1600, "coarse_1901.npz", full=False, with_imgs=False)
        return
    if only == ["a9"]:                                            # round 6: the index tables of Compute_positions_and_ranges
        gen_positions_ranges(R)
        return
    if only == ["gnn"]:
        gen_gnn(R)
        return
    if only == ["heads"]:
        gen_heads(R)
        return
    if only == ["dropin"]:                                        # round 4: the drop-in test's fixture
        gen_dropin(R)
        return
    if only == ["ring"]:                                          # round 3: the a16 gather on the border ring of the cell grid
        gen_third_desc(R, "third_desc_ring.npz", synth.third_maps_ring())
        return
    gen_kat(R)
    gen_sinkhorn_raw(R)
    gen_ties(R)
    run_coarse(R, synth.coarse_inputs(), 480, 640, "coarse_301.npz")
    run_coarse(R, synth.coarse_inputs(seed=synth.SEED + 20, h=20, w=15), 640, 480,
               "coarse_portrait.npz", full=True, with_imgs=False)
    run_coarse(R, synth.coarse_inputs(seed=synth.SEED + 21, h=24, w=32), 768, 1024,
               "coarse_769.npz", full=False, with_imgs=False)
    run_coarse(R, synth.coarse_inputs(seed=synth.SEED + 22, h=38, w=50), 1216, 1600, "coarse_1901.npz", full=False, with_imgs=False)
    gen_fine(R, "fine_145.npz", 6, True, synth.SEED + 1)
    gen_fine(R, "fine_145_indoor.npz", 2, False, synth.SEED + 31)
    gen_third(R, "third_65.npz", 32, True, synth.SEED + 2)
    gen_third(R, "third_65_indoor.npz", 8, False, synth.SEED + 32)
    gen_resize_small(R)
    gen_fine_desc(R)
    gen_third_desc(R)
    gen_third_desc(R, "third_desc_ring.npz", synth.third_maps_ring())
    gen_merge(R, "merge_new.npz", True, synth.SEED + 7)
    gen_merge(R, "merge_old.npz", False, synth.SEED + 9)
    gen_merge(R, "merge_new_portrait.npz", True, synth.SEED + 10, h=20, w=15)
    gen_result(R, "result.npz", synth.SEED + 8, False)
    gen_result(R, "result_mixed.npz", synth.SEED + 11, True)
    gen_positions_ranges(R)
    gen_attention(R)
    gen_gnn(R)
    gen_heads(R)
    gen_dropin(R)
    gen_pipeline(R, "pipeline_outdoor.npz", synth.SEED + 40, 5, 6, True, True, True)
    gen_pipeline(R, "pipeline_indoor.npz", synth.SEED + 41, 4, 5, False, False, False)
    gen_pipeline(R, "pipeline_640x480_outdoor.npz", synth.SEED + 50, 15, 20, True, True, True)
    gen_pipeline(R, "pipeline_640x480_indoor.npz", synth.SEED + 51, 15, 20, False, False, False)
    gen_roofline(R)


if __name__ == "__main__":
    main()
