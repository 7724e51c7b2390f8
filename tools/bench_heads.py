#!/usr/bin/env python3
"""What a layer computes between its backbone and its OT problem (pats_amd.heads), timed at one pair's sizes with the
reference's layer counts (first_layer.py:14: 18 GNN layers; second_layer.py: 18; third_layer.py:91: 10), random weights.
One JSON line per level: ms per pair, split into KeypointEncoder / gather, GNN, final_proj, scale head."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pats_amd import ops, heads, synth

def cu(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def gnn(C, L):
    return [ops.PropagationParams(synth.gnn_params(seed=100 + i, C=C)) for i in range(L)], (["self", "cross"] * L)[:L]

g = torch.Generator(device="cuda"); g.manual_seed(0)
# ---- coarse: one pair, 15 x 20 grid, 448 channels, 18 layers --------------------------------------------------------------
C, h, w = 448, 15, 20
layers, names = gnn(C, 18)
fp = synth.final_proj_params(C=C); (sw,), (sb,) = synth.scale_head_params(C=C)
H1 = heads.CoarseHeads(ops.MLPParams(synth.kenc_params(feature_dim=C), prefix="encoder."), layers, names, (cu(fp["weight"]), cu(fp["bias"])), (cu(sw), cu(sb)))
dl, dr = torch.randn((1, C, h, w), device="cuda", generator=g), torch.randn((1, C, h, w), device="cuda", generator=g)
d0 = dl.reshape(1, C, -1).contiguous()
print(json.dumps({"level": "coarse (first_layer.py:74-107)", "shape": "1 x [448, 300], 18 GNN layers", "ms_per_pair": timed(lambda: H1(dl, dr)),
                  "gnn_ms": timed(lambda: ops.attentional_gnn(d0, d0, layers, names)),
                  "kenc_ms": timed(lambda: ops.keypoint_encoder(heads.grid_kpts(h, w, "cuda"), H1.kenc)),
                  "final_proj_ms": 2 * timed(lambda: ops.conv1d(d0, *H1.final_proj)),
                  "scale_head_ms": timed(lambda: ops.scale_head(d0, h, w, [cu(sw)], [cu(sb)]))}))
# ---- fine: 432 crops per pair, 264 channels, 18 layers ----------------------------------------------------------------------
B, C = 432, 264
layers, names = gnn(C, 18)
fp = synth.final_proj_params(C=C); (sxw, syw), (sxb, syb) = synth.scale_head_params(C=C, heads=2)
H2 = heads.FineHeads(layers, names, (cu(fp["weight"]), cu(fp["bias"])), (cu(sxw), cu(sxb)), (cu(syw), cu(syb)))
fm = [torch.randn(s, device="cuda", generator=g) for s in ((2 * B, 64, 48, 48), (2 * B, 64, 24, 24), (2 * B, 128, 12, 12))]
title, rub = torch.randn((B, 8), device="cuda", generator=g), torch.randn((B, 264), device="cuda", generator=g)
d0 = torch.randn((B, C, 145), device="cuda", generator=g)
print(json.dumps({"level": "fine (second_layer.py:71-97)", "shape": "432 x [264, 145], 18 GNN layers", "ms_per_pair": timed(lambda: H2(fm, title, rub)),
                  "gnn_ms": timed(lambda: ops.attentional_gnn(d0, d0, layers, names)),
                  "gather_ms": timed(lambda: ops.fine_descriptors(fm, title, rub)),
                  "final_proj_ms": 2 * timed(lambda: ops.conv1d(d0, *H2.final_proj)),
                  "scale_head_ms": timed(lambda: ops.scale_head(d0, 12, 12, [cu(sxw), cu(syw)], [cu(sxb), cu(syb)]))}))
# ---- third: 25 920 windows per pair, 128 channels, 10 layers, BatchNorm on batch statistics --------------------------------
P, B3, C = 25920, 432, 128
layers, names = gnn(C, 10)
(s3w,), (s3b,) = synth.scale_head_params(C=C)
H3 = heads.ThirdHeads(ops.MLPParams(synth.kenc_params(feature_dim=C), prefix="encoder."), layers, names, (cu(s3w), cu(s3b)), bn_train=True)
ff0, ff1 = torch.randn((B3, 128, 52, 52), device="cuda", generator=g), torch.randn((B3, 128, 52, 52), device="cuda", generator=g)
mk0 = (torch.randint(1, 11, (P, 2), device="cuda") * 8 + 4).float(); mk1 = torch.randint(8, 185, (P, 2), device="cuda").float() * 0.5
bid = torch.randint(0, B3, (P,), device="cuda"); rub3 = torch.randn((B3, 128, 144), device="cuda", generator=g)
d0 = torch.randn((P, C, 65), device="cuda", generator=g)
print(json.dumps({"level": "third (third_layer.py:121-152)", "shape": "25920 x [128, 65], 10 GNN layers, BatchNorm on batch statistics",
                  "ms_per_pair": timed(lambda: H3(ff0, ff1, mk0, mk1, bid, rub3)),
                  "gnn_ms": timed(lambda: ops.attentional_gnn(d0, d0, layers, names, bn_train=True)),
                  "gather_ms": timed(lambda: ops.third_descriptors(ff0, ff1, mk0, mk1, bid, torch.zeros((128, 64), device="cuda"), rub3)),
                  "scale_head_ms": timed(lambda: ops.scale_head(d0, 8, 8, [cu(s3w)], [cu(s3b)]))}))
