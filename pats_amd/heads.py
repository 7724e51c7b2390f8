"""What the three layers compute BETWEEN their backbone and their optimal-transport problem, on pats_amd.ops -
ready-made `nets` callbacks for pipeline.forward_path.

    CoarseHeads   models/first_layer.py:74-107    grid keypoints -> KeypointEncoder, + descriptors, AttentionalGNN,
                                                  final_proj, scale head            -> mdesc0, mdesc1, scale, |bin_score|
    FineHeads     models/second_layer.py:71-97    descriptor sampling (+ title / dustbin feature), AttentionalGNN,
                                                  final_proj, two scale heads       -> mdesc0, mdesc1, scale_x, scale_y
    ThirdHeads    models/third_layer.py:121-152   8x8 window gather + KeypointEncoder + dustbin feature, AttentionalGNN,
                                                  scale head                        -> feat0, feat1, scale

The backbones (ResNet / FPN, the `compress_*` convolutions on their outputs) stay outside: each class takes what the
backbone hands over.  Every step is a C-ABI call (KeypointEncoder: six GEMMs; a GNN layer: six GEMMs + the attention
kernel; final_proj: one GEMM; the scale head: one stencil kernel); parameters come from the reference modules' own
state_dicts (ops.MLPParams / ops.PropagationParams) and stay on the device.
"""
import torch

from . import ops


_GRID_CACHE = {}


def grid_kpts(h, w, device):
    """first_layer.py:74-79 / third_layer.py:132-136: [:,0] = row / h, [:,1] = column / w, row-major, float32.  Built on the
    HOST like the reference builds it (`torch.arange(...) / float(h)` runs on the CPU there and only the result moves to
    the device: the CPU division is correctly rounded, the device's tensor-by-scalar division is a reciprocal multiply
    and lands one ulp off for e.g. 5 / 6), once per (h, w, device)."""
    key = (int(h), int(w), str(device))
    kpts = _GRID_CACHE.get(key)
    if kpts is None:
        cols = torch.arange(0, h).reshape(h, 1).repeat(1, w).reshape(-1) / float(h)
        rows = torch.arange(0, w).reshape(1, w).repeat(h, 1).reshape(-1) / float(w)
        kpts = torch.zeros((h * w, 2))
        kpts[:, 0] = cols
        kpts[:, 1] = rows
        kpts = _GRID_CACHE[key] = kpts.to(device)
    return kpts


class CoarseHeads:
    """first_layer.py:74-107.  kenc: ops.MLPParams; gnn: [ops.PropagationParams] with `names`; final_proj / scalex_proj:
    (weight, bias) GPU tensors; bin_score: float or 0-d tensor."""

    def __init__(self, kenc, gnn, names, final_proj, scalex_proj, bin_score=0.0, heads=4):
        self.kenc, self.gnn, self.names, self.final_proj, self.scalex_proj = kenc, gnn, list(names), final_proj, scalex_proj
        self.bin_score, self.heads = bin_score, heads

    def __call__(self, desc_left, desc_right):
        """desc_*: [b,448,h,w] (the concatenated compress_0/1/2 maps, first_layer.py:73,92) -> mdesc0, mdesc1 [b,448,h*w],
        scale [b,1,h*w], alpha."""
        b, C, h, w = desc_left.shape
        k = ops.keypoint_encoder(grid_kpts(h, w, desc_left.device), self.kenc)                  # [1,C,h*w]  (:81, :99)
        desc0 = (desc_left.reshape(b, C, h * w) + k).contiguous()
        desc1 = (desc_right.reshape(b, C, h * w) + k).contiguous()
        desc0, desc1 = ops.attentional_gnn(desc0, desc1, self.gnn, self.names, heads=self.heads)     # :102
        mdesc0 = ops.conv1d(desc0, *self.final_proj)                                                # :105
        mdesc1 = ops.conv1d(desc1, *self.final_proj)
        scale = ops.scale_head(mdesc1, h, w, [self.scalex_proj[0]], [self.scalex_proj[1]])         # :106-107
        alpha = self.bin_score.abs() if isinstance(self.bin_score, torch.Tensor) else abs(float(self.bin_score))
        return mdesc0, mdesc1, scale, alpha


class FineHeads:
    """second_layer.py:71-97 after ResNet2.forward2 and the two `compress` convolutions on the coarse descriptor."""

    def __init__(self, gnn, names, final_proj, scalex_proj, scaley_proj, heads=4):
        self.gnn, self.names, self.final_proj = gnn, list(names), final_proj
        self.scalex_proj, self.scaley_proj, self.heads = scalex_proj, scaley_proj, heads

    def __call__(self, desc0_, title, rubbish):
        """desc0_: the three forward2 maps of the 2B stacked crops; title [B,8], rubbish [B,264] (compress_1 / compress_2 of
        desc_l, :82-83) -> mdesc0, mdesc1 [B,264,145], scale_x, scale_y [B,1,144]."""
        desc = ops.fine_descriptors(desc0_, title, rubbish)                                          # :71-86
        desc0, desc1 = ops.attentional_gnn(desc[0], desc[1], self.gnn, self.names, heads=self.heads)   # :89
        mdesc0 = ops.conv1d(desc0, *self.final_proj)                                                 # :91
        mdesc1 = ops.conv1d(desc1, *self.final_proj)
        _, (sx, sy) = ops.scale_head(mdesc1, 12, 12, [self.scalex_proj[0], self.scaley_proj[0]],
                                     [self.scalex_proj[1], self.scaley_proj[1]], return_heads=True)    # :92-97
        return mdesc0, mdesc1, sx.contiguous(), sy.contiguous()


class ThirdHeads:
    """third_layer.py:121-152 after ResNet2 and the `compress` of the fine features."""

    def __init__(self, kenc, gnn, names, scale_proj, heads=4, bn_train=False):
        self.kenc, self.gnn, self.names, self.scale_proj, self.heads, self.bn_train = kenc, gnn, list(names), scale_proj, heads, bn_train

    def __call__(self, feat_f0, feat_f1, mkpts0_c, mkpts1_c, b_ids, rubbish):
        """feat_f*: [B,128,52,52] padded half-resolution maps; mkpts*_c [P,2]; b_ids [P]; rubbish [B,128,144] ->
        feat0, feat1 [P,128,65], scale [P,1,64], and the points on the 4-px lattice."""
        k = ops.keypoint_encoder(grid_kpts(8, 8, feat_f0.device), self.kenc, bn_train=self.bn_train)     # :132-140
        f0, f1, ps, pt = ops.third_descriptors(feat_f0, feat_f1, mkpts0_c, mkpts1_c, b_ids, k.reshape(128, 64), rubbish)
        f0, f1 = ops.attentional_gnn(f0, f1, self.gnn, self.names, heads=self.heads, bn_train=self.bn_train)   # :148
        scale = ops.scale_head(f1, 8, 8, [self.scale_proj[0]], [self.scale_proj[1]])                     # :151-152
        return f0, f1, scale, ps, pt
