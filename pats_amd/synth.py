"""Deterministic synthetic inputs at the reference's real shapes.

No weights or datasets exist for the reference here, so every workload is synthetic: descriptors
are `3*(base + 0.3*eps)` so the diagonal logit (after the reference's `0.1/sqrt(D)` factor,
/root/reference/models/first_layer.py:110-114) beats the dustbin marginal, target areas follow the
reference's scale head `exp(sigmoid(x)*ln256 - ln256/2)` (/root/reference/models/first_layer.py:106).
All draws come from numpy's PCG64 `default_rng(seed)` in float32, in a fixed order, so goldens,
tests and bench.py regenerate identical tensors (fixtures additionally store an input checksum).
Seed 18027 is the reference configs' seed (/root/reference/configs/test_megadepth.yaml `seed`).
"""
import math

import numpy as np

SEED = 18027
LN256 = math.log(256.0)


def _scale_head(rng, shape, spread=0.3):
    x = spread * rng.standard_normal(shape, dtype=np.float32)
    sig = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    return np.exp(sig * LN256 - LN256 / 2).astype(np.float32)


def _desc_pair(rng, shape, amp=3.0, noise=0.3, drop=0.0):
    base = rng.standard_normal(shape, dtype=np.float32)
    e0 = rng.standard_normal(shape, dtype=np.float32)
    e1 = rng.standard_normal(shape, dtype=np.float32)
    d0 = (amp * (base + noise * e0)).astype(np.float32)
    d1 = (amp * (base + noise * e1)).astype(np.float32)
    if drop > 0.0:
        # "dropped" source patches have no counterpart: their descriptor is fresh noise, so the
        # transport sends them to the dustbin (exercises the no-match paths downstream)
        fresh = rng.standard_normal(shape, dtype=np.float32)
        gone = rng.random((shape[0], 1, shape[2])) < drop
        d0 = np.where(gone, (amp * 1.04 * fresh).astype(np.float32), d0)
    return d0, d1


def coarse_inputs(seed=SEED, h=15, w=20, D=448, b=1):
    """L1 (a1,a4): mdesc0/mdesc1 [b,D,N], ns [b,1,N] target areas, alpha = |bin_score| (0-d)."""
    rng = np.random.default_rng(seed)
    n = h * w
    d0, d1 = _desc_pair(rng, (b, D, n))
    ns = _scale_head(rng, (b, 1, n))
    alpha = np.float32(0.0)
    return {"d0": d0, "d1": d1, "ns": ns, "alpha": alpha, "h": h, "w": w}


def fine_inputs(seed=SEED + 1, B=6, D=264, n=145):
    """L2 (a2,a5): mdesc [B,D,145] (last column = dustbin feature), scale_x, scale_y [B,1,144]."""
    rng = np.random.default_rng(seed)
    d0, d1 = _desc_pair(rng, (B, D, n), drop=0.12)
    # keep the dustbin feature column moderate so both matches and non-matches occur
    d0[:, :, -1] *= 0.5
    d1[:, :, -1] *= 0.5
    sx = _scale_head(rng, (B, 1, n - 1))
    sy = _scale_head(rng, (B, 1, n - 1))
    return {"d0": d0, "d1": d1, "scale_x": sx, "scale_y": sy}


def third_inputs(seed=SEED + 2, P=32, D=128, W=8):
    """L3 (a3,a5,a17): feat [P,128,65], scale [P,1,64], coarse points p_s,p_t [P,2] (multiples of 4)."""
    rng = np.random.default_rng(seed)
    n = W * W + 1
    d0, d1 = _desc_pair(rng, (P, D, n), drop=0.12)
    d0[:, :, -1] *= 0.5
    d1[:, :, -1] *= 0.5
    scale = _scale_head(rng, (P, 1, n - 1))
    p_s = (rng.integers(1, 23, size=(P, 2)) * 4).astype(np.int64)
    p_t = (rng.integers(0, 25, size=(P, 2)) * 4).astype(np.int64)
    return {"d0": d0, "d1": d1, "scale": scale, "p_s": p_s, "p_t": p_t}


def image_pair(seed=SEED + 3, H=480, W=640):
    """Two uint8-valued HWC images as float32 [1,H,W,3] (what evaluate.py hands to the model)."""
    rng = np.random.default_rng(seed)
    left = rng.integers(0, 256, size=(1, H, W, 3)).astype(np.float32)
    right = rng.integers(0, 256, size=(1, H, W, 3)).astype(np.float32)
    # smooth a little along x so bilinear taps are not pure noise
    right = (0.5 * right + 0.5 * np.roll(right, 1, axis=2)).astype(np.float32)
    return left, right


def roofline_inputs(seed=SEED + 4, N=4096, D=448):
    """Config 5: d0,d1 ~ N(0,1) [1,D,N], ns ~ U(0.5,2) [1,1,N]."""
    rng = np.random.default_rng(seed)
    d0 = rng.standard_normal((1, D, N), dtype=np.float32)
    d1 = rng.standard_normal((1, D, N), dtype=np.float32)
    ns = rng.uniform(0.5, 2.0, size=(1, 1, N)).astype(np.float32)
    return {"d0": d0, "d1": d1, "ns": ns, "alpha": np.float32(1.0)}


def checksum(*arrays):
    """Order-sensitive float64 checksum used to prove regenerated inputs equal the fixture's."""
    acc = 0.0
    for k, a in enumerate(arrays):
        a = np.ascontiguousarray(a).astype(np.float64).ravel()
        idx = np.arange(1, a.size + 1, dtype=np.float64)
        acc += (k + 1) * float(np.dot(a, np.cos(idx * 0.37)))
    return acc


def fine_maps(seed=SEED + 5, B=3):
    """a15 inputs: the three ResNet2.forward2 maps of the stacked left|right crops, the 8-channel
    title (compress_1 output) and the dustbin feature (compress_2 output)."""
    rng = np.random.default_rng(seed)
    f0 = rng.standard_normal((2 * B, 64, 48, 48), dtype=np.float32)
    f1 = rng.standard_normal((2 * B, 64, 24, 24), dtype=np.float32)
    f2 = rng.standard_normal((2 * B, 128, 12, 12), dtype=np.float32)
    title = rng.standard_normal((B, 8, 1), dtype=np.float32)
    rubbish = rng.standard_normal((B, 264, 1), dtype=np.float32)
    return {"f0": f0, "f1": f1, "f2": f2, "title": title, "rubbish": rubbish}


def third_maps(seed=SEED + 6, B=3, P=40):
    """a16 inputs: FPN maps [B,128,52,52], coarse points in crop pixels (source on the 8c+4 lattice as
    pats.py:57-61 builds them, targets arbitrary halves/quarters to exercise round-half-even), kenc, rubbish."""
    rng = np.random.default_rng(seed)
    ff0 = rng.standard_normal((B, 128, 52, 52), dtype=np.float32)
    ff1 = rng.standard_normal((B, 128, 52, 52), dtype=np.float32)
    # source cells 1..10 only: PATS masks the border ring (second_layer.py:140-149), and the reference's
    # own dustbin index round(mk/8) leaves the 12x12 map for cell 11
    mk0 = (rng.integers(1, 11, size=(P, 2)) * 8 + 4).astype(np.float32)
    mk1 = (rng.integers(8, 185, size=(P, 2)) * 0.5).astype(np.float32)        # 4.0 .. 92.0 in steps of 0.5
    mk1[:6] = np.array([[10, 6], [14, 18], [22, 26], [6, 90], [90.5, 4.0], [50, 2 * 23 + 0.0]], np.float32)
    b_ids = rng.integers(0, B, size=(P,)).astype(np.int64)
    kenc = rng.standard_normal((1, 128, 64), dtype=np.float32)
    rubbish = rng.standard_normal((B, 128, 144), dtype=np.float32)
    return {"ff0": ff0, "ff1": ff1, "mk0": mk0, "mk1": mk1, "b_ids": b_ids, "kenc": kenc, "rubbish": rubbish}


def third_maps_ring(seed=SEED + 60, B=4, P=48):
    """a16 inputs with source points on the BORDER RING of the 12x12 cell grid (cells 0 and 11: 8c+4 = 4 and 92), where
    the reference's window leaves the 52x52 map on one side and wraps into the neighbouring row / image of the flattened
    NHWC view (third_layer.py:127) and its dustbin index round(92 / 8) = 12 reads the NEXT patch's feature (:141-144).
    Patches 1 .. B-2 only: from the first / last patch those indices leave the tensor and torch.gather raises."""
    rng = np.random.default_rng(seed)
    d = third_maps(seed=seed, B=B, P=P)
    cells = rng.integers(0, 12, size=(P, 2))
    ring = rng.integers(0, 2, size=(P, 2)) * 11
    pick = rng.integers(0, 3, size=P)                        # ring in x, ring in y, ring in both
    cells[pick != 1, 0] = ring[pick != 1, 0]
    cells[pick != 0, 1] = ring[pick != 0, 1]
    cells[:4] = [[11, 11], [0, 0], [11, 0], [0, 11]]
    d["mk0"] = (cells * 8 + 4).astype(np.float32)
    d["mk1"] = (rng.integers(-8, 209, size=(P, 2)) * 0.5).astype(np.float32)       # -4.0 .. 104.0: clamped to [0, 96] (:128-129)
    d["mk1"][:4] = np.array([[96, 96], [0, 0], [97.5, -3], [-0.5, 200]], np.float32)
    d["b_ids"] = rng.integers(1, B - 1, size=(P,)).astype(np.int64)
    return d


def merge_inputs(seed=SEED + 7, h=15, w=20, chunks=3, tie_step=0.125):
    """Inputs of merge_patches_new/old (second_layer.py:137-238) for `chunks` successive L2 chunks of
    one pair: the coarse no-match mask, and per chunk the chunk mask (first_layer.py:137-139 style:
    a contiguous run of matched patches, neighbouring chunks overlapping by one grid row), the L2
    trust scores [B,144] and L2 no-match flags [B,144].  Half of the trust values sit on a 1/8 grid so
    that exact ties reach the argsort."""
    rng = np.random.default_rng(seed)
    N = h * w
    ifn_all = rng.random((1, N)) < 0.12
    cum = np.cumsum(~ifn_all, axis=1)
    K = int(cum[0, -1])
    edges = np.linspace(0, K, chunks + 1).astype(int)
    out = []
    for c in range(chunks):
        lo = max(0, edges[c] - (w if c else 0))
        hi = edges[c + 1]
        ifn_L1 = ifn_all | (cum <= lo) | (cum > hi)
        B = int((~ifn_L1).sum())
        trust = rng.lognormal(-1.0, 0.9, (B, 144)).astype(np.float32)
        q = rng.random((B, 144)) < 0.5
        trust = np.where(q, np.maximum(tie_step, np.round(trust / tie_step) * tie_step), trust).astype(np.float32)
        ifn2 = rng.random((B, 144)) < 0.25
        out.append({"ifn_L1": ifn_L1, "trust": trust, "ifn2": ifn2})
    return {"h": h, "w": w, "ifn_all": ifn_all, "chunks": out}


def result_inputs(seed=SEED + 8, h=15, w=20, bs=1, mixed_choice=False):
    """Inputs of get_result / the third-level scatter (utils.py:189-213, pats.py:53-78): L1 cells with
    their re-centred points and scales, L2 points / flags per surviving cell, third-level outputs."""
    rng = np.random.default_rng(seed)
    N = h * w
    ifn0 = rng.random((bs, N)) < 0.15
    K = int((~ifn0).sum())
    ap0 = (rng.uniform(1.0, [h - 1.0, w - 1.0], (bs, N, 2))).astype(np.float32)
    sc0 = np.exp(rng.uniform(-1.2, 1.2, (bs, N, 2))).astype(np.float32)
    ifn2 = rng.random((K, 144)) < 0.55
    ifn2[K // 3] = True                                   # a row with nothing left (pats.py:41-42)
    pts = rng.uniform(0.0, 12.0, (K, 144, 2)).astype(np.float32)
    pts[0, :8] = np.array([[0.125, 0.375], [0.625, 1.125], [2.875, 3.375], [5.5, 5.5], [6.125, 0.0],
                           [11.875, 11.625], [1.375, 2.625], [3.0, 3.0]], np.float32)   # round-half-even cases
    P = int((~ifn2).sum())
    mkpts1 = rng.uniform(0.0, 96.0, (P, 16, 2)).astype(np.float32)
    label0 = np.where(rng.random((P * 16,)) < 0.2, -10.0, 1e8).astype(np.float32)
    ch0 = np.ones((bs,), bool)
    ch1 = np.ones((K,), bool)
    if mixed_choice:
        ch1 = rng.random((K,)) < 0.5
    return {"h": h, "w": w, "ifn0": ifn0, "ap0": ap0, "sc0": sc0, "ifn2": ifn2, "pts": pts, "mkpts1": mkpts1,
            "label0": label0, "choice0": ch0, "choice1": ch1}


def attention_inputs(seed=SEED + 12, b=3, dim=32, heads=4, n=65, m=None, amp=1.0):
    """q, k, v of modules.py:84 ([b, dim, heads, tokens], the view MultiHeadedAttention makes of its
    projections, :101-102).  amp scales q and k: larger values give peaked softmax rows."""
    rng = np.random.default_rng(seed)
    m = n if m is None else m
    q = (amp * rng.standard_normal((b, dim, heads, n))).astype(np.float32)
    k = (amp * rng.standard_normal((b, dim, heads, m))).astype(np.float32)
    v = rng.standard_normal((b, dim, heads, m)).astype(np.float32)
    return {"q": q, "k": k, "v": v}


class SynthNets:
    """Deterministic stand-ins for the three networks of PATS (numpy): the tensors a layer's forward holds
    right before its cost build.  Both the reference-side chain of tools/make_golden.py and
    pats_amd.pipeline.forward_path consume them, so the two chains see identical inputs as long as they
    agree on the chunk sizes B and the third-level counts P."""

    def __init__(self, seed=SEED + 40, h=5, w=6):
        self.seed, self.h, self.w = seed, h, w

    def images(self):
        return image_pair(seed=self.seed + 1, H=self.h * 32, W=self.w * 32)

    def coarse(self):
        return coarse_inputs(seed=self.seed, h=self.h, w=self.w)

    def fine(self, num, B):
        return fine_inputs(seed=self.seed + 101 + num, B=B)

    def third(self, num, P):
        return third_inputs(seed=self.seed + 201 + num, P=P)


def gnn_params(seed=SEED + 70, C=128):
    """Weights of one AttentionalPropagation layer (modules.py:107-113) under the reference's state_dict names,
    drawn with numpy so that fixtures, tests and the oracle regenerate them: Conv1d weights ~ U(-1/sqrt(fan_in), +),
    BatchNorm with non-trivial affine and running statistics."""
    rng = np.random.default_rng(seed)

    def conv(cout, cin):
        k = 1.0 / np.sqrt(cin)
        return (rng.uniform(-k, k, (cout, cin, 1)).astype(np.float32), rng.uniform(-k, k, (cout,)).astype(np.float32))
    p = {}
    for i in range(3):
        p["attn.proj.%d.weight" % i], p["attn.proj.%d.bias" % i] = conv(C, C)
    p["attn.merge.weight"], p["attn.merge.bias"] = conv(C, C)
    p["mlp.0.weight"], p["mlp.0.bias"] = conv(2 * C, 2 * C)
    p["mlp.1.weight"] = rng.uniform(0.5, 1.5, (2 * C,)).astype(np.float32)
    p["mlp.1.bias"] = (0.2 * rng.standard_normal((2 * C,))).astype(np.float32)
    p["mlp.1.running_mean"] = (0.3 * rng.standard_normal((2 * C,))).astype(np.float32)
    p["mlp.1.running_var"] = rng.uniform(0.5, 2.0, (2 * C,)).astype(np.float32)
    p["mlp.3.weight"], p["mlp.3.bias"] = conv(C, 2 * C)
    p["mlp.3.bias"][:] = 0.0                                   # nn.init.constant_(self.mlp[-1].bias, 0.0), modules.py:112
    return p


def gnn_inputs(seed=SEED + 71, b=3, C=128, n=65, m=None):
    rng = np.random.default_rng(seed)
    m = n if m is None else m
    return {"x": rng.standard_normal((b, C, n)).astype(np.float32), "source": rng.standard_normal((b, C, m)).astype(np.float32)}


def kenc_params(seed=SEED + 100, feature_dim=128, layers=(32, 64, 128, 256, 512)):
    """Weights of one KeypointEncoder (modules.py:70-76: MLP([2] + layers + [feature_dim])) under the reference's
    state_dict names ("encoder.0.weight", "encoder.1.running_mean", ...), drawn with numpy like gnn_params."""
    rng = np.random.default_rng(seed)
    ch = [2] + list(layers) + [feature_dim]
    p = {}
    for i in range(1, len(ch)):
        k = 1.0 / np.sqrt(ch[i - 1])
        j = 3 * (i - 1)                                # Conv1d, BatchNorm1d, ReLU per hidden layer
        p["encoder.%d.weight" % j] = rng.uniform(-k, k, (ch[i], ch[i - 1], 1)).astype(np.float32)
        p["encoder.%d.bias" % j] = rng.uniform(-k, k, (ch[i],)).astype(np.float32)
        if i < len(ch) - 1:
            p["encoder.%d.weight" % (j + 1)] = rng.uniform(0.5, 1.5, (ch[i],)).astype(np.float32)
            p["encoder.%d.bias" % (j + 1)] = (0.2 * rng.standard_normal((ch[i],))).astype(np.float32)
            p["encoder.%d.running_mean" % (j + 1)] = (0.1 * rng.standard_normal((ch[i],))).astype(np.float32)
            p["encoder.%d.running_var" % (j + 1)] = rng.uniform(0.05, 0.5, (ch[i],)).astype(np.float32)
    p["encoder.%d.bias" % (3 * (len(ch) - 2))][:] = 0.0       # nn.init.constant_(self.encoder[-1].bias, 0.0), modules.py:75
    return p


def grid_kpts(h, w):
    """The keypoint grid the layers feed their encoder (first_layer.py:74-79 with h, w = the descriptor map;
    third_layer.py:132-136 with h = w = 8): [:,0] = row / h, [:,1] = column / w, row-major."""
    cols = (np.arange(h, dtype=np.float32).reshape(h, 1).repeat(w, 1).reshape(-1) / np.float32(h)).astype(np.float32)
    rows = (np.arange(w, dtype=np.float32).reshape(1, w).repeat(h, 0).reshape(-1) / np.float32(w)).astype(np.float32)
    return np.stack([cols, rows], axis=1).astype(np.float32)


def final_proj_params(seed=SEED + 110, C=448):
    """nn.Conv1d(C, C, kernel_size=1, bias=True) (first_layer.py:34-36, second_layer.py:40-42)."""
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(C)
    return {"weight": rng.uniform(-k, k, (C, C, 1)).astype(np.float32), "bias": rng.uniform(-k, k, (C,)).astype(np.float32)}


def scale_head_params(seed=SEED + 120, C=128, heads=1):
    """nn.Conv2d(C, 1, kernel_size=3, padding=1) per head (first_layer.py:39-40, second_layer.py:33-36, third_layer.py:88-89):
    lists (weights [1,C,3,3], biases [1])."""
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(9.0 * C)
    return ([rng.uniform(-k, k, (1, C, 3, 3)).astype(np.float32) for _ in range(heads)],
            [rng.uniform(-k, k, (1,)).astype(np.float32) for _ in range(heads)])
