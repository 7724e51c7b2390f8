#!/usr/bin/env python3
"""AttentionalPropagation (modules.py:107-117) at the path's three shapes: the HIP composition (six fp32 MFMA GEMM
launches + the attention kernel, no cat, BN folded into the last GEMM's operand staging) against the same layer
written in stock PyTorch (nn.Conv1d / BatchNorm1d / einsum + softmax, i.e. what the reference executes) on the same GPU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from pats_amd import ops, synth


def stock_layer(C, params, dev):
    """the reference's module structure restated with stock torch.nn (weights from the same dict)"""
    class MHA(nn.Module):
        def __init__(s):
            super().__init__()
            s.merge = nn.Conv1d(C, C, 1); s.proj = nn.ModuleList([nn.Conv1d(C, C, 1) for _ in range(3)])
        def forward(s, q, k, v):
            b = q.size(0)
            q, k, v = [l(x).view(b, C // 4, 4, -1) for l, x in zip(s.proj, (q, k, v))]
            sc = torch.einsum('bdhn,bdhm->bhnm', q, k) / (C // 4) ** .5
            x = torch.einsum('bhnm,bdhm->bdhn', torch.softmax(sc, dim=-1), v)
            return s.merge(x.contiguous().view(b, C, -1))
    class Prop(nn.Module):
        def __init__(s):
            super().__init__()
            s.attn = MHA(); s.mlp = nn.Sequential(nn.Conv1d(2 * C, 2 * C, 1), nn.BatchNorm1d(2 * C), nn.ReLU(), nn.Conv1d(2 * C, C, 1))
        def forward(s, x, src):
            return s.mlp(torch.cat([x, s.attn(x, src, src)], dim=1))
    m = Prop()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    return m.to(dev).eval()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dev = torch.device("cuda")
for C, b, n, tag in ((128, 25920, 65, "third level, one pair (25 920 x 65 tokens)"), (264, 432, 145, "fine level, one pair (432 x 145 tokens)"),
                     (448, 16, 300, "coarse level, 16 pairs (300 tokens)")):
    params = synth.gnn_params(seed=3, C=C)
    P = ops.PropagationParams(params)
    x = torch.randn((b, C, n), device=dev); s = torch.randn((b, C, n), device=dev)
    m = stock_layer(C, params, dev)
    with torch.no_grad():
        ref = m(x[:64], s[:64]); got = ops.attentional_propagation(x[:64].contiguous(), s[:64].contiguous(), P)
        err = (ref - got).abs().max().item()
        t_hip = timeit(lambda: ops.attentional_propagation(x, s, P))
        t_ref = timeit(lambda: m(x, s))
        # BatchNorm on batch statistics: the mode the third layer runs in under PATS.eval() (pats.py:112-120)
        t_hip_tr = timeit(lambda: ops.attentional_propagation(x, s, P, bn_train=True))
        m.train()
        t_ref_tr = timeit(lambda: m(x, s))
        m.eval()
    flops = b * n * (2.0 * C * C * 4 + 2.0 * 2 * C * 2 * C + 2.0 * 2 * C * C) + b * 4 * (2.0 * n * n * (C // 4)) * 2
    print(json.dumps({"shape": tag, "C": C, "hip_ms": t_hip, "stock_pytorch_ms": t_ref, "speedup": t_ref / t_hip,
                      "hip_tflops": flops / t_hip / 1e9, "max_abs_diff_vs_stock": err,
                      "bn_train_hip_ms": t_hip_tr, "bn_train_stock_pytorch_ms": t_ref_tr, "bn_train_speedup": t_ref_tr / t_hip_tr}))
