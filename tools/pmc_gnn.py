#!/usr/bin/env python3
"""A few launches of the fused AttentionalPropagation layer (gnn_fused.hip) at the third level's shape for rocprofv3 passes
(25 920 x [128, 65]; BN=train stops the kernel behind mlp[0]); B / C / NTOK select another shape, e.g. the fine level 4096 x [264, 145]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pats_amd import ops, synth
b, C, n = int(os.environ.get("B", "25920")), int(os.environ.get("C", "128")), int(os.environ.get("NTOK", "65"))
P = ops.PropagationParams(synth.gnn_params(seed=3, C=C))
x = torch.randn((b, C, n), device="cuda"); s = torch.randn((b, C, n), device="cuda")
train = os.environ.get("BN", "eval") == "train"
for _ in range(int(os.environ.get("N", "4"))):
    ops.attentional_propagation(x, s, P, bn_train=train, residual=x)
torch.cuda.synchronize()
print("done")
