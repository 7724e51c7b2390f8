#!/usr/bin/env python3
"""Does a conv stack switched to torch.channels_last (dropin.prepare_backbones) hand the descriptor gathers channels-last maps,
and what does the switch cost the stack itself?  A stand-in for ResNet2.forward2's trunk (the reference's backbones need
torchvision weights that are not here): conv7x7/2 + 4 x (conv3x3 + BN + ReLU) at 64 channels on 96x96 crops, stacked
[2B,3,96,96] NCHW-contiguous inputs exactly as second_layer.py:66-69 builds them.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from pats_amd import dropin

def stack():
    layers = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True)]
    for _ in range(4):
        layers += [nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True)]
    return nn.Sequential(*layers).cuda().eval()

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = int(os.environ.get("B", "2048"))
x = torch.randn((2 * B, 3, 96, 96), device="cuda")
res = {"input": "[%d,3,96,96] NCHW-contiguous" % (2 * B), "PYTORCH_MIOPEN_SUGGEST_NHWC": os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC")}
with torch.no_grad():
    m = stack()
    y0 = m(x)
    res["nchw_ms"] = timeit(lambda: m(x))
    res["nchw_output_is_channels_last"] = bool(y0.is_contiguous(memory_format=torch.channels_last))
    class Holder(nn.Module):
        def __init__(s, m):
            super().__init__(); s.descriptor_extract = m
    dropin.prepare_backbones(Holder(m))
    y1 = m(x)
    res["channels_last_ms"] = timeit(lambda: m(x))
    res["channels_last_output_is_channels_last"] = bool(y1.is_contiguous(memory_format=torch.channels_last))
    res["max_abs_diff"] = float((y0 - y1).abs().max())
print(json.dumps(res))
