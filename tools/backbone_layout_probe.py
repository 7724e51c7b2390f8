#!/usr/bin/env python3
"""Do the backbones whose OUTPUTS the two descriptor gathers read hand over channels-last maps once their parameters are
channels-last (dropin.prepare_backbones), and what does the switch cost the backbones themselves under MIOpen?

Round 5 (verdict r04 item 7): no stand-in trunk any more - the two stacks are built here from torch.nn primitives BY SHAPE
(layer list read off models/resnet.py:149-200 and models/third_layer.py:19-77; no forward is copied, weights are random):
  second level  ResNet2(BasicBlock, [3, 4, ..]).forward2 on the stacked 96 x 96 crops (second_layer.py:66-69):
                conv7x7/2 + BN + ReLU -> x0 [.,64,48,48]; maxpool/2 + 3 basic blocks(64) -> x1 [.,64,24,24];
                4 basic blocks(128, first stride 2 with a 1x1 downsample) -> x2 [.,128,12,12]          (the a15 gather's three maps)
  third level   the same forward2 again (third_layer.py:113-115) + FPN_8_2.forward (:63-77): 1x1 / 3x3 convolutions, BatchNorm,
                LeakyReLU, two bilinear x2 upsamplings, zero pads -> [2B,128,52,52]                      (the a16 gather's maps)
at 2 B = 832 crops (one 48-pair step's worst chunk: 416 rows) and NCHW-contiguous inputs exactly as the reference builds them.
For both memory formats: time per call, whether EVERY map the gathers read arrives channels-last without an explicit
conversion, and the largest difference between the two formats' outputs.  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import torch.nn.functional as F
from pats_amd import dropin


class Basic(nn.Module):                      # torchvision-style BasicBlock, by shape
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.c1, self.b1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout)
        self.c2, self.b2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        self.down = None if stride == 1 and cin == cout else nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.b2(self.c2(torch.relu(self.b1(self.c1(x)))))
        return torch.relu(y + (x if self.down is None else self.down(x)))


class Trunk(nn.Module):                      # the three maps of forward2
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU())
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.l1 = nn.Sequential(Basic(64, 64), Basic(64, 64), Basic(64, 64))
        self.l2 = nn.Sequential(Basic(64, 128, 2), Basic(128, 128), Basic(128, 128), Basic(128, 128))

    def forward(self, x):
        x0 = self.stem(x)
        x1 = self.l1(self.pool(x0))
        return x0, x1, self.l2(x1)


class Pyramid(nn.Module):                    # FPN_8_2 by shape: dims 128 / 192 / 264 over maps of 64 / 64 / 128 channels
    def __init__(self):
        super().__init__()
        c3 = lambda a, b: nn.Conv2d(a, b, 3, 1, 1, bias=False)
        c1 = lambda a, b: nn.Conv2d(a, b, 1, 1, 0, bias=False)
        self.o3, self.o3b = c1(128, 264), nn.Sequential(c3(264, 264), nn.BatchNorm2d(264), nn.LeakyReLU(), c3(264, 264))
        self.o2, self.o2b = c1(64, 264), nn.Sequential(c3(264, 264), nn.BatchNorm2d(264), nn.LeakyReLU(), c3(264, 192))
        self.o1, self.o1b = c1(64, 192), nn.Sequential(c3(192, 192), nn.BatchNorm2d(192), nn.LeakyReLU(), c3(192, 128))

    def forward(self, x, before):
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=False)
        x3 = self.o3b(x) + self.o3(before[2])
        x2 = self.o2b(F.pad(self.o2(before[1]), (1, 1, 1, 1)) + F.pad(up(x3), (1, 1, 1, 1)))
        return self.o1b(F.pad(self.o1(before[0]), (2, 2, 2, 2)) + up(x2))


class Holder(nn.Module):                     # the attribute names dropin.prepare_backbones looks for
    def __init__(self):
        super().__init__()
        self.descriptor_extract, self.backbone = Trunk(), Pyramid()


def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B2 = int(os.environ.get("B2", "832"))
    torch.manual_seed(3)
    pic = torch.randn((B2, 3, 96, 96), device="cuda")                     # cat([left, right]) of permuted HWC crops: NCHW-contiguous
    mdesc = torch.randn((B2, 264, 12, 12), device="cuda")                 # mdesc[:, :, :-1].reshape(b, -1, 12, 12), third_layer.py:116
    m = Holder().cuda().eval()
    is_cl = lambda t: bool(t.is_contiguous(memory_format=torch.channels_last))
    res = {"crops": B2, "PYTORCH_MIOPEN_SUGGEST_NHWC": os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"),
           "torch": torch.__version__, "device": torch.cuda.get_device_name(0)}
    outs = {}
    with torch.no_grad():
        for tag in ("nchw", "channels_last"):
            if tag == "channels_last":
                dropin.prepare_backbones(m)                                # parameters -> channels_last; inputs stay as the reference builds them
            maps = m.descriptor_extract(pic)
            half = m.backbone(mdesc, maps)
            outs[tag] = [t.float().clone() for t in maps] + [half.clone()]
            res[tag] = {"forward2_ms": timeit(lambda: m.descriptor_extract(pic)),
                        "fpn_ms": timeit(lambda: m.backbone(mdesc, maps)),
                        "maps_arrive_channels_last": {"x0 [.,64,48,48]": is_cl(maps[0]), "x1 [.,64,24,24]": is_cl(maps[1]),
                                                      "x2 [.,128,12,12]": is_cl(maps[2]), "fpn [.,128,52,52]": is_cl(half)}}
    res["max_abs_diff_between_formats"] = max(float((a - b).abs().max()) for a, b in zip(outs["nchw"], outs["channels_last"]))
    res["max_abs_value"] = max(float(a.abs().max()) for a in outs["nchw"])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
