#!/usr/bin/env python3
"""Static check of a HIP source's gfx950 ISA: per kernel, VGPRs / scratch bytes, and for every loop
that holds many FMAs (a hot loop) the number of scratch (spill) accesses and VALU instructions
inside it.  usage: check_hot_loops.py pats_amd/csrc/third_fused.hip [min_fma=20] [extra hipcc flags]
A spill reload inside a Sinkhorn sweep loop costs ~10 % of the kernel (toolchain notes)."""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
min_fma = int(sys.argv[2]) if len(sys.argv) > 2 else 20
extra = sys.argv[3:]                    # extra hipcc flags, e.g. -fno-slp-vectorize
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math",
                       "-ffp-contract=off"] + extra + ["-x", "hip", "-S", "--cuda-device-only", src, "-o", out],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
os.unlink(out)
kern, bad = None, 0
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
i = 0
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m: kern = m.group(1); continue
    m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and kern and labels.get(m.group(1), i + 1) < i:        # backward branch: a loop
        body = lines[labels[m.group(1)]:i + 1]
        fma = sum(1 for b in body if re.match(r"\s+v_(pk_)?(fma|fmac|mfma)", b))
        if fma < min_fma: continue
        scr = sum(1 for b in body if re.match(r"\s+scratch_", b))
        valu = sum(1 for b in body if re.match(r"\s+v_", b))
        inner = sum(1 for b in body[:-1] if re.match(r"\s+s_cbranch", b))
        print("%-60s loop@%d len=%d fma=%d valu=%d scratch=%d%s" % (kern[:60], labels[m.group(1)], len(body), fma, valu, scr,
                                                                  " (has inner branches)" if inner else ""))
        bad += scr > 0 and not inner
    m = re.match(r"\s+\.(vgpr_count|private_segment_fixed_size|vgpr_spill_count):\s+(\d+)", l)
    if m: print("    .%s %s" % (m.group(1), m.group(2)))
sys.exit(1 if bad else 0)
