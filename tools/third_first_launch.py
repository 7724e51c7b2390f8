#!/usr/bin/env python3
"""What is special about the FIRST launch of the third-level kernel in a process (profiles/r03_third_determinism.md: 0-6 of
414 720 problems differ from every later launch by <= 1e-4 px; every later launch is bit-identical)?  One fresh process per
call; the scenario comes from the environment:

  (none)            launches 0..L-1 back to back; reports |launch i != launch 1| per launch and WHERE the differing problems
                    sit in the launch (problem index / P = the fraction of the launch that had been dispatched)
  SYNC_FIRST=1      synchronise and sleep 0.2 s before launch 0 (launch 0 otherwise queues right behind the input generation)
  HOST_QUIET=1      synchronise right behind every launch: no host activity (allocations, further launches) while it runs
  IDLE=<s>          after launch 2: sleep <s> seconds, launch again, twice (does a long idle re-create the effect?)
  PREHEAT=<kind>:<s>  before launch 0, <s> seconds of: matmul (8192^3 fp32) | vec (elementwise fp32 fma chain, no matrix pipe) |
                    copy (HBM copies) | same (the third-level kernel itself on the same inputs) | same1 (one such launch) |
                    other (the kernel on COPIES of the inputs) | touch (<s> passes of a.sum() over every input) |
                    outbufs (tensors of the output shapes written and freed <s> times: the allocator hands launch 0 used blocks) |
                    iters1 (round 6: one full-size launch with ONE sweep: all code on all CUs) | small (the full solve on 256 problems)
  PATS_REVERSE_BLOCKS=1  (round 6, diagnostic library) workgroup b solves problem P - 1 - b: do the affected problems follow the
                    dispatch order or the data?
  MATMUL_CHECK=1    additionally: is hipBLASLt's own first heavy launch reproducible?  (x @ x) repeated, first result against later
  SMI=1             print sclk / power from sysfs right before and after launch 0
  POISON=<hex>[:k]  (round 5, diagnostic library) before launch 0 - and again before launch k (default 2) - overwrite every vector
                    register, accumulation register and LDS word of the chip with the 32-bit pattern (pats_diag_poison): 0 = start
                    from clean leftovers; 7fc00000 = a quiet NaN in whatever the kernel reads before writing it.  If launch 0's
                    divergence goes away with 0, or a LATER launch diverges behind a NaN poison, the kernel consumes leftovers.
  P=<n>             problems per launch (default 414720)
The kernel under study is the fp16-split instantiation (PATS_THIRD_VARIANT=1350), which since round 4 exists in the diagnostic
library only: build it with `python -m pats_amd.build --diag` and run with PATS_AMD_DIAG_LIB=1 PATS_THIRD_VARIANT=1350 (the
round-4 logs were taken while it was still the production default); PATS_THIRD_VARIANT=300 = the production build."""
import glob
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pats_amd import ops, synth  # noqa: E402

P = int(os.environ.get("P", "414720"))
L = int(os.environ.get("L", "4"))


def smi(tag):
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            cur = [ln.strip() for ln in open(f) if "*" in ln]
            out.append("sclk " + ",".join(cur))
        except OSError:
            pass
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_*"):
        if f.endswith(("average", "input")):
            try:
                out.append("%s %.0f W" % (os.path.basename(f), int(open(f).read()) / 1e6))
            except (OSError, ValueError):
                pass
    print("[smi %s] %s" % (tag, "; ".join(out)), flush=True)


def inputs():
    gen = torch.Generator(device="cuda")
    gen.manual_seed(synth.SEED + 300)
    shape = (P, 128, 65)
    base = torch.randn(shape, device="cuda", generator=gen)
    d0 = 3.0 * (base + 0.3 * torch.randn(shape, device="cuda", generator=gen))
    d1 = 3.0 * (base + 0.3 * torch.randn(shape, device="cuda", generator=gen))
    gone = torch.rand((P, 1, 65), device="cuda", generator=gen) < 0.12
    d0 = torch.where(gone, 3.12 * torch.randn(shape, device="cuda", generator=gen), d0)
    d0[:, :, -1] *= 0.5
    d1[:, :, -1] *= 0.5
    sc = torch.exp(torch.sigmoid(0.3 * torch.randn((P, 1, 64), device="cuda", generator=gen)) * synth.LN256 - synth.LN256 / 2)
    p_s = torch.randint(1, 23, (P, 2), device="cuda", generator=gen) * 4
    p_t = torch.randint(0, 25, (P, 2), device="cuda", generator=gen) * 4
    return d0.contiguous(), d1.contiguous(), sc, p_s, p_t


def preheat(kind, seconds, args):
    t0 = time.perf_counter()
    if kind == "matmul":
        x = torch.randn((8192, 8192), device="cuda")
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                x = (x @ x) * 1e-4
            torch.cuda.synchronize()
    elif kind == "vec":
        x = torch.randn((1 << 28,), device="cuda")
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                x = torch.addcmul(x, x, x, value=1e-9)
            torch.cuda.synchronize()
    elif kind == "copy":
        x = torch.randn((1 << 30,), device="cuda")
        y = torch.empty_like(x)
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                y.copy_(x)
            torch.cuda.synchronize()
    elif kind == "same":
        while time.perf_counter() - t0 < seconds:
            ops.third_level(*args, outdoor=True)
            torch.cuda.synchronize()
    elif kind == "same1":          # ONE full-size launch of the kernel on the same inputs
        ops.third_level(*args, outdoor=True)
        torch.cuda.synchronize()
    elif kind == "other":          # the kernel, full size, for <seconds> on COPIES of the inputs (other memory), freed afterwards
        cp = [a.clone() for a in args]
        while time.perf_counter() - t0 < seconds:
            ops.third_level(*cp, outdoor=True)
            torch.cuda.synchronize()
        del cp
    elif kind == "outbufs":        # tensors of exactly the OUTPUT shapes written and freed: launch 0's outputs land in used blocks
        for _ in range(max(1, int(seconds))):
            o = [torch.zeros((P, 16, 2), device="cuda"), torch.zeros((P, 16, 2), device="cuda"), torch.zeros((P * 16, 2), device="cuda"),
                 torch.zeros((P, 16), dtype=torch.uint8, device="cuda")]
            torch.cuda.synchronize()
            del o
    elif kind == "iters1":         # round 6: ONE full-size launch of the same code object with a single sweep - every workgroup
        ops.third_level(*args, outdoor=True, iters=1)     # fetches and executes every code path on every CU, 1 % of the work / heat
        torch.cuda.synchronize()
    elif kind == "small":          # round 6: the full solve on 256 problems only - the code is in L2, one or two CUs have executed it
        ops.third_level(*[a[:256].contiguous() for a in args], outdoor=True)
        torch.cuda.synchronize()
    elif kind == "touch":          # every byte of the inputs read once by another kernel (address translations, no compute)
        for _ in range(max(1, int(seconds))):
            for a in args:
                a.sum()
        torch.cuda.synchronize()
    else:
        raise SystemExit("unknown PREHEAT kind " + kind)


def differing(a, b):
    return ((a[1] != b[1]).any(-1).any(-1) | (a[2].reshape(P, 16, 2)[..., 0] != b[2].reshape(P, 16, 2)[..., 0]).any(-1))


def main():
    args = inputs()
    quiet = bool(os.environ.get("HOST_QUIET"))
    if os.environ.get("MATMUL_CHECK"):
        x = torch.randn((8192, 8192), device="cuda")
        outs = [x @ x for _ in range(24)]
        torch.cuda.synchronize()
        print("hipBLASLt 8192^3 fp32, results differing from the LAST of 24 back-to-back products (elements):",
              [int((o != outs[-1]).sum()) for o in outs[:-1]], flush=True)
        del outs, x
    if os.environ.get("SYNC_FIRST"):
        torch.cuda.synchronize()
        time.sleep(0.2)
    if os.environ.get("PREHEAT"):
        kind, sec = os.environ["PREHEAT"].split(":")
        preheat(kind, float(sec), args)
    if os.environ.get("SMI"):
        torch.cuda.synchronize()
        smi("before launch 0")
    poison = None
    if os.environ.get("POISON"):
        import ctypes
        from pats_amd import _lib
        spec = os.environ["POISON"].split(":")
        poison = (int(spec[0], 16), int(spec[1]) if len(spec) > 1 else 2)
        lib = ctypes.CDLL(_lib.LIB_PATH)
        lib.pats_diag_poison.restype = ctypes.c_int
        lib.pats_diag_poison.argtypes = [ctypes.c_uint, ctypes.c_void_p]
        do_poison = lambda: lib.pats_diag_poison(ctypes.c_uint(poison[0]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    runs = []
    for i in range(L):
        if poison is not None and i in (0, poison[1]):
            assert do_poison() == 0
        runs.append(ops.third_level(*args, outdoor=True))
        if quiet or (i == 0 and os.environ.get("SMI")):
            torch.cuda.synchronize()
        if i == 0 and os.environ.get("SMI"):
            smi("after launch 0")
    torch.cuda.synchronize()
    if os.environ.get("IDLE"):
        for _ in range(2):
            time.sleep(float(os.environ["IDLE"]))
            if os.environ.get("SMI"):
                smi("after idle")
            runs.append(ops.third_level(*args, outdoor=True))
            torch.cuda.synchronize()
    ref = runs[1]
    counts = []
    for i, r in enumerate(runs):
        b = differing(r, ref)
        n = int(b.sum())
        counts.append(n)
        if n:
            idx = torch.nonzero(b).flatten().cpu().tolist()
            d = float((r[1] - ref[1]).abs().max())
            print("launch %d: %d problems differ from launch 1, max |d mkpts1_f| = %.2e px; index / P = %s" %
                  (i, n, d, ["%.3f" % (k / P) for k in idx[:8]]), flush=True)
    print("RESULT variant=%s scenario=%s differing-from-launch-1 per launch = %s" %
          (os.environ.get("PATS_THIRD_VARIANT", "default"),
           ",".join("%s=%s" % (k, os.environ[k]) for k in ("SYNC_FIRST", "HOST_QUIET", "IDLE", "PREHEAT", "POISON", "P", "PATS_REVERSE_BLOCKS") if k in os.environ) or "plain",
           counts), flush=True)


if __name__ == "__main__":
    main()
