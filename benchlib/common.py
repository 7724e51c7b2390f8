"""bench.py's shared constants (peaks from /opt/skills/guides/MI355X_MICROARCH.md, the workloads of BASELINE.json, the algorithmic
bytes per unit of the path's data-moving kernels - DESIGN.md section 6)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from pats_amd import synth  # noqa: E402,F401

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak


F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA


ITERS = 100


DTYPE = "f32 (contractions: fp32 operands split into fp16 hi + lo, three exact-product MFMA passes, fp32 accumulate)"


# name -> (grid h, grid w, if_local, outdoor, default pairs per step, label); BASELINE.json configs[1..3], SURVEY.md 8d
WORKLOADS = {"megadepth": (15, 20, True, True, 48, "configs[1]: MegaDepth 640x480 shapes, outdoor (if_local chunks of 2w, +ln2, label from the dustbin, merge_new)"),
             "scannet": (15, 20, False, False, 48, "configs[2]: ScanNet 640x480 shapes, indoor (one L2 chunk, cap 512; +ln3; fixed-cell label; merge_old)"),
             "yfcc": (24, 32, True, True, 16, "configs[3]: YFCC 768x1024 shapes (24x32 grid, 769x769 coarse problem), outdoor, merge_new")}


# algorithmic HBM bytes per unit of the four data-moving kernels of a step (kernel table); the same figures main() prices
# the headline's kernels with
THIRD_BYTES_PER_PROBLEM = 2 * 128 * 65 * 4 + 64 * 4 + 2 * 2 * 8 + 2 * 16 * 2 * 4 + 16 * 2 * 4 + 16


FINE_BYTES_PER_ROW = 2.0 * 264 * 145 * 4 + 145 * 145 * 4


FD_BYTES_PER_IMAGE = (2 * 64 * 144 * 4 + 128 * 144 + 8 + 264) * 4 + 264 * 145 * 4


TD_BYTES_PER_POINT = 2 * 128 * 64 * 4 + 128 * 4 + 2 * 128 * 65 * 4 + 2 * 2 * 4 + 8 + 2 * 2 * 8
