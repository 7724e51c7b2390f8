#!/usr/bin/env python3
"""SQ-counter table of the bench's kernels from rocprofv3 --pmc passes (tools/pmc_sq.sh): per kernel (largest grid of its name)
mean over the launches of a pass.  GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 -> / 8 = cycles of the launch;
VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles);
LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles); conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
waves resident = SQ_WAVE_CYCLES / cycles (of 4096 wave slots at 4 per SIMD); waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.
usage: pmc_sq.py <dir of pass 1> <dir of pass 2> "title" > profiles/r04_pmc_sq.md"""
import collections, csv, glob, sys

def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]), r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, g, c), v in per.items():
            acc[(k, g)][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}

a, b = load(sys.argv[1]), load(sys.argv[2])
keys = [k for k in a if k in b and k[0].startswith("pats::")]
best = {}
for k in keys:                      # the largest launch of every kernel name
    if k[0] not in best or b[k].get("GRBM_GUI_ACTIVE", 0) > b[best[k[0]]].get("GRBM_GUI_ACTIVE", 0):
        best[k[0]] = k
rows = []
for name, k in best.items():
    cyc = b[k].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc < 2.0e4:
        continue
    A, B = a[k], b[k]
    rows.append((cyc, name, k[1], A, B))
rows.sort(reverse=True)
print("# %s\n" % sys.argv[3])
print(__doc__.split("usage:")[0].strip().replace("\n", " ") + "\n")
print("| kernel (grid) | cycles | VALU busy | MFMA busy | LDS busy | LDS conflict share | waves resident | waiting | VALU instr | LDS instr |")
print("|---|---|---|---|---|---|---|---|---|---|")
for cyc, name, grid, A, B in rows[:12]:
    valu = A.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc)
    mfma = B.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc)
    lds = B.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc)
    conf = B.get("SQ_LDS_BANK_CONFLICT", 0) / max(B.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    res = A.get("SQ_WAVE_CYCLES", 0) / cyc
    wait = A.get("SQ_WAIT_ANY", 0) / max(A.get("SQ_WAVE_CYCLES", 0), 1)
    print("| `%s` (%d) | %.3g | %.0f %% | %.0f %% | %.0f %% | %.2f | %.0f | %.0f %% | %.3g | %.3g |"
          % (name[:60], grid, cyc, 100 * valu, 100 * mfma, 100 * lds, conf, res, 100 * wait, A.get("SQ_INSTS_VALU", 0), A.get("SQ_INSTS_LDS", 0)))
