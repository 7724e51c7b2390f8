#!/usr/bin/env python3
"""Per-kernel HBM traffic of one bench step from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
WRITE_SIZE in separate passes, no trace domains beside them; FETCH_SIZE calibrated on a kernel of known byte count in
the same run, because other access widths than 16 B / lane are uncalibrated on gfx950).

usage: pmc_step.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <bench.py's JSON line of one pass> > profiles/r04_pmc_step_<layout>.json
Both passes ran `bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline` (tools/pmc_step.sh)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pats_amd", "csrc")


def csrc_sha16():
    """sha256[:16] of every kernel source: bench.py compares them with the tree it runs from, so a traffic figure that is older
    than the kernels it describes is visible in the bench line (roofline.traffic_age)."""
    out = {}
    for root, _, files in os.walk(CSRC):
        for fn in sorted(files):
            if fn.endswith((".hip", ".hpp", ".cpp")):
                out[os.path.relpath(os.path.join(root, fn), CSRC)] = hashlib.sha256(open(os.path.join(root, fn), "rb").read()).hexdigest()[:16]
    return out


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))] += float(r["Counter_Value"])
        for (_, k, g), v in per.items():
            acc[(k, g)].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def summarise(fetch_dir, write_dir, bench_line_path):
    fetch, nf = load(fetch_dir, "FETCH_SIZE")
    write, _ = load(write_dir, "WRITE_SIZE")
    # the fine level's cost build is the cost_mfma_kernel launch with the largest grid: one 256-thread workgroup per row
    grids = [k[1] for k in fetch if k[0].startswith("pats::cost_mfma_kernel")]
    bench = json.loads(open(bench_line_path).read().strip().splitlines()[-1])
    rows_cap = max(grids) // 256 if grids else int(bench["rows_cap"])
    rows_live = int(bench["rows_in_use_per_step"])      # the launches cover rows_cap, workgroups past the device-side count return
    # rocprofv3 reports both counters in KB
    KB = 1024.0
    # calibrator: the fine-level cost build reads exactly 2 x 264 x 145 fp32 per problem (8-byte lane loads, coalesced)
    # and writes 145 x 145 fp32 per problem; one 256-thread workgroup per problem
    cal_key = [k for k in fetch if k[0].startswith("pats::cost_mfma_kernel") and k[1] == rows_cap * 256]
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python bench.py --steps 3 --warmup 1 "
                      "--no-secondary --no-cpu-baseline (one run per counter, no trace domains)",
           "unit": "bytes per launch (mean over the launches of the run)"}
    # WRITE_SIZE needs no factor: tools/write_patterns.hip (profiles/r06_write_patterns.txt) writes 1.68 GB once with 16- / 8- / 4-byte
    # linear stores and the counter x 1024 equals the byte count to 5 digits.  Until round 5 the cost build's "known" write count
    # calibrated it to x0.93 - but that kernel's MFMA-layout stores really move 1.05-1.08x their algorithmic bytes (partial lines:
    # pattern d of the same probe), so every other kernel's writes were under-reported by 7 %.
    WRITE_FACTOR = 1.0
    f_fac = w_fac = None
    if cal_key:
        k = cal_key[0]
        known_r, known_w = 2.0 * 264 * 145 * 4 * rows_live, 145.0 * 145 * 4 * rows_live
        f_fac, w_fac = known_r / (fetch[k] * KB), known_w / (write[k] * KB)
        out["calibration"] = {"kernel": k[0], "grid": k[1], "known_read_bytes": known_r, "FETCH_SIZE_raw_bytes": fetch[k] * KB,
                              "fetch_factor": f_fac, "known_write_bytes": known_w, "WRITE_SIZE_raw_bytes": write[k] * KB,
                              "write_factor": WRITE_FACTOR, "cost_build_write_overhead": 1.0 / w_fac,
                              "how_writes": "WRITE_SIZE x 1024 = bytes for linear 16- / 8- / 4-byte stores (tools/write_patterns.hip, "
                                            "profiles/r06_write_patterns.txt): no factor; cost_build_write_overhead = what this kernel's "
                                            "MFMA-layout stores move over their algorithmic bytes",
                              "how": "pats::cost_mfma_kernel at the fine level reads 2 x 264 x 145 x 4 B and writes 145 x 145 x 4 B per "
                                     "problem, nothing else; FETCH_SIZE of that launch in the same run gives the read factor "
                                     "applied to every kernel below (the guide: x2 for 16-byte lane loads; backed independently by "
                                     "tools/fetch_patterns.hip)"}
    # FETCH_SIZE counts fabric requests at 64 B each whatever their size, so the factor depends on the access width of
    # the kernel: the cost build's loads come out at x2.00 (the guide's figure for 16-byte lane loads); the third-level
    # kernel streams its descriptors with 8-byte lane loads, calibrated in round 2 on cost65_kernel (the same loads,
    # known byte count, same run): x1.381 (profiles/r02_pmc_third.json)
    override = {"pats::third_fused3_kernel": 1.381}
    out["rows_cap"] = rows_cap
    out["rows_in_use"] = rows_live
    ks = {}
    for k in sorted(fetch, key=lambda k: -fetch[k]):
        if not k[0].startswith("pats::"):
            continue
        r, w = fetch[k] * KB, write.get(k, 0.0) * KB
        ff = next((v for n, v in override.items() if k[0].startswith(n)), f_fac or 1.0)
        ks["%s grid=%d" % k] = {"launches_seen": nf[k], "FETCH_SIZE_raw_bytes": r, "WRITE_SIZE_raw_bytes": w, "fetch_factor": ff,
                                "hbm_read_bytes": r * ff, "hbm_write_bytes": w * WRITE_FACTOR,
                                "hbm_bytes": r * ff + w * WRITE_FACTOR}
    out["kernels"] = ks
    out["csrc_sha16"] = csrc_sha16()
    return out


def main():
    json.dump(summarise(sys.argv[1], sys.argv[2], sys.argv[3]), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
