#!/usr/bin/env python3
"""Static check of the SHIPPED gfx950 code (the code objects bundled in libpats_amd.so), see pats_amd/asm_pass.py.

Rule 1 (enforced, exit status 1): every `s_barrier` has `s_waitcnt ... lgkmcnt(0)` as the instruction directly before it.
hipcc (ROCm 7.2) leaves the wait out in front of some barriers - e.g. at the top of a sweep loop whose latch ends in a
ds_write - and on MI355X the waves released by the barrier then read LDS the late wave has not written yet: the
run-to-run differences of the fine-level Sinkhorn solve measured in round 3 (profiles/r03_determinism.md).

Informational: v_permlane*_swap count, and the sites of the refuted transcendental-source hypothesis (--trans).
usage: check_code_objects.py [libpats_amd.so] [--trans]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pats_amd.asm_pass import TRANS, Insn, exposed, load  # noqa: E402  (the analysis the build's pass uses)


def disassemble(path):
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True,
                          check=True).stdout.splitlines()


def scan(path, want_trans):
    lines = disassemble(path)
    kernel, prev = None, ""
    barriers, bare, swaps = 0, [], 0
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln.strip())
        if m:
            if not m.group(1).startswith("L"):
                kernel = m.group(1)
            prev = ""                 # a branch target: whatever preceded it in the listing is not what executes before it
            continue
        s = ln.split("//")[0].strip()
        if not s:
            continue
        mn = s.split()[0]
        if mn == "s_barrier":
            barriers += 1
            if not (prev.startswith("s_waitcnt") and "lgkmcnt(0)" in prev):
                bare.append((kernel, prev))
        if mn.startswith("v_permlane") and "_swap" in mn:
            swaps += 1
        prev = s
    ntrans, sites = 0, []
    if want_trans:
        items, labels, where = load(lines, "//")
        for i, x in enumerate(items):
            if isinstance(x, Insn) and TRANS.match(x.mn):
                ntrans += 1
                sites += [(x.text, items[j].text, d) for j, d in exposed(items, labels, i, follow=False)]
    return barriers, bare, swaps, ntrans, sites


def check(lib, want_trans=False):
    tmp = tempfile.mkdtemp(prefix="codeobj")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, check=True, cwd=tmp)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        tot = {"objects": len(objs), "barriers": 0, "bare": [], "swaps": 0, "trans": 0, "trans_sites": []}
        for f in objs:
            b, bare, sw, nt, sites = scan(os.path.join(tmp, f), want_trans)
            tot["barriers"] += b
            tot["bare"] += bare
            tot["swaps"] += sw
            tot["trans"] += nt
            tot["trans_sites"] += sites
        return tot
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pats_amd", "libpats_amd.so")
    t = check(lib, "--trans" in sys.argv)
    if not t["objects"]:
        print("no gfx950 code objects found in", lib)
        return 2
    print("%s: %d code objects, %d s_barrier, %d without `s_waitcnt lgkmcnt(0)` directly in front; %d v_permlane*_swap"
          % (os.path.basename(lib), t["objects"], t["barriers"], len(t["bare"]), t["swaps"]))
    for k, prev in t["bare"]:
        print("  %-70s preceded by: %s" % ((k or "?")[:70], prev or "(a label)"))
    if "--trans" in sys.argv:
        print("transcendental instructions: %d; source overwritten by a VALU instruction before the result is read: %d sites "
              "(hypothesis refuted, see asm_pass.py)" % (t["trans"], len(t["trans_sites"])))
    return 1 if t["bare"] else 0


if __name__ == "__main__":
    sys.exit(main())
