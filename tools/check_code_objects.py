#!/usr/bin/env python3
"""Static check of the SHIPPED gfx950 code (the code objects bundled in libpats_amd.so).

Rule (enforced, exit status 1): every `s_barrier` has `s_waitcnt ... lgkmcnt(0)` in front of it in its basic block with no
LDS-memory / scalar-memory / flat operation in between (ds_bpermute / ds_permute / ds_swizzle - lane exchanges on the LDS
crossbar that touch no LDS memory - may sit behind the wait).  `wg_barrier()` of pats_amd/csrc/common.hpp puts the wait there in the
SOURCE (round 4; round 3 patched it into the compiler's assembly): `__syncthreads()` compiled by hipcc (ROCm 7.2) lacks it in
front of some barriers - e.g. at the top of a sweep loop whose latch ends in a ds_write - and on MI355X the waves released by
the barrier then read LDS the late wave has not written yet: the run-to-run differences of the fine-level Sinkhorn solve
measured in round 3 (profiles/r03_determinism.md).  A kernel that falls back to a bare `__syncthreads()` fails here.

Informational: the v_permlane*_swap count.
usage: check_code_objects.py [libpats_amd.so]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(path):
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True,
                          check=True).stdout.splitlines()


LGKM = re.compile(r"^(ds_|s_load|s_buffer_load|s_scratch_load|s_store|s_buffer_store|s_atomic|s_buffer_atomic|s_dcache|s_memtime|"
                  r"s_memrealtime|s_sendmsg|flat_|s_gl1_inv|s_atc_probe)")


CROSSBAR = re.compile(r"^(ds_bpermute_b32|ds_permute_b32|ds_swizzle_b32)$")


def scan_lines(lines):
    """A barrier is covered when, walking back from it inside its basic block, `s_waitcnt ... lgkmcnt(0)` is met before any
    instruction that puts an operation on the LGKM counter (LDS, scalar memory, flat, messages).  The scheduler may move VALU /
    SALU work between the wait of wg_barrier() and its s_barrier; that changes nothing for the hand-over.  A label or a branch
    in between (another path could join with LDS operations pending) counts as not covered."""
    kernel = None
    block = []                        # instructions of the current basic block, in order
    barriers, bare, swaps = 0, [], 0
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln.strip())
        if m:
            if not m.group(1).startswith("L"):
                kernel = m.group(1)
            block = []                # a branch target: whatever preceded it in the listing is not what executes before it
            continue
        s = ln.split("//")[0].strip()
        if not s:
            continue
        mn = s.split()[0]
        if mn == "s_barrier":
            barriers += 1
            covered, why = False, "(a label)"
            for q in reversed(block):
                qm = q.split()[0]
                if qm == "s_waitcnt" and "lgkmcnt(0)" in q:
                    covered = True
                    break
                if CROSSBAR.match(qm):
                    continue          # lane exchanges on the LDS crossbar: on the LGKM counter, but no LDS MEMORY is touched -
                                      # nothing another wave could read stale; the scheduler may place them behind the wait
                if LGKM.match(qm) or qm.startswith("s_cbranch") or qm in ("s_branch", "s_setpc_b64", "s_swappc_b64"):
                    why = q
                    break
            if not covered:
                bare.append((kernel, why))
        if mn.startswith("v_permlane") and "_swap" in mn:
            swaps += 1
        block.append(s)
    return barriers, bare, swaps


def scan(path):
    return scan_lines(disassemble(path))


def check(lib):
    tmp = tempfile.mkdtemp(prefix="codeobj")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, check=True, cwd=tmp)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        tot = {"objects": len(objs), "barriers": 0, "bare": [], "swaps": 0}
        for f in objs:
            b, bare, sw = scan(os.path.join(tmp, f))
            tot["barriers"] += b
            tot["bare"] += bare
            tot["swaps"] += sw
        return tot
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pats_amd", "libpats_amd.so")
    t = check(lib)
    if not t["objects"]:
        print("no gfx950 code objects found in", lib)
        return 2
    print("%s: %d code objects, %d s_barrier, %d without `s_waitcnt lgkmcnt(0)` in front; %d v_permlane*_swap"
          % (os.path.basename(lib), t["objects"], t["barriers"], len(t["bare"]), t["swaps"]))
    for k, prev in t["bare"]:
        print("  %-70s first LGKM operation / block boundary met walking back: %s" % ((k or "?")[:70], prev))
    return 1 if t["bare"] else 0


if __name__ == "__main__":
    sys.exit(main())
