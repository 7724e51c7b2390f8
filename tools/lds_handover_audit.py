#!/usr/bin/env python3
"""Best-effort audit for the round-4 verdict's item 4c: LDS hand-overs between lanes that are NOT separated by a barrier.
For every kernel file: the shared-memory objects (`__shared__` declarations, `extern __shared__` arrays and the pointers derived
from them by name), and every stretch of source between two barrier calls (wg_barrier / wg_barrier_global / wg_barrier_or /
wave_lds_sync / ring_barrier / __builtin_amdgcn_wave_barrier) in which such an object is first WRITTEN and later READ.  Those
stretches are printed for a human to classify: (a) the reader is the writing lane itself, (b) a hand-over inside one wave that
needs wave_lds_sync(), (c) a false positive of the text match.  The classification of the current tree is in DESIGN.md."""
import os, re, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pats_amd", "csrc")
BARRIER = re.compile(r"\b(wg_barrier(_global|_or)?|wave_lds_sync|ring_barrier|__builtin_amdgcn_wave_barrier|__syncthreads)\s*\(")
for fn in sorted(os.listdir(CSRC)):
    if not fn.endswith((".hip", ".hpp")):
        continue
    lines = open(os.path.join(CSRC, fn)).read().split("\n")
    names = set()
    for ln in lines:
        m = re.search(r"__shared__[^;=]*?\b(\w+)\s*(\[[^\]]*\])*\s*;", ln)
        if m:
            names.add(m.group(1))
    if not names:
        continue
    # pointers / references derived from them on one line: `T* p = lds + ...`, `auto& r = lds.x`
    for _ in range(2):
        for ln in lines:
            m = re.search(r"[\*&]\s*(\w+)\s*=\s*[^;]*\b(%s)\b" % "|".join(map(re.escape, names)), ln)
            if m and "const" not in ln.split("=")[0]:
                names.add(m.group(1))
    pat = "|".join(map(re.escape, sorted(names)))
    wr = re.compile(r"\b(%s)\b[^;=]*\]\s*(\.\w+\s*)?(=|\+=|\|=)[^=]|reinterpret_cast<[^>]*\*>\s*\(\s*(%s)\b[^;]*\)\s*=[^=]" % (pat, pat))
    rd = re.compile(r"\b(%s)\b" % pat)
    seg_start, written = 0, {}
    for i, ln in enumerate(lines):
        code = ln.split("//")[0]
        if BARRIER.search(code) or re.match(r"^\}", ln):
            written = {}
            continue
        w = wr.search(code)
        reads = [m.group(1) for m in rd.finditer(code)]
        for r_ in reads:
            if r_ in written and written[r_] != i and not (w and (w.group(1) or w.group(4)) == r_ and len(reads) == 1):
                print("%s:%d  reads `%s` written at line %d with no barrier between" % (fn, i + 1, r_, written[r_] + 1))
                del written[r_]
        if w:
            written[w.group(1) or w.group(4)] = i
