"""Build step between hipcc's device compile and the assembler: fix-ups on the gfx950 assembly (pats_amd/build.py).

RULE 1 - `s_waitcnt lgkmcnt(0)` in front of every `s_barrier`.   THE BUG THIS EXISTS FOR (round 3).
    hipcc (ROCm 7.2, clang 22, gfx950) omits the wait where it reasons that the LDS operations of all waves execute in
    one total order, e.g. at the top of a Sinkhorn sweep loop whose latch ends in `ds_write_b32` (this wave's part of the
    scaling vector b, which the OTHER waves of the workgroup read after the barrier):

        ds_write_b32 v118, v89 offset:768      ; end of sweep k
        s_branch .LBB1_27  ...  s_cbranch_scc1 .LBB1_39
    .LBB1_28:
        s_barrier                              ; <- every other barrier of the kernel had `s_waitcnt lgkmcnt(0)` first
        ds_read_b128 v[2:5], v101 offset:768   ; sweep k + 1 reads b

    On MI355X a wave does pass the barrier with its write still queued, and the waves the barrier releases read the
    previous sweep's value.  Measured with tools/fine_determinism4.py on sinkhorn_blk145_kernel and on the log-domain
    sinkhorn_rc_kernel: 1-9 of 8 192 fine-level problems per launch ended with perturbed duals (rank-one difference
    u_i + v_j of up to 2e-2 in the log-plan, other problems in every launch, growing with the sweep count, none below 10
    sweeps), the rate swinging between 0 and 50 per 32 launches with anything that moved the loop head by a few cycles
    (profiles/r03_determinism.md).  With the wait in place: 0 in 256 launches over four differently scheduled builds.
    The wait costs nothing when no LDS operation is pending.

RULE 2 - fences for the sources of transcendental-unit instructions.   OFF by default; kept as a tool.
    The first hypothesis for the same symptom (a quarter-rate v_rcp_f32 / v_exp_f32 ... reading its source late, after a
    following VALU instruction has overwritten it).  Keeping divisors alive and fencing all ~590 such sites did change
    the error rates - by moving the code around the unsynchronised barrier, as it turned out: a build with rule 1 alone
    is clean.  `fence_asm(text, trans=True)` / PATS_BUILD_TRANS_FENCE=1 still apply it, tools/check_code_objects.py
    lists the sites.

tools/check_code_objects.py checks the code objects inside the BUILT library (what ships) for rule 1;
tests/test_host_abi.py runs it."""
import re

TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy|_clamp)?_(f16|f32|f64|bf16)(_e32|_e64|_sdwa|_dpp)?$")
VREG = re.compile(r"(?<![A-Za-z0-9_.])v(?:\[(\d+):(\d+)\]|(\d+))(?![A-Za-z0-9_])")
READS_DST = re.compile(r"^v_(pk_)?(fmac|mac|dot\d\w*c)_")
TWO_DST = re.compile(r"^v_(permlane\d+_swap|swap)_")
ASYNC = re.compile(r"^(ds_read|ds_load|global_load|buffer_load|scratch_load|flat_load|ds_bpermute|ds_permute|ds_swizzle|ds_\w+_rtn)")
MEM = re.compile(r"^(ds_|global_|buffer_|scratch_|flat_)")
MODIFIER = re.compile(r"^(row_|quad_perm|bank_mask|row_mask|bound_ctrl|offset|op_sel|neg_|clamp|mul:|div:|wave_|fi:|dst_sel|src\d_sel|"
                      r"dst_unused|glc|slc|nt|sc\d|off$|cbsz|abid|blgp|gds)")
WINDOW = 48


def vregs(op):
    out = set()
    for m in VREG.finditer(op):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def split_ops(s):
    ops, depth, cur = [], 0, ""
    for ch in s:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


class Insn:
    __slots__ = ("mn", "writes", "reads", "valu", "asyncw", "text", "target")

    def __init__(self, mn, writes, reads, valu, asyncw, text, target=None):
        self.mn, self.writes, self.reads, self.valu, self.asyncw, self.text, self.target = mn, writes, reads, valu, asyncw, text, target


def parse(line, comment):
    """One line of assembly / disassembly -> Insn, or None for labels, directives, blanks."""
    line = line.split(comment)[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    mn = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    ops = [o for o in split_ops(rest) if not MODIFIER.match(o)]
    # trailing modifiers separated by blanks only ("v_add_f32_dpp v1, v2, v3 row_ror:8 row_mask:0xf"): cut them off the last operand
    ops = [re.split(r"\s+(?=(?:row_|quad_perm|bank_mask|row_mask|bound_ctrl|offset|op_sel|neg_|clamp|mul:|div:|dst_sel|src\d_sel|dst_unused|"
                    r"glc|slc|nt\b|sc\d|cbsz|abid|blgp|gds|wave_|fi:))", o)[0] for o in ops]
    if mn.startswith("v_"):
        ndst = 2 if TWO_DST.match(mn) else 1
        writes, reads = set(), set()
        for o in ops[:ndst]:
            writes |= vregs(o)
        for o in ops[ndst:]:
            reads |= vregs(o)
        if ndst == 2 or READS_DST.match(mn) or "_dpp" in mn or " row_" in line or "quad_perm" in line or "_sdwa" in mn \
                or mn.startswith("v_mfma") or mn.startswith("v_smfma"):
            reads = reads | writes          # in-place exchange / accumulate / DPP and SDWA keep parts of the old value
        return Insn(mn, writes, reads, True, False, line)
    if ASYNC.match(mn):
        w = vregs(ops[0]) if ops else set()
        r = set()
        for o in ops[1:]:
            r |= vregs(o)
        return Insn(mn, w, r, False, True, line)
    if MEM.match(mn):                       # stores, atomics without return, exports of LDS
        r = set()
        for o in ops:
            r |= vregs(o)
        return Insn(mn, set(), r, False, False, line)
    target = None
    if mn.startswith("s_cbranch") or mn == "s_branch":
        target = rest.strip()
    return Insn(mn, set(), set(), False, False, line, target)


def exposed(items, labels, i, window=WINDOW, follow=True):
    """Sites where a source register of the transcendental instruction items[i] is overwritten by an ordinary VALU instruction
    before its result is read.  items: list of Insn or ("label", name).  Returns [(index of W, distance)]."""
    t = items[i]
    dst, src = t.writes, t.reads - t.writes
    out = []
    if not src:
        return out
    stack, seen_at = [(i + 1, 0)], {}
    while stack:
        j, n = stack.pop()
        while j < len(items) and n < window:
            if seen_at.get(j, window + 1) <= n:
                break
            seen_at[j] = n
            x = items[j]
            if not isinstance(x, Insn):
                j += 1
                continue
            if x.mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
                break
            if x.reads & dst:
                break                                   # the interlock on D has waited for T
            if x.writes & dst and x.valu:
                break
            if x.writes & src:
                if TRANS.match(x.mn):
                    break                               # same in-order unit
                if x.valu:
                    out.append((j, n + 1))
                    break
            if x.target is not None and follow:
                tj = labels.get(x.target)
                if tj is not None:
                    if x.mn == "s_branch":
                        j = tj
                        continue
                    stack.append((tj, n))
            elif x.mn == "s_branch" and not follow:
                pass
            if x.valu or x.asyncw or not x.mn.startswith("s_"):
                n += 1
            j += 1
    return out


def load(lines, comment):
    items, labels, where = [], {}, []
    for k, ln in enumerate(lines):
        s = ln.split(comment)[0].strip()
        m = re.match(r"^([A-Za-z_.$][\w.$]*):$", s)
        if m:
            labels[m.group(1)] = len(items)
            items.append(("label", m.group(1)))
            where.append(k)
            continue
        p = parse(ln, comment)
        if p is not None:
            items.append(p)
            where.append(k)
    return items, labels, where


def barrier_waits(lines):
    """Rule 1: `s_waitcnt lgkmcnt(0)` in front of every s_barrier that does not already have one directly before it."""
    out, n = [], 0
    prev = ""
    for ln in lines:
        s = ln.split(";")[0].strip()
        if s == "s_barrier" and not (prev.startswith("s_waitcnt") and "lgkmcnt(0)" in prev):
            out.append("\ts_waitcnt lgkmcnt(0)")
            n += 1
        out.append(ln)
        if s.endswith(":"):
            prev = ""                                      # a label: another path joins here, the wait must follow it
        elif s and not s.startswith("."):
            prev = s
    return out, n


def fence_asm(text, trans=False, barriers=True):
    """Patched assembly text and the number of fences inserted."""
    lines = text.split("\n")
    nb = 0
    if barriers:
        lines, nb = barrier_waits(lines)
    if not trans:
        return "\n".join(lines), nb
    items, labels, where = load(lines, ";")
    need = {}                                           # line index of W -> ordered list of destination registers to read
    for i, x in enumerate(items):
        if isinstance(x, Insn) and TRANS.match(x.mn):
            for j, _ in exposed(items, labels, i):
                d = min(x.writes)
                lst = need.setdefault(where[j], [])
                if d not in lst:
                    lst.append(d)
    if not need:
        return "\n".join(lines), nb
    out = []
    for k, ln in enumerate(lines):
        if k in need:
            out.append("\ts_nop 0")
            for d in need[k]:
                out.append("\tv_mov_b32_e32 v%d, v%d" % (d, d))
            out.append("\ts_nop 1")
        out.append(ln)
    return "\n".join(out), nb + sum(len(v) for v in need.values())
