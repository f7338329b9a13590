#!/usr/bin/env python3
"""ISA check of the DPP wait-state rule the inline-asm broadcast-FMAs rely on (gfx950): a VGPR
that a VALU instruction writes must not be read as a DPP operand by either of the next two
VALU-issue slots (two wait states; `s_nop N` supplies N+1).  hipcc pads its own DPP instructions
but cannot see a DPP modifier inside an `asm` statement, so lssvr_wave.hpp passes every DPP source
through an `s_nop 1` statement -- and this script proves on the assembly of the actual build that
nothing (a live-range split copy, a rescheduled instruction) ended up in between.  Second rule checked:
a VALU write of EXEC (v_cmpx*; hipcc masks with v_cmp + s_and_saveexec, so none is expected) needs five
wait states before a DPP instruction; an SALU write of EXEC needs none.

usage: check_dpp_hazard.py file.s [more.s ...]     (hipcc -S --cuda-device-only output)
Exit status 1 and a listing if a violation is found.  Run by __graft_entry__.build()."""
import re
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path):
    bad = 0
    ndpp = 0
    kernel = None
    hist = []          # (wait states elapsed since, written vgprs, text) of recent VALU writes
    cmpx = None        # wait states since the last VALU write of EXEC
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if not s or s[0] in ";./" or s.startswith("//"):
            if s.endswith(":") and s.startswith("_Z"):
                kernel, hist = s[:-1], []
            continue
        if s.endswith(":"):
            if s.startswith("_Z"):
                kernel, hist = s[:-1], []
            continue               # (block labels: the fall-through predecessor still counts)
        op = s.split()[0]
        if op == "s_nop":
            n = int(s.split()[1]) + 1
            hist = [(w + n, r, t) for (w, r, t) in hist]
            cmpx = None if cmpx is None else cmpx + n
            continue
        if not op.startswith("v_"):
            # non-VALU instructions take an issue slot as well
            hist = [(w + 1, r, t) for (w, r, t) in hist]
            cmpx = None if cmpx is None else cmpx + 1
            continue
        args = s[len(op):].split(",")
        if "row_newbcast" in s or "_dpp" in op:
            ndpp += 1
            if cmpx is not None and cmpx < 5:
                bad += 1
                print(f"{path}:{ln}: {kernel}: DPP instruction {cmpx} wait state(s) after a VALU write of EXEC\n    {s}")
            src0 = regs(args[1]) if len(args) > 1 else set()      # the DPP operand is src0
            for w, r, t in hist:
                if w < 2 and (r & src0):
                    bad += 1
                    print(f"{path}:{ln}: {kernel}: DPP read of v{sorted(r & src0)} {w} wait state(s) after\n"
                          f"    {t}\n    {s}")
        dst = regs(args[0]) if args else set()
        cmpx = 0 if op.startswith("v_cmpx") else (None if cmpx is None else cmpx + 1)
        hist = [(w + 1, r, t) for (w, r, t) in hist if w + 1 < 2]
        if dst and not op.startswith("v_cmp"):
            hist.append((0, dst, s))
    return ndpp, bad


if __name__ == "__main__":
    total_bad = 0
    for f in sys.argv[1:]:
        n, b = check(f)
        print(f"{f}: {n} DPP instructions checked, {b} hazard(s)")
        total_bad += b
    sys.exit(1 if total_bad else 0)
