#!/usr/bin/env python3
"""Build check for the 16-byte store-data hazard of integrate.hip.

A buffer_store_dwordx4 reads its four data VGPRs for some cycles after issue.  hipcc's hazard recogniser exempts
buffer stores whose soffset is an SGPR, and once scheduled a v_cndmask into the first data register right behind
such a store: that value, not the voxel, reached memory (512^3 only, timing dependent).  buf_store() therefore keeps
the data registers live across an `s_nop 7`.  This script reads the gfx950 assembly of integrate.hip and fails unless
EVERY buffer_store_dwordx4 is followed, in its own basic block and within a few instructions, by that s_nop, with no
instruction in between that writes one of the store's data registers.  `make` runs it (yak_amd/csrc/Makefile: hazard-check), so a compiler
or source change that brings the hazard back stops the build.

  python tools/check_store_hazard.py yak_amd/csrc/integrate.gfx950.s
"""
import re
import sys

WINDOW = 16  # instructions after the store within which the s_nop must come (scalar work may sit in between)
STORE = re.compile(r"^\s*buffer_store_dwordx4\s+v\[(\d+):(\d+)\]")
NOP = re.compile(r"^\s*s_nop\s+(\d+)")
REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def instructions(text):
    """(line number, text) of every instruction line (labels, directives and comments dropped)."""
    out = []
    for n, line in enumerate(text.splitlines(), 1):
        s = line.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":") or s.startswith("//"):
            continue
        out.append((n, s))
    return out


def written_vgprs(instr):
    """VGPRs an instruction writes: its first operand if that is a vector register (VALU, loads, readlane targets...).
    Stores and scalar instructions write none."""
    parts = instr.split(None, 1)
    if len(parts) < 2:
        return set()
    op = parts[0]
    if "store" in op or op.startswith("s_") or op.startswith("ds_write") or op.startswith("global_atomic"):
        return set()
    first = parts[1].split(",")[0].strip()
    m = REG.match(first)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def check(text):
    ins = instructions(text)
    problems, stores = [], 0
    for i, (n, s) in enumerate(ins):
        m = STORE.match(s)
        if not m:
            continue
        stores += 1
        data = set(range(int(m.group(1)), int(m.group(2)) + 1))
        guarded = False
        for n2, s2 in ins[i + 1:i + 1 + WINDOW]:
            nop = NOP.match(s2)
            if nop and int(nop.group(1)) >= 7:
                guarded = True
                break
            if s2.startswith("s_cbranch") or s2.startswith("s_branch") or s2.startswith("s_endpgm"):
                problems.append("line %d: control leaves the block of the store at line %d before the s_nop" % (n2, n))
                break
            hit = written_vgprs(s2) & data
            if hit:
                problems.append("line %d: '%s' writes v%s, data of the store at line %d, before the s_nop" % (
                    n2, s2, sorted(hit), n))
                break
        if not guarded and not any(("line %d" % n) in p.split(":", 1)[1] for p in problems):
            problems.append("line %d: no 's_nop 7' within %d instructions of '%s'" % (n, WINDOW, s))
    return stores, problems


def main():
    if len(sys.argv) != 2:
        sys.exit(__doc__)
    stores, problems = check(open(sys.argv[1]).read())
    if stores == 0:
        problems.append("no buffer_store_dwordx4 found: wrong file, or the kernels no longer use the 16-byte path")
    for p in problems:
        print("store-data hazard:", p)
    print("%d buffer_store_dwordx4 checked, %d problem(s)" % (stores, len(problems)))
    sys.exit(1 if problems else 0)


if __name__ == "__main__":
    main()
