#!/usr/bin/env python3
"""Loop census of a kernel in hipcc's gfx950 assembly (-S --cuda-device-only): for every backward branch, the
instruction mix of the code between its target and itself.  usage: isa_loops.py file.s <substring of the kernel symbol>"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = next(i for i, l in enumerate(txt) if l.startswith("_ZN") and key in l and l.rstrip().endswith(tuple([":"])) or (l.startswith("_ZN") and key in l and ": ;" in l))
    end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
    body = txt[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    print("%d lines" % len(body))
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            ins = [x.strip().split()[0] for x in body[a:i + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            c = collections.Counter(ins)
            pref = lambda *p: sum(v for k, v in c.items() if k.startswith(p))
            print("%-12s lines %6d..%6d instr %5d valu %5d salu %5d waitcnt %3d nop %3d branch %3d readlane %3d writelane %3d ds %3d buffer %3d global %3d scratch %3d trans %3d"
                  % (m.group(1), a, i, len(ins), pref("v_"), pref("s_") - c["s_waitcnt"] - c["s_nop"] - pref("s_cbranch", "s_branch"),
                     c["s_waitcnt"], c["s_nop"], pref("s_cbranch", "s_branch"), c["v_readlane_b32"], c["v_writelane_b32"], pref("ds_"),
                     pref("buffer_"), pref("global_"), pref("scratch_"), pref("v_rcp", "v_sqrt", "v_rsq")))


if __name__ == "__main__":
    main()
