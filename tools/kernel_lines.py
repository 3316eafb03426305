#!/usr/bin/env python3
"""Which source lines a read kernel's instructions come from (static counts from an assembly listing with line information).

    python tools/kernel_lines.py spec.s rsq_spec_fill_reads [--loop] [--top 40] [--by function|line]

`spec.s`: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -I reseq_amd/csrc --cuda-device-only -S of the program
Profile.compile_read_kernel(out_path="x.hip") writes.  --loop: only the instructions inside the kernel's largest innermost-but-one loop nest that holds packed
multiplies (the per-base step loop).  Counts are static: a line inside a rare branch counts like one on the common path -- read them next to the source."""
import argparse
import collections
import re


def parse(path, kernel):
    files, cur, infn, out, labels = {}, None, False, [], {}
    for l in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if re.match(r"^%s:" % re.escape(kernel), l):
            infn = True
            continue
        if not infn:
            continue
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?").split("/")[-1], int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = len(out)
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        out.append((cur, t.split()[0], t))
    return out, labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("kernel")
    ap.add_argument("--loop", action="store_true")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--valu", action="store_true", help="count vector ALU instructions only")
    ap.add_argument("--range", default=None, help="first:last instruction of the kernel (as --loops of this tool lists them)")
    ap.add_argument("--loops", action="store_true", help="list the kernel's loops (backward branches): first, last, length, packed multiplies, 64-bit multiply-adds")
    a = ap.parse_args()
    insts, labels = parse(a.listing, a.kernel)
    lo, hi = 0, len(insts)
    if a.loops:
        for i, (_, op, text) in enumerate(insts):
            m = re.search(r"(\.LBB[0-9_]+)", text)
            if op.startswith("s_cbranch") and m and labels.get(m.group(1), i + 1) <= i:
                s = labels[m.group(1)]
                print(s, i, i - s, sum(1 for _, o, _ in insts[s:i] if o.startswith("v_pk_mul")), sum(1 for _, o, _ in insts[s:i] if o.startswith("v_mad_u64_u32")))
        return
    if a.range:
        lo, hi = (int(x) for x in a.range.split(":"))
    elif a.loop:
        loops = []
        for i, (_, op, text) in enumerate(insts):
            m = re.search(r"(\.LBB[0-9_]+)", text)
            if op.startswith("s_cbranch") and m and labels.get(m.group(1), i + 1) <= i:
                loops.append((labels[m.group(1)], i))
        with_pk = [(b - s, s, b) for s, b in loops if sum(1 for _, op, _ in insts[s:b] if op.startswith("v_pk_mul")) >= 20]
        _, lo, hi = min(with_pk)                                   # the smallest loop that holds the draws
        print(f"step loop: instructions {lo}..{hi} of {len(insts)}")
    sel = [(ln, op) for ln, op, _ in insts[lo:hi] if not a.valu or (op.startswith("v_") and not op.startswith("v_readlane") and not op.startswith("v_writelane"))]
    by = collections.Counter(ln for ln, _ in sel)
    print(f"{len(sel)} instructions")
    for (f, n), c in by.most_common(a.top):
        print(f"{c:6d}  {f}:{n}")


if __name__ == "__main__":
    main()
