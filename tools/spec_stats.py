#!/usr/bin/env python3
"""The read kernel compiled for profile P0 (what hiprtc builds on the device, rsq_spec.h), here with hipcc and line tables -- seconds per try, no GPU:
registers, spills, and the static make-up of the per-base step loop (the loop whose backward branch belongs to fill_wave_reads' `for (t = 0; any(running); ++t)`).

    python tools/spec_stats.py [--kind reads|records] [--variants] [--keep DIR] [-D NAME=VALUE ...]

Static counts: a rare branch inside the loop counts like the common path (the phase changes of advance_parts, the calls of the double-precision route, which are
listed separately).  The dynamic figure is the VALU counter of profiles/collect.sh."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_lines as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", choices=["reads", "records"], default="reads")
    ap.add_argument("--variants", action="store_true")
    ap.add_argument("--keep", default=None, help="directory for p0.hip / p0.s (default: a temporary one)")
    ap.add_argument("-D", action="append", default=[], help="extra -D for the compilation (experiments)")
    ap.add_argument("--dump", action="store_true", help="print the step loop")
    a = ap.parse_args()
    from reseq_amd import api, synth
    d = a.keep or tempfile.mkdtemp(prefix="rsq_spec_")
    os.makedirs(d, exist_ok=True)
    ppath = os.path.join(d, "p0.rsqp")
    if not os.path.exists(ppath):
        synth.write_profile(ppath, synth.make_profile(synth.P0, seed=103741084))
    prof = api.Profile(ppath)
    src = os.path.join(d, "p0.hip")
    prof.compile_read_kernel(kind=0 if a.kind == "reads" else 1, with_variants=a.variants, out_path=src)
    prof.close()
    kernel = "rsq_spec_fill_reads" if a.kind == "reads" else "rsq_spec_fill_records"
    asm = os.path.join(d, "p0.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-gline-tables-only", "-I", os.path.join(ROOT, "reseq_amd", "csrc"),
                    "--cuda-device-only", "-S", "-o", asm, src] + [f"-D{x}" for x in a.D], check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read()
    meta = {}
    block = text[text.index(f".amdhsa_kernel {kernel}"):]
    for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset"):
        m = re.search(rf"\.amdhsa_{key}\s+(\d+)", block)
        meta[key] = int(m.group(1)) if m else None
    for key in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "vgpr_count", "sgpr_count"):
        m = re.search(rf"\.{key}:\s+(\d+)", text[text.index(f".name:           {kernel}") - 3000:text.index(f".name:           {kernel}") + 3000])
        meta[key] = int(m.group(1)) if m else None
    insts, labels = K.parse(asm, kernel)
    # the step loop: the smallest loop (a backward branch and its target) that holds every packed multiply of the three draws
    loops = []
    for i, (ln, op, t) in enumerate(insts):
        m = re.search(r"(\.LBB[0-9_]+)", t)
        if op.startswith(("s_cbranch", "s_branch")) and m and labels.get(m.group(1), i + 1) <= i:
            s0 = labels[m.group(1)]
            loops.append((sum(1 for _, o, _ in insts[s0:i] if o.startswith("v_pk_mul")), -(i - s0), s0, i))
    # the step loops: since round 5 there are two (every lane in the template part / the general one) -- the smallest loops that hold the three draws' packed
    # multiplies (at least 60 of them), none containing another
    # (a loop shows as several backward branches to neighbouring labels: overlapping ranges are one loop, reported over their hull -- but a range that spans two
    # loops which each hold the draws is the chunk loop around them, not a step loop)
    cands = sorted((l for l in loops if l[0] >= 60), key=lambda l: l[3] - l[2])
    picked = []
    for l in cands:
        inside = [p for p in picked if p[2] >= l[2] and p[3] <= l[3]]
        if len({(p[2], p[3]) for p in inside}) >= 2 and not any(p[2] <= l[2] and p[3] >= l[3] for p in picked):
            lone = [p for p in inside if not any(q is not p and q[2] < p[3] and p[2] < q[3] for q in inside)]
            if len(lone) >= 2 or sum(1 for p in inside) >= 2 and max(p[3] for p in inside) - min(p[2] for p in inside) > 1.5 * max(p[3] - p[2] for p in inside):
                continue
        picked.append(l)
    hulls = []
    for l in sorted(picked, key=lambda l: l[2]):
        if hulls and l[2] <= hulls[-1][3]:
            hulls[-1] = (hulls[-1][0], hulls[-1][1], hulls[-1][2], max(hulls[-1][3], l[3]))
        else:
            hulls.append(l)
    picked = hulls
    print(f"{kernel}: vgprs {meta['vgpr_count']} sgprs {meta['sgpr_count']} spilled vgprs {meta['vgpr_spill_count']} sgprs {meta['sgpr_spill_count']} scratch {meta['private_segment_fixed_size']} B; "
          f"{len(insts)} instructions")
    for which, (_, _, lo, hi) in enumerate(picked):
        report(insts, labels, lo, hi, a.dump and which == 0, f"step loop {which + 1} of {len(picked)}")
    print(f"files in {d}")


def report(insts, labels, lo, hi, dump, title):
    body = insts[lo:hi + 1]
    ops = collections.Counter(op for _, op, _ in body)
    valu = sum(v for k, v in ops.items() if k.startswith("v_") and not k.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
    group = collections.Counter()
    for k, v in ops.items():
        for cls in ("v_mov_b32", "v_cndmask", "v_cmp", "v_pk_mul", "v_pk_add", "v_mad_u64", "v_lshl_add_u64", "v_readlane", "v_writelane", "ds_read", "ds_write", "global_load", "global_store",
                    "buffer_load", "buffer_store", "scratch_", "s_load", "s_nop", "s_waitcnt", "s_cbranch", "s_swappc"):
            if k.startswith(cls):
                group[cls] += v
    print(f"{title} [{lo}, {hi}]: {hi - lo + 1} instructions, VALU {valu}, SALU {sum(v for k, v in ops.items() if k.startswith('s_'))}")
    print("  " + ", ".join(f"{k} {v}" for k, v in group.most_common()))
    if dump:
        inv = {}
        for k, v in labels.items():
            inv.setdefault(v, []).append(k)
        for i in range(lo, hi + 1):
            ln, op, t = insts[i]
            print(f"{i} {' '.join(inv.get(i, [])):10s} {ln[0][4:7] if ln else ''}:{ln[1] if ln else ''}\t{t}")


if __name__ == "__main__":
    main()
