#!/usr/bin/env python3
"""The largest loops of a kernel in libreseq_amd.so and what they are made of (static instruction counts from the gfx950 disassembly).

    python tools/kernel_loops.py 'k_fill_readsILj10ELb0ELb0' [--lib path] [--loops 4]

A loop = a backward branch; loops are listed by length.  Useful to see whether scratch traffic (spills) or lane reads of spilled scalar
registers sit inside the per-base step loop of the read kernels."""
import argparse
import collections
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = ("scratch_load", "scratch_store", "v_readlane", "v_writelane", "s_load", "global_load", "global_store", "global_atomic", "ds_read", "ds_write", "v_pk_", "v_fma_f64",
           "v_mul_f64", "v_add_f64", "s_waitcnt", "s_cbranch", "v_", "s_")


def disassemble(lib, code_object=None):
    if code_object:
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", code_object], check=True, capture_output=True, text=True).stdout
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "lib.so")
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, capture_output=True, cwd=d)
        co = [f for f in os.listdir(d) if "amdgcn" in f][0]
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(d, co)], check=True, capture_output=True, text=True).stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern")
    ap.add_argument("--lib", default=os.path.join(ROOT, "reseq_amd", "libreseq_amd.so"))
    ap.add_argument("--loops", type=int, default=4)
    ap.add_argument("--code-object", default=None, help="a bare code object instead of the library (Profile.compile_read_kernel(out_path=...))")
    a = ap.parse_args()
    funcs, cur = collections.OrderedDict(), None
    for line in disassemble(a.lib, a.code_object).splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
        if m:
            cur = (m.group(2), int(m.group(1), 16))
            funcs[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur:
            t = re.search(r"\+0x([0-9a-f]+)>\s*$", line)
            funcs[cur].append((int(m.group(3), 16), m.group(1), int(t.group(1), 16) if t else None))
    for (name, base), ins in funcs.items():
        if a.pattern not in name:
            continue
        index = {addr: i for i, (addr, _, _) in enumerate(ins)}
        loops = []
        for i, (addr, op, target) in enumerate(ins):
            if (op.startswith("s_cbranch") or op == "s_branch") and target is not None and base + target <= addr and base + target in index:
                loops.append((index[base + target], i))
        loops.sort(key=lambda x: x[0] - x[1])
        print(f"{name[:90]}: {len(ins)} instructions")
        for lo, hi in loops[:a.loops]:
            c = collections.Counter()
            for _, op, _ in ins[lo:hi + 1]:
                for k in CLASSES:
                    if op.startswith(k):
                        c[k] += 1
                        if k not in ("v_", "s_"):
                            break
            print(f"   loop [{lo}, {hi}] {hi - lo + 1} instructions: " + ", ".join(f"{k} {v}" for k, v in c.most_common()))


if __name__ == "__main__":
    main()
