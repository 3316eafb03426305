#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in libreseq_amd.so (read off the gfx950 code object's metadata).

    python tools/kernel_resources.py [pattern] [--lib path]

Unbundles the code object with clang-offload-bundler semantics (llvm-objdump --offloading), parses the AMDGPU metadata
note and prints one line per kernel whose demangled name contains `pattern`."""
import argparse
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"]


def kernels(lib, code_object=None):
    if code_object:                                   # a bare gfx code object (what rsq_profile_compile_read_kernel writes), not a library with a bundle
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", code_object], check=True, capture_output=True, text=True).stdout
        return parse_notes(notes)
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, "lib.so")
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, capture_output=True, cwd=d)
        cos = [f for f in os.listdir(d) if "amdgcn" in f]
        if not cos:
            raise SystemExit("no gfx code object found in " + lib)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(d, cos[0])], check=True, capture_output=True, text=True).stdout
    return parse_notes(notes)


def parse_notes(notes):
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- .") and key in ("agpr_count", "args"):      # a new kernel entry starts with its first key
            cur = {}
            out.append(cur)
        if cur is not None and (key in FIELDS or key == "name"):
            cur[key] = val
    return [k for k in out if "name" in k]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern", nargs="?", default="")
    ap.add_argument("--lib", default=os.path.join(ROOT, "reseq_amd", "libreseq_amd.so"))
    ap.add_argument("--code-object", default=None, help="a bare code object instead of the library (Profile.compile_read_kernel(out_path=...))")
    a = ap.parse_args()
    ks = kernels(a.lib, a.code_object)
    names = demangle([k["name"] for k in ks])
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>6}  kernel")
    for k, n in sorted(zip(ks, names), key=lambda x: x[1]):
        short = re.sub(r"\(.*", "", n).replace("void rsq::", "")
        if a.pattern and a.pattern not in short:
            continue
        print(f"{k.get('vgpr_count', '?'):>5} {k.get('agpr_count', '0'):>5} {k.get('sgpr_count', '?'):>5} {k.get('vgpr_spill_count', '0'):>6} {k.get('sgpr_spill_count', '0'):>6} "
              f"{k.get('private_segment_fixed_size', '0'):>7} {k.get('group_segment_fixed_size', '0'):>6}  {short}")


if __name__ == "__main__":
    main()
