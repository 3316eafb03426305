#!/usr/bin/env python3
"""Samples of rocprofv3's PC sampling per (code object, offset), most frequent first; with the stochastic method also by stall reason / instruction type.
usage: pcsample_summary.py pcs_pc_sampling_*.csv"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
by_pc, by_field = collections.Counter(), collections.defaultdict(collections.Counter)
n = 0
for r in rows:
    n += 1
    key = (r.get("Code_Object_Id") or r.get("code_object_id"), r.get("Code_Object_Offset") or r.get("code_object_offset"), r.get("Instruction") or "", r.get("Instruction_Comment") or "")
    by_pc[key] += 1
    for k, v in r.items():
        if k and any(w in k.lower() for w in ("stall", "inst_type", "wave_issued", "reason", "arb")):
            by_field[k][v] += 1
print("samples", n, "columns", rows.fieldnames)
for k, c in by_field.items():
    print(k, c.most_common(12))
for (co, off, ins, com), c in by_pc.most_common(400):
    print(f"{c:8d} {100.0 * c / n:6.2f}%  co {co} +{off}  {ins}  {com[:60]}")
