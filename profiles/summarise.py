#!/usr/bin/env python3
"""Summarises one profiles/collect.sh run into small tracked files:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table (per-kernel calls, total/average duration)
   <tag>_pmc.json           mean counter value per launch and kernel
   <tag>_traffic.json       HBM bytes per k_fill_reads launch: 2 x FETCH_SIZE + WRITE_SIZE, both in KiB as rocprofv3 reports them
                            (the factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md "HBM"; WRITE_SIZE uncalibrated)"""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reseq_amd.provenance import kernel_source_hash
root, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(root + "/stats/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pmc_*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("rsq::") or k.startswith("rsq_spec_"):        # the library's kernels and the read kernels compiled for the profile (rsq_spec.h)
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
pmc = {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
pmc["_kernel_source_hash"] = kernel_source_hash()           # bench.py: counters_stale when the sources it runs from hash differently
json.dump(pmc, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
fill = sorted((k for k in pmc if "fill_reads" in k and not k.startswith("_")), key=lambda k: not k.startswith("rsq_spec_"))       # the profile's own kernel first
if fill and "FETCH_SIZE" in pmc[fill[0]] and "WRITE_SIZE" in pmc[fill[0]]:
    k = fill[0]
    fetch_kib, write_kib = pmc[k]["FETCH_SIZE"]["mean"], pmc[k]["WRITE_SIZE"]["mean"]
    json.dump({"kernel": k, "fetch_size_kib_per_launch": fetch_kib, "write_size_kib_per_launch": write_kib,
               "hbm_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024,
               "note": "mean over the k_fill_reads launches of bench.py --steps 1 --warmup 0 --no-host-delivery (the sizing pass and the step: the whole job per launch); FETCH_SIZE doubled per the guide's gfx950 correction"},
              open(f"profiles/{tag}_traffic.json", "w"), indent=1)
for f in (root + "/bench.json", root + "/bench_under_rocprof.json"):
    try:
        line = open(f).read().strip().split("\n")[-1]
        json.loads(line)
        open(f"profiles/{tag}_{f.split('/')[-1]}", "w").write(line + "\n")
    except Exception as e:
        print("no bench line in", f, e)
print(open(f"profiles/{tag}_kernel_stats.csv").read()[:3000])
