#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of one rocprofv3 --pmc pass (csv output) and the MFMA busy fraction.
usage: pmc_sq_summary.py <pass dir> "<command line that was profiled>" > summary.json"""
import collections
import csv
import glob
import json
import re
import sys

root, command = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
out = {}
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0.0))[:12]:
    e = {"dispatches": len(seen[k])}
    e.update({c: d[c] for c in sorted(d)})
    if d.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * d["SQ_BUSY_CU_CYCLES"])
    if d.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_per_active"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    out[k] = e
json.dump({"command": command,
           "note": "sums over all dispatches of a kernel; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)",
           "kernels": out}, sys.stdout, indent=1)
