#!/usr/bin/env python3
"""Per-launch HBM traffic of the halo-tile conv from rocprofv3 --pmc csv output (FETCH_SIZE and WRITE_SIZE
passes).  FETCH_SIZE is doubled: on gfx950 it tallies the 128-byte requests of wide coalesced reads at 64 bytes.
Counter units are taken as reported by rocprofv3 for the derived metrics (KiB)."""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_amd.buildid import library_build_id
root = sys.argv[1]
vals = collections.defaultdict(list)
for f in glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3x3_halo_kernel<" in k:
            targs = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            if len(targs) >= 6 and targs[5] == "true":      # the NCHW-output form of the last layer
                continue
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"kernel": "conv3x3_halo_kernel (all launches but the last layer)", "command": "bench.py --num-of-step 2",
       "build_id": library_build_id()}      # the library the counters were taken on
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = vals.get(c, [])
    out[c + "_mean_raw"] = sum(v) / len(v) if v else None
    out[c + "_launches"] = len(v)
if vals.get("FETCH_SIZE") and vals.get("WRITE_SIZE"):
    kib = 1024.0
    out["fetch_bytes_per_launch"] = 2.0 * out["FETCH_SIZE_mean_raw"] * kib
    out["write_bytes_per_launch"] = out["WRITE_SIZE_mean_raw"] * kib
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
print(json.dumps(out, indent=1))
