#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/prof/ into the summaries committed
under profiles/:
  profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, verbatim)
  profiles/<tag>_pmc.json           (per-kernel mean FETCH_SIZE / WRITE_SIZE per
                                     launch and the corrected HBM bytes)
gfx950 corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB;
FETCH_SIZE reports exactly half of a wide coalesced streaming read -- checked
here on k_add_source, whose read volume is known exactly (2 fields x rows x
pitch x 4 B) -- so reads are doubled; WRITE_SIZE is exact.

usage: tools/summarize_profiles.py r01 [gpurun_out/prof]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    name = name.replace("void ", "")
    name = name.split("(")[0].replace("fluid::", "")
    # k_jacobi_tb<8, 1, float>: the pressure solves (one field per launch); <8, 2, float>: the
    # batched u/v/density diffusion (three fields per launch) -- kept apart
    return name


def main():
    tag = sys.argv[1]
    root = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
    os.makedirs("profiles", exist_ok=True)
    for f in glob.glob(os.path.join(root, "trace*", "*", "*_kernel_stats.csv")):
        sub = os.path.basename(os.path.dirname(os.path.dirname(f)))
        shutil.copy(f, "profiles/%s_%s_kernel_stats.csv" % (tag, sub))
        print("kernel stats ->", "profiles/%s_%s_kernel_stats.csv" % (tag, sub))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in sorted(pmc.items()):
        if not k.startswith("k_"):
            continue
        fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 if c.get("FETCH_SIZE") else None
        write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024 if c.get("WRITE_SIZE") else None
        out[k] = {"launches_sampled": len(c.get("FETCH_SIZE", c.get("WRITE_SIZE", []))),
                  "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
                  "read_bytes_corrected_x2": None if fetch is None else 2 * fetch,
                  "hbm_bytes_per_launch": None if fetch is None or write is None else 2 * fetch + write}
    if out:
        path = "profiles/%s_pmc.json" % tag
        json.dump(out, open(path, "w"), indent=1)
        print("pmc ->", path)
        for k, v in out.items():
            print("  %-28s read(x2) %8.1f MB  write %8.1f MB" % (
                k, (v["read_bytes_corrected_x2"] or 0) / 1e6, (v["WRITE_SIZE_bytes"] or 0) / 1e6))


if __name__ == "__main__":
    main()
