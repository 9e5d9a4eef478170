#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_bench.sh into the summaries committed under profiles/:
  profiles/<tag>_<grid>_trace_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, verbatim)
  profiles/<tag>_<grid>_pmc.json                 per kernel: mean FETCH_SIZE / WRITE_SIZE per launch and the corrected
                                                 HBM bytes; vector instructions per launch (SQ_INSTS_VALU); the clock
                                                 the chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration); wait shares
gfx950 corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE reports exactly half of a wide
coalesced streaming read (calibrated in round 1 on k_add_source, whose read volume is known exactly), so reads are
doubled; WRITE_SIZE is exact.

usage: tools/summarize_profiles.py r02 gpurun_out/prof_4096 [suffix]
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import library_source_hash  # noqa: E402  (the hash bench.py compares a summary's against)


def short(name):
    return name.replace("void ", "").split("(")[0].replace("fluid::", "")


def main():
    tag, root = sys.argv[1], sys.argv[2].rstrip("/")
    grid = re.search(r"prof_(\d+)", os.path.basename(root)).group(1)
    suffix = ("_" + sys.argv[3]) if len(sys.argv) > 3 else ""
    os.makedirs("profiles", exist_ok=True)
    for f in glob.glob(os.path.join(root, "trace", "*_kernel_stats.csv")):
        dst = "profiles/%s_%s%s_trace_kernel_stats.csv" % (tag, grid, suffix)
        shutil.copy(f, dst)
        print("kernel stats ->", dst)
    if os.path.exists(os.path.join(root, "bench_line.json")):
        shutil.copy(os.path.join(root, "bench_line.json"), "profiles/%s_%s%s_bench.json" % (tag, grid, suffix))
    # bench.py first lets the library's strip-height tuner finish in a throw-away context (trial launches with
    # deliberately varied strip heights): only the later half of each kernel's dispatches in a pass -- the warm-up and
    # the timed steps -- goes into the means
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, "*", "*_counter_collection.csv")):
        rows = list(csv.DictReader(open(f)))
        ids = collections.defaultdict(set)
        for r in rows:
            ids[short(r["Kernel_Name"])].add(int(r["Dispatch_Id"]))
        late = {k: set(sorted(v)[len(v) // 2:]) for k, v in ids.items()}
        for r in rows:
            k = short(r["Kernel_Name"])
            if int(r["Dispatch_Id"]) not in late[k]:
                continue
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur[k][r["Dispatch_Id"]] = (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    mean = lambda v: sum(v) / len(v)
    for k, c in sorted(pmc.items()):
        if not k.startswith("k_"):
            continue
        fetch = mean(c["FETCH_SIZE"]) * 1024 if c.get("FETCH_SIZE") else None
        write = mean(c["WRITE_SIZE"]) * 1024 if c.get("WRITE_SIZE") else None
        e = {"launches_sampled": len(c.get("FETCH_SIZE") or c.get("WRITE_SIZE") or c.get("SQ_INSTS_VALU") or []),
             "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
             "read_bytes_corrected_x2": None if fetch is None else 2 * fetch,
             "hbm_bytes_per_launch": None if fetch is None or write is None else 2 * fetch + write}
        if c.get("SQ_INSTS_VALU"):
            e["valu_insts_per_launch"] = mean(c["SQ_INSTS_VALU"])
            wc = sum(c["SQ_WAVE_CYCLES"]) or 1.0
            e["wait_any_share"] = sum(c["SQ_WAIT_ANY"]) / wc
            e["wait_inst_share"] = sum(c["SQ_WAIT_INST_ANY"]) / wc
            e["active_valu_share"] = sum(c["SQ_ACTIVE_INST_VALU"]) / wc
        if dur.get(k):
            g = sum(v[0] for v in dur[k].values())
            t = sum(v[1] for v in dur[k].values())
            e["clock_GHz_under_pmc"] = g / 8 / t
            e["mean_us_under_pmc"] = t / len(dur[k]) / 1e3
        out[k] = e
    if out:
        # what ties these counters to the kernels they were taken from: bench.py drops them (stale_profile) when the hash of
        # the library sources it runs with differs.  The profile must be summarised from the same tree it was taken on.
        stamp = os.path.join(root, "source_sha256")          # written on the GPU box by tools/profile_bench.sh
        out["_source_sha256"] = open(stamp).read().strip() if os.path.exists(stamp) else library_source_hash()
        path = "profiles/%s_%s%s_pmc.json" % (tag, grid, suffix)
        json.dump(out, open(path, "w"), indent=1)
        print("pmc ->", path)
        for k, v in out.items():
            if not isinstance(v, dict):
                continue
            print("  %-30s read(x2) %8.1f MB  write %8.1f MB  valu %s  clock %s" % (
                k, (v["read_bytes_corrected_x2"] or 0) / 1e6, (v["WRITE_SIZE_bytes"] or 0) / 1e6,
                "%.3g" % v["valu_insts_per_launch"] if v.get("valu_insts_per_launch") else "-",
                "%.2f GHz" % v["clock_GHz_under_pmc"] if v.get("clock_GHz_under_pmc") else "-"))


if __name__ == "__main__":
    main()
