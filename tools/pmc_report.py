#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs of tools/prof_solve.py (directories gpurun_out/pmc*_*): per fused-Jacobi kernel the
mean duration, the effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), VALU instructions per launch and the
share of the SIMDs' issue capacity they take (4 cycles per instruction of this mix, tools/ubench/valu_peak.hip)."""
import collections
import csv
import glob
import os
import sys

for d in sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc*_*")):
    if not os.path.isdir(d):
        continue
    rows = list(csv.DictReader(open(d + "/p_counter_collection.csv")))
    kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(d + "/p_kernel_trace.csv"))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if "k_jacobi_tb" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            t = kt.get(r["Dispatch_Id"])
            if t:
                agg[k]["dur_ns"] += int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    for k, v in agg.items():
        n = cnt[k]
        dur = v["dur_ns"] / n
        ghz = v["GRBM_GUI_ACTIVE"] / n / 8 / dur
        wc = v["SQ_WAVE_CYCLES"]
        valu = v["SQ_INSTS_VALU"] / n
        print("%-22s %-34s x%-3d %7.1f us  %.2f GHz  VALU %.3g/launch = %4.1f%% of issue  wait_any %4.1f%% wait_inst %4.1f%% active_valu %4.1f%%" % (
            os.path.basename(d), k[-34:], n, dur / 1e3, ghz, valu, 100 * valu * 4 / 1024 / (dur * ghz),
            100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_VALU"] / wc))
