#!/usr/bin/env python3
"""One step of one rank's kernel timeline out of a rocprofv3 --kernel-trace of tools/slab_timing.py:
    rocprofv3 --kernel-trace -d D -o p --output-format csv -- python3 tools/slab_timing.py 8192 8
    python tools/slab_trace.py D/p_kernel_trace.csv [step-from-the-end, default 3] > profiles/rNN_slab8_step_timeline.txt
A step ends with the density advection (k_advect<...>); streams are numbered in order of appearance."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0].replace("fluid::", "")
ends = [i for i, r in enumerate(rows) if name(r).startswith("k_advect<")]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
queues = {}
print("#    start      dur     gap  q  kernel        (us from the step's first kernel; gap = no kernel of this process running before it)")
busy_until = t0
for r in step:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r["Queue_Id"], len(queues) + 1)
    gap = max(0, a - busy_until) / 1e3
    print("%10.1f %8.1f %7.1f  %d  %s" % ((a - t0) / 1e3, (b - a) / 1e3, gap, q, name(r)))
    busy_until = max(busy_until, b)
print("# step span %.1f us, %d kernels" % ((busy_until - t0) / 1e3, len(step)))
