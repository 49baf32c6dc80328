#!/usr/bin/env python3
"""Per-step kernel sequence (duration, gap) from a rocprofv3 --kernel-trace CSV:  trace_summary.py x_kernel_trace.csv [n]"""
import csv
import re
import sys

t = list(csv.DictReader(open(sys.argv[1])))
t.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prev = None
tot = 0.0
for r in t[len(t) - n - 4: len(t) - 4]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void kurbm::", "").replace("kurbm::", "")
    print("%-58s wg %5d  dur %6.1f  gap %5.1f" % (name[:58], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                                                  (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0))
    prev = e
