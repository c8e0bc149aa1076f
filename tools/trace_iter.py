#!/usr/bin/env python3
"""The timeline of the last iterations of an ICP loop from a rocprofv3 --kernel-trace CSV: per kernel its duration and the gap to the previous kernel's end.
usage: trace_iter.py <kernel_trace.csv> [last_n=12]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prev = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("pcr::", "")
    out.append(f"{name[:44]:44s} dur {(e - s) / 1e3:7.2f} us   gap {((s - prev) / 1e3 if prev else 0):7.2f} us")
    prev = e
print("\n".join(out[-last:]))
