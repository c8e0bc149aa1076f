#!/usr/bin/env python3
"""Per-launch durations of the 1-NN kernels from a rocprofv3 --kernel-trace CSV, in launch order (tile search / list walk / plain walk).
usage: trace_split.py <kernel_trace.csv> [last_n=60]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "nn1_stile_kernel" in n or "nn1_grid_kernel" in n:
        kind = "tile" if "tile" in n else ("list" if ", true>" in n.split("(")[0] else "walk")
        seq.append((int(r["Start_Timestamp"]), kind, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
seq.sort()
print(" ".join(f"{k}:{d:.2f}" for _, k, d in seq[-last:]))
