#!/usr/bin/env python3
"""Condense the --pmc passes of tools/gpu_pmc_grid.sh into profiles/<tag>_grid_pmc.md + profiles/latest_pmc_grid.json
(read by bench.py --nn grid / --workload c5 for roofline.traffic).  usage: python tools/summarize_grid_pmc.py r01 10000000"""
import collections, csv, glob, hashlib, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the counters describe ONE build of the library: bench.py trusts them only while the sha of the loaded libpcr_hip.so is the same
LIB_SHA16 = hashlib.sha256(open(os.path.join(root, "hands-on-point-cloud-processing_amd", "libpcr_hip.so"), "rb").read()).hexdigest()[:16]
go, out = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
vals, durs = {}, {}
SKIP = 2
for d in ("pmc_grid_fetch", "pmc_grid_write", "pmc_grid_sq"):
    files = glob.glob(os.path.join(go, d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    f = max(files, key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "nn1_grid_kernel" in r["Kernel_Name"]:
            agg[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    per = collections.defaultdict(list)
    for (c, _), v in agg.items():
        per[c].append(sum(v))
    for c, v in per.items():
        vals[c] = sum(v[SKIP:]) / max(len(v) - SKIP, 1) if len(v) > SKIP else v[-1]      # skip the index-building and the cold launch
    kt = max(glob.glob(os.path.join(go, d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt)) if "nn1_grid_kernel" in r["Kernel_Name"]]
    durs[d] = sum(dd[SKIP:]) / max(len(dd) - SKIP, 1) if len(dd) > SKIP else dd[-1]
lines = [f"# {tag}: PMC passes of pcr::nn1_grid_kernel inside an ICP loop at the converged pose, {n} x {n} (tools/gpu_pmc_grid.sh)\n",
         "| counter | mean / launch (warm launches) |", "|---|---|"]
for c, v in vals.items():
    lines.append(f"| {c} | {v:.5g} |")
lines.append("")
for d, ms in durs.items():
    lines.append(f"launch duration in pass {d}: {ms:.3f} ms")
if "FETCH_SIZE" in vals:
    fetch = vals["FETCH_SIZE"] * 2 * 1024
    write = vals.get("WRITE_SIZE", 0.0) * 1024
    ms = durs.get("pmc_grid_fetch")
    lines.append(f"\nHBM-side traffic per launch: FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB x 2 (gfx950 correction, MI355X_MICROARCH.md §HBM) = {fetch/1e6:.1f} MB"
                 f" + WRITE_SIZE {write/1e6:.1f} MB -> {(fetch+write)/ (ms*1e-3)/1e9:.0f} GB/s over the {ms:.2f} ms launch; compulsory: "
                 f"{n*(12+16+4+8)/1e6:.0f} MB (queries 12 B, records 16 B once, order 4 B, key 8 B)")
    json.dump({"lib_sha16": LIB_SHA16, "kernel": "pcr::nn1_grid_kernel", "n": n, "source": f"profiles/{tag}_grid_pmc.md", "fetch_bytes_per_launch_corrected_x2": fetch,
               "write_bytes_per_launch": write, "launch_ms_in_pass": ms}, open(os.path.join(out, "latest_pmc_grid.json"), "w"), indent=1)
open(os.path.join(out, f"{tag}_grid_pmc.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
