#!/usr/bin/env python3
"""--pmc passes of tools/gpu_pmc_c4.sh -> profiles/<tag>_c4_pmc.md + profiles/latest_pmc_c4.json (bench.py: c4.roofline.traffic).
The radius kernels of the LAST filled call (count pass + emit launches) are summed.  usage: python tools/summarize_c4_pmc.py r04"""
import collections, csv, glob, hashlib, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(open(os.path.join(root, "hands-on-point-cloud-processing_amd", "libpcr_hip.so"), "rb").read()).hexdigest()[:16]
go = os.path.join(root, "gpurun_out")
res, lines = {}, []
for d, ctr in (("pmc_c4_fetch", "FETCH_SIZE"), ("pmc_c4_write", "WRITE_SIZE")):
    f = max(glob.glob(os.path.join(go, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and "radius_" in r["Kernel_Name"]]
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"].split("(")[0].replace("void pcr::", ""), 0.0])[1] += float(r["Counter_Value"])
    ids = sorted(per)
    # the last filled call = the trailing dispatches from the last radius count kernel on
    last_count = max(i for i in ids if "count" in per[i][0] or "grid_kernel<false" in per[i][0]) if any("count" in per[i][0] for i in ids) else ids[0]
    tail = [i for i in ids if i >= last_count]
    res[ctr] = sum(per[i][1] for i in tail)
    by = collections.defaultdict(float)
    for i in tail:
        by[per[i][0]] += per[i][1]
    lines.append(f"{ctr} (KiB) of the last filled call: " + ", ".join(f"{k} {v:.0f}" for k, v in by.items()))
fetch_b = res["FETCH_SIZE"] * 1024 * 2            # gfx950: FETCH_SIZE counts 64-byte units as 32 (MI355X_MICROARCH.md, HBM section)
write_b = res["WRITE_SIZE"] * 1024
out = {"lib_sha16": sha, "fetch_bytes_per_launch_corrected_x2": fetch_b, "write_bytes_per_launch": write_b,
       "source": f"profiles/{tag}_c4_pmc.md (tools/gpu_pmc_c4.sh: separate rocprofv3 --pmc passes of the radius r = 1 search of the 120 000-point scan)"}
json.dump(out, open(os.path.join(root, "profiles", "latest_pmc_c4.json"), "w"), indent=1)
md = [f"# {tag}: PMC passes of the configs[3] radius search (120 000 x 120 000, r = 1), library {sha}", ""] + lines + ["",
      f"HBM-side traffic of one filled call: FETCH_SIZE x 2 = {fetch_b / 1e6:.1f} MB + WRITE_SIZE {write_b / 1e6:.1f} MB = {(fetch_b + write_b) / 1e6:.1f} MB"]
open(os.path.join(root, "profiles", f"{tag}_c4_pmc.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
