#!/usr/bin/env python3
"""Condense the --pmc passes of tools/gpu_pmc_grid.sh into profiles/<tag>_grid_pmc.md + profiles/latest_pmc_grid.json
(read by bench.py --nn grid / --workload c5 for roofline.traffic).  A SEARCH is one launch of the cell walk, or one launch of the tile
search plus the launch of the list walk that serves its deferred queries (csrc/grid_stile.hpp); counters and durations are summed per
search and averaged over the warm searches.  usage: python tools/summarize_grid_pmc.py r03 10000000 [suffix of the second set, e.g. _walk]"""
import collections, csv, glob, hashlib, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
other = sys.argv[3] if len(sys.argv) > 3 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the counters describe ONE build of the library: bench.py trusts them only while the sha of the loaded libpcr_hip.so is the same
LIB_SHA16 = hashlib.sha256(open(os.path.join(root, "hands-on-point-cloud-processing_amd", "libpcr_hip.so"), "rb").read()).hexdigest()[:16]
go, out = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
SKIP = 2          # the index-building one-shot search and the cold first search of the loop


def kind(name):
    if "nn1_tile_kernel" in name or "nn1_stile_kernel" in name:
        return "tile"
    if "nn1_grid_kernel" in name:
        return "list" if ", true>" in name.split("(")[0] else "walk"
    return None


def searches(items):
    """items: [(order, kind, value)] -> per-search sums [(kinds, total, {kind: value})]"""
    res = []
    for _, k, v in sorted(items):
        if k == "list" and res and res[-1][0] == ["tile"]:
            res[-1][0].append("list"); res[-1][1] += v; res[-1][2]["list"] = v
        else:
            res.append([[k], v, {k: v}])
    return res


def collect(suffix):
    vals, durs, split = {}, {}, {}
    for d in ("pmc_grid_fetch", "pmc_grid_write", "pmc_grid_sq"):
        files = glob.glob(os.path.join(go, d + suffix, "*", "*_counter_collection.csv"))
        if not files:
            continue
        f = max(files, key=os.path.getmtime)
        agg = collections.defaultdict(float)
        kinds = {}
        for r in csv.DictReader(open(f)):
            k = kind(r["Kernel_Name"])
            if k:
                agg[(r["Counter_Name"], int(r["Dispatch_Id"]))] += float(r["Counter_Value"])
                kinds[int(r["Dispatch_Id"])] = k
        per = collections.defaultdict(list)
        for (c, disp), v in agg.items():
            per[c].append((disp, kinds[disp], v))
        for c, items in per.items():
            s = searches(items)[SKIP:]
            vals[c] = sum(x[1] for x in s) / max(len(s), 1)
            for kk in ("tile", "list", "walk"):
                part = [x[2][kk] for x in s if kk in x[2]]
                if part:
                    split[(c, kk)] = sum(part) / len(s)
        kt = max(glob.glob(os.path.join(go, d + suffix, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
        items = [(int(r["Start_Timestamp"]), kind(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
                 for r in csv.DictReader(open(kt)) if kind(r["Kernel_Name"])]
        s = searches(items)[SKIP:]
        durs[d] = sum(x[1] for x in s) / max(len(s), 1)
        for kk in ("tile", "list", "walk"):
            part = [x[2][kk] for x in s if kk in x[2]]
            if part:
                split[("ms:" + d, kk)] = sum(part) / len(s)
    return vals, durs, split


vals, durs, split = collect("")
lines = [f"# {tag}: PMC passes of the exact grid search inside an ICP loop at the converged pose, {n} x {n} (tools/gpu_pmc_grid.sh), library {LIB_SHA16}\n",
         "One search = the tile kernel (pcr::nn1_stile_kernel; rounds 2-3: nn1_tile_kernel) + pcr::nn1_grid_kernel<16, false, 2, true> (the list walk of its deferred queries); figures are sums over the "
         "two launches, mean over the warm searches of the loop.\n",
         "| counter | per search | of which tile kernel | list walk |", "|---|---|---|---|"]
for c, v in vals.items():
    lines.append(f"| {c} | {v:.5g} | {split.get((c, 'tile'), 0.0):.5g} | {split.get((c, 'list'), split.get((c, 'walk'), 0.0)):.5g} |")
lines.append("")
for d, ms in durs.items():
    lines.append(f"search duration in pass {d}: {ms:.3f} ms (tile {split.get(('ms:' + d, 'tile'), 0.0):.3f} + list {split.get(('ms:' + d, 'list'), split.get(('ms:' + d, 'walk'), 0.0)):.3f})")
compulsory = n * (12 + 16 + 4 + 8)
if "FETCH_SIZE" in vals:
    fetch = vals["FETCH_SIZE"] * 2 * 1024
    write = vals.get("WRITE_SIZE", 0.0) * 1024
    ms = durs.get("pmc_grid_fetch")
    lines.append(f"\nHBM-side traffic per search: FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB x 2 (gfx950 correction, MI355X_MICROARCH.md §HBM) = {fetch/1e6:.1f} MB"
                 f" + WRITE_SIZE {write/1e6:.1f} MB = {(fetch+write)/1e6:.0f} MB -> {(fetch+write)/ (ms*1e-3)/1e9:.0f} GB/s over the {ms:.2f} ms search; compulsory: "
                 f"{compulsory/1e6:.0f} MB (queries 12 B, records 16 B once, winner position 4 B, key 8 B): x {(fetch+write)/compulsory:.2f}")
    json.dump({"lib_sha16": LIB_SHA16, "kernel": "pcr::nn1_stile_kernel + list walk", "n": n, "source": f"profiles/{tag}_grid_pmc.md", "fetch_bytes_per_launch_corrected_x2": fetch,
               "write_bytes_per_launch": write, "launch_ms_in_pass": ms, "valu_insts_per_launch": vals.get("SQ_INSTS_VALU")}, open(os.path.join(out, "latest_pmc_grid.json"), "w"), indent=1)
if other:
    v2, d2, _ = collect(other)
    lines.append(f"\n## the same passes with the cell walk alone (suffix {other}: tune grid_tile = 2), same library, same box\n")
    lines += ["| counter | cell walk alone | tile search + list walk | ratio |", "|---|---|---|---|"]
    for c in vals:
        if c in v2 and v2[c]:
            lines.append(f"| {c} | {v2[c]:.5g} | {vals[c]:.5g} | {vals[c] / v2[c]:.2f} |")
    for d in durs:
        if d in d2:
            lines.append(f"| ms per search, pass {d} | {d2[d]:.3f} | {durs[d]:.3f} | {durs[d] / d2[d]:.2f} |")
    if "FETCH_SIZE" in v2:
        f2 = v2["FETCH_SIZE"] * 2 * 1024 + v2.get("WRITE_SIZE", 0.0) * 1024
        lines.append(f"\ncell walk alone: FETCH x 2 + WRITE = {f2/1e6:.0f} MB per search = x {f2/compulsory:.2f} of the compulsory {compulsory/1e6:.0f} MB")
open(os.path.join(out, f"{tag}_grid_pmc.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
