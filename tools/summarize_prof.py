#!/usr/bin/env python3
"""Condense the rocprofv3 outputs merged into gpurun_out/ into small text files under profiles/ (committed).
usage: python tools/summarize_prof.py r01"""
import collections, csv, glob, hashlib, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the counters describe ONE build of the library: bench.py trusts them only while the sha of the loaded libpcr_hip.so is the same
LIB_SHA16 = hashlib.sha256(open(os.path.join(root, "hands-on-point-cloud-processing_amd", "libpcr_hip.so"), "rb").read()).hexdigest()[:16]
go = os.path.join(root, "gpurun_out")
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def newest(pattern):
    files = glob.glob(os.path.join(go, pattern))
    return max(files, key=os.path.getmtime) if files else None


lines = []
f = newest("prof_bench/*/*_kernel_stats.csv")
if f:
    lines.append("## rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (kernel_stats.csv)\n")
    lines.append("| kernel | calls | total ms | avg us | % |")
    lines.append("|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        lines.append(f"| {name} | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    lines.append("")
bj = os.path.join(go, "bench.json")
if os.path.exists(bj):
    try:
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        lines.append("## bench.py line of the same call (un-profiled run)\n")
        lines.append("```json\n" + json.dumps(b, indent=1) + "\n```\n")
    except Exception as e:  # noqa: BLE001
        lines.append(f"(bench.json unreadable: {e})\n")

pm = collections.OrderedDict()
durs = {}
for d in ("pmc_sq", "pmc_sq2", "pmc_mfma", "pmc_fetch", "pmc_write"):
    f = newest(f"{d}/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    rows = list(csv.DictReader(open(f)))
    # the dominant kernel of the profiled ICP loop: the warm-start kernel when it ran, else the cold one
    want = next((w for w in ("nn1_strack", "nn1_btrack", "nn1_etrack") if any(w in r["Kernel_Name"] for r in rows)), "nn1_ftrack")
    per_dispatch = collections.defaultdict(float)
    for r in rows:
        if want in r["Kernel_Name"]:
            per_dispatch[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
            kname = r["Kernel_Name"].split("(")[0]
    for (c, _), v in per_dispatch.items():
        agg[c].append(v)
    kt = newest(f"{d}/*/*_kernel_trace.csv")
    dd = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt)) if want in r["Kernel_Name"]]
    durs[d] = dd
    for k, v in agg.items():
        pm[k] = (sum(v) / len(v), len(v), d)
if pm:
    lines.append(f"## PMC counters of the dominant kernel `{kname}` (mean per launch; separate --pmc passes, tools/gpu_check.sh)\n")
    lines.append("| counter | mean / launch | launches | pass | launch ms in that pass |")
    lines.append("|---|---|---|---|---|")
    for k, (m, n, d) in pm.items():
        lines.append(f"| {k} | {m:.4g} | {n} | {d} | {sum(durs[d])/len(durs[d]):.3f} |")
    lines.append("")
    if "GRBM_GUI_ACTIVE" in pm:
        ms = sum(durs["pmc_sq"]) / len(durs["pmc_sq"])
        clk = pm["GRBM_GUI_ACTIVE"][0] / 8 / (ms * 1e-3) / 1e9
        lines.append(f"effective shader clock in the profiled pass = GRBM_GUI_ACTIVE / 8 XCDs / kernel time = **{clk:.2f} GHz**")
        if "SQ_INSTS_VALU" in pm:
            cyc = pm["GRBM_GUI_ACTIVE"][0] / 8 * 1024
            lines.append(f"VALU issue occupancy = SQ_INSTS_VALU x 2 cycles (wave64 on SIMD32) / (1024 SIMDs x cycles) = "
                         f"**{pm['SQ_INSTS_VALU'][0]*2/cyc:.3f}**")
            lines.append(f"VALU wave-instructions per (query,target) pair x 64 lanes = {pm['SQ_INSTS_VALU'][0]*64/1.44e10:.3f} lane-ops/pair "
                         "(algorithmic convention: 9)")
            lines.append("(the 2-cycle price undercounts this loop: the min / max / med3 class takes about twice the slot of an add / fma — "
                         "profiles/r01_ubench_valu_rate.txt)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in pm:
            cyc = pm["GRBM_GUI_ACTIVE"][0] / 8 * 1024
            lines.append(f"matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = **{pm['SQ_VALU_MFMA_BUSY_CYCLES'][0]/cyc:.3f}** "
                         f"({pm.get('SQ_INSTS_MFMA', (0,))[0]:.4g} MFMA instructions per launch; cycles in which matrix and vector instructions "
                         f"execute together, SQ_VALU_MFMA_COEXEC_CYCLES: {pm.get('SQ_VALU_MFMA_COEXEC_CYCLES', (0,))[0]/cyc:.3f} of the launch)")
    if "FETCH_SIZE" in pm:
        lines.append(f"HBM traffic per launch: FETCH_SIZE {pm['FETCH_SIZE'][0]:.0f} KiB x 2 (gfx950 correction for wide coalesced "
                     f"reads, MI355X_MICROARCH.md §HBM) = {pm['FETCH_SIZE'][0]*2*1024/1e6:.2f} MB; "
                     f"WRITE_SIZE {pm.get('WRITE_SIZE', (0,))[0]:.0f} KiB = {pm.get('WRITE_SIZE', (0,))[0]*1024/1e6:.2f} MB "
                     "(compulsory: 2.88 MB read, 0.96 MB key write + memset)")
if pm and "FETCH_SIZE" in pm:
    ms = sum(durs["pmc_sq"]) / len(durs["pmc_sq"]) if "pmc_sq" in durs else None
    js = {"lib_sha16": LIB_SHA16, "kernel": kname.replace("void ", ""), "source": f"profiles/{tag}_bench_rocprof_summary.md",
          "fetch_bytes_per_launch_corrected_x2": pm["FETCH_SIZE"][0] * 2 * 1024,
          "write_bytes_per_launch": pm.get("WRITE_SIZE", (0,))[0] * 1024,
          "clock_ghz_profiled": (pm["GRBM_GUI_ACTIVE"][0] / 8 / (ms * 1e-3) / 1e9) if "GRBM_GUI_ACTIVE" in pm else None,
          "valu_insts_per_launch": pm.get("SQ_INSTS_VALU", (None,))[0],
          "salu_insts_per_launch": pm.get("SQ_INSTS_SALU", (None,))[0], "mfma_insts_per_launch": pm.get("SQ_INSTS_MFMA", (None,))[0],
          "waves_per_launch": pm.get("SQ_WAVES", (None,))[0],
          # SQ_ACTIVE_INST_VALU counts quad-cycles of vector issue summed over the SIMDs; the launch offers 1024 SIMDs x its cycles
          "valu_busy_frac": (pm["SQ_ACTIVE_INST_VALU"][0] * 4 / (1024 * pm["GRBM_GUI_ACTIVE"][0] / 8)) if ("SQ_ACTIVE_INST_VALU" in pm and "GRBM_GUI_ACTIVE" in pm) else None,
          "mfma_busy_frac": (pm["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024 * pm["GRBM_GUI_ACTIVE"][0] / 8)) if ("SQ_VALU_MFMA_BUSY_CYCLES" in pm and "GRBM_GUI_ACTIVE" in pm) else None,
          "wave_life_us": (pm["SQ_WAVE_CYCLES"][0] * 4 / pm["SQ_WAVES"][0] / (pm["GRBM_GUI_ACTIVE"][0] / 8 / (ms * 1e-3)) * 1e6)
                          if all(k in pm for k in ("SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE")) and ms else None}
    json.dump(js, open(os.path.join(out, "latest_pmc.json"), "w"), indent=1)
open(os.path.join(out, f"{tag}_bench_rocprof_summary.md"), "w").write(
    f"# {tag}: rocprofv3 summary (generated by tools/summarize_prof.py from gpurun_out/)\n\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
