"""Turn the rocprofv3 --pmc passes of tools/collect_pmc_util.sh into <tag>_pmc_util.json: per kernel the per-launch
averages of every collected counter and the utilisation figures derived from them.

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants" and "rocprofv3 PMC slots"): SQ_VALU_MFMA_BUSY_CYCLES counts
shader cycles summed over all SIMDs; GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles, so the cycles one launch
lasted are GRBM_GUI_ACTIVE / 8 and the matrix pipes' capacity in that time is (GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_LDS_BANK_CONFLICT = extra LDS cycles,
SQ_LDS_IDX_ACTIVE = all LDS-array cycles."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "util_g*", "**", f"{tag}_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
        if not m:
            continue
        acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[m.group(1)][row["Counter_Name"]] += 1
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "util_g*", "**", f"{tag}_kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
        if m:
            dur[m.group(1)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
kernels = {}
for k in acc:
    c = {name: acc[k][name] / cnt[k][name] for name in acc[k]}
    d = {"launches": max(cnt[k].values()), "per_launch": {n: round(v, 1) for n, v in sorted(c.items())}}
    if dur[k]:
        d["avg_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]), 2)
    gui = c.get("GRBM_GUI_ACTIVE")
    if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d["mfma_busy_pct"] = round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 256 * 4), 2)
    if c.get("SQ_WAVE_CYCLES"):
        for name, key in (("SQ_WAIT_ANY", "wave_parked_pct"), ("SQ_WAIT_INST_ANY", "issue_stall_pct"),
                          ("SQ_ACTIVE_INST_ANY", "issuing_pct")):
            if name in c:
                d[key] = round(100.0 * c[name] / c["SQ_WAVE_CYCLES"], 2)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_pct_of_lds_cycles"] = round(100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 2)
    if c.get("SQ_INSTS_VALU") and "SQ_INSTS_MFMA" in c:
        d["mfma_share_of_valu_insts_pct"] = round(100.0 * c["SQ_INSTS_MFMA"] / c["SQ_INSTS_VALU"], 2)
    kernels[k] = d
doc = {
    "command": f"rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu-baseline --workload {wl} --no-graph "
               f"--steps 20 --warmup 5   (one pass per counter group; tools/collect_pmc_util.sh)",
    "units": "per_launch = counter value averaged over the launches of that kernel; mfma_busy_pct = "
             "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs); wave_parked / issue_stall / issuing = "
             "SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; profiled passes run at a lower "
             "clock than unprofiled ones, so ratios are the figures to read, not the durations",
    "kernels": kernels,
}
json.dump(doc, open(os.path.join(out, f"{tag}_pmc_util.json"), "w"), indent=1)
print(json.dumps(kernels, indent=1))
