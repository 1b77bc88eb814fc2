"""profiles/r02_m3ae_mfma_utilisation.json from scripts/pmc_m3ae.sh: MFMA-busy fraction and effective clock per kernel of the
M3AE step (serialized), same formula as scripts/pmc_summarize.py."""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for math in ("split", "f32"):
    base = os.path.join(ROOT, "gpurun_out", "pmc_m3ae", math)
    cc = glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        continue
    per = defaultdict(lambda: defaultdict(dict))
    for row in csv.DictReader(open(cc[0])):
        d = per[row["Kernel_Name"]][row["Dispatch_Id"]]
        d[row["Counter_Name"]] = float(row["Counter_Value"])
        d["ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    rows = {}
    for k, v in per.items():
        busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in v.values())
        act = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in v.values())
        ns = sum(d.get("ns", 0.0) for d in v.values())
        if busy <= 0 or act <= 0:
            continue
        name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
        rows[name] = {"dispatches": len(v), "mfma_busy_frac": round(busy / (act / 8.0 * 1024.0), 4),
                      "avg_us": round(ns / len(v) / 1e3, 1), "clock_GHz": round(act / 8.0 / ns, 3)}
    out[math] = dict(sorted(rows.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["dispatches"]))
out["method"] = ("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 scripts/bench_m3ae.py "
                 "(OVERLAP=0 STEPS=2, B = 64, depth 12); mfma_busy_frac = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 XCDs x 256 CUs x 4 SIMDs)")
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_m3ae_mfma_utilisation.json"), "w"), indent=1)
for math in ("split", "f32"):
    for k, v in list(out.get(math, {}).items())[:8]:
        print(math, k[:60], v)
