"""Per-kernel means of the counters collected by scripts/pmc_kernels.sh (all sets merged by kernel name)."""
import csv, glob, os, sys
from collections import defaultdict

base = sys.argv[1]
per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(dict)
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"], row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    for (k, _d), c in acc.items():
        for name, v in c.items():
            per[k][name].append(v)


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


for k, c in sorted(per.items()):
    n = max(len(v) for v in c.values())
    if n < 3:
        continue
    m = {name: sum(v) / len(v) for name, v in c.items()}
    print(f"== {short(k)}  ({n} dispatches)")
    for name in sorted(m):
        print(f"   {name:28s} {m[name]:16.1f}")
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        simd_cycles = m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        print(f"   mfma_busy_frac               {m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:16.3f}")
    if "SQ_WAVE_CYCLES" in m:
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if nm in m:
                print(f"   {nm + ' / WAVE_CYCLES':28s} {m[nm] / m['SQ_WAVE_CYCLES']:16.3f}")
    if "SQ_LDS_IDX_ACTIVE" in m and "SQ_LDS_BANK_CONFLICT" in m and m["SQ_LDS_IDX_ACTIVE"] > 0:
        print(f"   lds_conflict_frac            {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:16.3f}")
    if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
        print(f"   hbm_MB (2*FETCH+WRITE)       {(2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024 / 1e6:16.1f}")
