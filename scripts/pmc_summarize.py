"""Summarise the rocprofv3 --pmc passes of scripts/pmc_bench.sh into profiles/<R>_igemm_traffic.json (HBM bytes per conv call,
gfx950 correction: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, MI355X_MICROARCH.md HBM section) and
profiles/<R>_mfma_utilisation.json (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD-cycles of the dispatch)."""
import csv, glob, json, os, sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", f"pmc_{R}")
STEPS = 3          # 1 warm-up + 2 timed + (bench.py's serialized roofline pass: 1 + 2) = 6 steps in the process
CALLS_PER_STEP = 76        # conv forward + input-gradient calls of the 64..512-channel layers (the stem runs on its own kernels since round 3)


def counters(tag):
    """kernel name -> list of dicts counter -> value (one per dispatch)"""
    files = glob.glob(os.path.join(base, tag, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: defaultdict(dict))
    for f in files:
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


traffic, util = {}, {}
for math, kname in (("f32", ("igemm_kernel",)), ("split", ("igemm_split_kernel", "patch_split_kernel", "patch64p_kernel"))):
    f = counters(f"{math}_FETCH_SIZE")
    w = counters(f"{math}_WRITE_SIZE")
    if not f or not w:
        continue
    def mine(k):
        sk = short(k)
        # (igemm_kernel<..., true> is the scalar-gather stem variant of the fp32 path: not one of the 76 calls)
        return any(sk.startswith(n + "<") for n in kname) and not (sk.startswith("igemm_kernel<") and "true>" in k)
    fs = [d["FETCH_SIZE"] for k, v in f.items() if mine(k) for d in v]
    ws = [d["WRITE_SIZE"] for k, v in w.items() if mine(k) for d in v]
    n = len(fs)
    steps_in_process = 6
    per_launch = (2 * sum(fs) / n + sum(ws) / len(ws)) * 1024
    per_step = per_launch * n / steps_in_process
    traffic[math] = {"kernel": " + ".join(kname), "launches_profiled": n, "fetch_size_kb_per_launch": sum(fs) / n,
                     "write_size_kb_per_launch": sum(ws) / len(ws), "hbm_bytes_per_launch": per_launch,
                     "kernel_launches_per_step": n / steps_in_process, "conv_calls_per_step": CALLS_PER_STEP + (2 if math == "f32" else 0),
                     "hbm_bytes_per_step": per_step, "hbm_bytes_per_call": per_step / (CALLS_PER_STEP + (2 if math == "f32" else 0))}
    u = counters(f"{math}_SQ_VALU_MFMA_BUSY_CYCLES")
    rows = {}
    for k, v in u.items():
        busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in v)
        act = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in v)
        if busy <= 0 or act <= 0:
            continue
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs (4 per CU x 256 CUs); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        rows[short(k)] = {"dispatches": len(v), "mfma_busy_frac": busy / (act / 8.0 * 1024.0)}
    util[math] = dict(sorted(rows.items(), key=lambda kv: -kv[1]["mfma_busy_frac"]))
traffic["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                     "--no-alt --no-overlap --math {f32,split}`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per "
                     "128-B request, MI355X_MICROARCH.md HBM section), averaged over all launches of the kernel")
traffic["note"] = ("a conv forward / input-gradient CALL (what bench.py's roofline counts: 40 + 38 per step) is one kernel launch, except a "
                   "stride-2 input gradient, which is one launch per output parity class; hbm_bytes_per_call = bytes per step / 76; the stem "
                   "kernels are excluded")
util["method"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over the same command; "
                  "mfma_busy_frac = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 XCDs x 256 CUs x 4 SIMDs), the same formula as round 1")
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{R}_igemm_traffic.json"), "w"), indent=1)
json.dump(util, open(os.path.join(ROOT, "profiles", f"{R}_mfma_utilisation.json"), "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in list(v.items())[:6]}) for k, v in util.items()}, indent=1))
print({k: round(v["hbm_bytes_per_call"] / 1e6, 1) for k, v in traffic.items() if isinstance(v, dict)}, "MB per conv call")
