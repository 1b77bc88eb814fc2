"""GPU idle time of the pipelined step from a rocprofv3 kernel trace: union of the kernel intervals against the wall span of the
timed steps.  usage: python scripts/trace_idle.py <dir with *_kernel_trace.csv> [window start fraction] [window end fraction]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]   # a steady-state window, by launch count
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
gaps.sort()
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy(union) {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %  sum of durations {tot / 1e6:.2f} ms "
      f"(avg concurrency {tot / busy:.2f})  gaps {len(gaps)}: total {sum(gaps) / 1e6:.2f} ms, median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, "
      f"max {gaps[-1] / 1e3 if gaps else 0:.1f} us")
