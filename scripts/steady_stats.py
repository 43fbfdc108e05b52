"""Steady-state per-kernel statistics from a rocprofv3 --kernel-trace CSV: the window from the end of the warm-up steps
(counted by a marker kernel that runs once per step) to the last timed step.  argv: trace dir tag under gpurun_out/, marker
kernel substring, warm-up steps, timed steps, output csv."""
import collections
import csv
import glob
import sys

tag, marker, skip, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
fs = glob.glob(f"gpurun_out/{tag}/*/*kernel_trace.csv")
if not fs:
    sys.exit(f"{tag}: no trace")
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
per_pass = len(marks) // (skip + steps) if marks else 0
if not per_pass:
    sys.exit(f"{tag}: marker not found")
lo, hi = marks[skip * per_pass - 1] + 1, marks[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["End_Timestamp"])
own = lambda k: k.startswith("k_") or "void k_" in k
for r in rows[lo:hi + 1]:
    k = r["Kernel_Name"].split("(")[0][:80]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
detail = sys.argv[6] if len(sys.argv) > 6 else None       # kernel launched several times per step: average by position in the step
if detail:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[lo:hi + 1] if r["Kernel_Name"].split("(")[0] == detail]
    n = len(d) // steps
    if n:
        print(detail, "by position in the step (us):", [round(sum(d[i::n]) / len(d[i::n]), 1) for i in range(n)])
tot = sum(v[1] for v in agg.values())
mine = sum(v[1] for k, v in agg.items() if own(k))
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "calls_per_step", "total_us", "avg_us", "pct_of_kernel_time"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, v[0], round(v[0] / steps, 2), round(v[1], 1), round(v[1] / v[0], 2), round(100 * v[1] / tot, 2)])
    w.writerow(["# steady-state window", f"{steps} steps", f"launches/step {sum(v[0] for v in agg.values()) / steps:.1f}",
                f"kernel time/step {tot / steps:.1f} us", f"wall/step {(t1 - t0) / 1e3 / steps:.1f} us",
                f"own kernels {100 * mine / tot:.1f} % of kernel time"])
print(tag, "launches/step", round(sum(v[0] for v in agg.values()) / steps, 1), "kernel us/step", round(tot / steps, 1),
      "wall us/step", round((t1 - t0) / 1e3 / steps, 1), "own %", round(100 * mine / tot, 1))
