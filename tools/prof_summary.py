"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and, for the
GEMM kernels, per-grid-size averages.  Usage: prof_summary.py <dir> [steps]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ks = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(ks)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.2f} ms/step")
for r in rows[:22]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:72]
    print(f"{name:72s} calls {int(r['Calls']):6d} ms/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} avg_us {float(r['AverageNs']) / 1e3:9.1f} {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
