"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE run.
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs and counts SHADER clocks; the launch duration is in
nanoseconds; the shader clock under load is not what GRBM_GUI_ACTIVE / duration suggests (that ratio comes out between
2.0 and 2.9 GHz, above the part's maximum: the counter runs in another clock domain), so utilisation is given for the
clock measured inside the kernels under the step's load (s_memtime against s_memrealtime, tools/rowpanel_stamps.py:
1.48 GHz for the MFMA-heavy kernels) and for the 2.4 GHz the peak figure assumes.
Usage: pmc_mfma_util.py <dir> [clock GHz under load = 1.48]"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 1.48
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    per[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
    per[r["Dispatch_Id"]]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for d in per.values():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    a = agg[d["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:62]]
    a[0] += 1; a[1] += d["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += d["ns"]
print(f"{'kernel':62s} {'calls':>6s} {'avg us':>8s} {'busy clk/SIMD':>13s} {'util @' + format(ghz, '.2f'):>10s} {'util @2.40':>10s}")
for k, (n, busy, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if busy <= 0:
        continue
    per_simd, us = busy / 1024 / n, ns / n / 1e3
    print(f"{k:62s} {n:6d} {us:8.1f} {per_simd:13.0f} {per_simd / (us * ghz * 1e3):10.3f} {per_simd / (us * 2.4e3):10.3f}")
