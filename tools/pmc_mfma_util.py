"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE run:
busy cycles summed over the 1024 SIMDs / (1024 x cycles the chip was active); GRBM_GUI_ACTIVE is the
sum over the 8 XCDs.  Usage: pmc_mfma_util.py <dir>"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    per[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
    per[r["Dispatch_Id"]]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for d in per.values():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    a = agg[d["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]]
    a[0] += 1; a[1] += d["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += d["GRBM_GUI_ACTIVE"]; a[3] += d["ns"]
print(f"{'kernel':70s} {'calls':>6s} {'MFMA busy':>10s} {'clock GHz':>9s}")
for k, (n, busy, act, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if busy <= 0:
        continue
    print(f"{k:70s} {n:6d} {busy / (1024 * act / 8):10.3f} {act / 8 / ns:9.2f}")
