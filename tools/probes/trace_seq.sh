set -e
R=$(pwd); O=$R/gpurun_out/seq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --steps 2 --warmup 2 --no-decode --no-cpu-baseline > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48] for r in rows]
# one step: between the last two adam_multi launches
ad = [i for i, n in enumerate(names) if n.startswith("adam_multi")]
a, b = ad[-2], ad[-1]
out = open("$O/one_step.txt", "w")
prev, cnt = None, 0
t0 = int(rows[a]["End_Timestamp"])
for i in range(a + 1, b + 1):
    r = rows[i]
    out.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {names[i]}\n")
out.close()
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
