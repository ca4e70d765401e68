"""HBM bytes per decode step from two rocprofv3 --pmc passes over `bench.py --decode-only` (FETCH_SIZE, WRITE_SIZE;
SK_DECODE_GRAPH=0 so that counters are attributed per launch): all kernels of the command summed and divided by the
steps, the largest contributors listed.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes).
Usage: pmc_decode_traffic.py <fetch_dir> <write_dir> <steps> <out.json>"""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]] += float(r["Counter_Value"])
    return agg


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
rd = {k: 2 * v * 1024 / steps for k, v in fetch.items()}
wr = {k: v * 1024 / steps for k, v in write.items()}
tot = {k: rd.get(k, 0) + wr.get(k, 0) for k in set(rd) | set(wr)}
top = sorted(tot, key=lambda k: -tot[k])[:4]
out = {"workload": "bench.py --decode-only (beam 5, 128 sentences, 150 steps, 4 batches), SK_DECODE_GRAPH=0, one-launch decoder stack",
       "steps": steps, "hbm_bytes_per_step": round(sum(tot.values())), "read_bytes_per_step": round(sum(rd.values())),
       "write_bytes_per_step": round(sum(wr.values())),
       "by_kernel_MB_per_step": {k: {"read": round(rd.get(k, 0) / 1e6, 1), "write": round(wr.get(k, 0) / 1e6, 1)} for k in top},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the whole command (the four encoder "
                 "passes included, <1 %); FETCH_SIZE doubled (gfx950)"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
