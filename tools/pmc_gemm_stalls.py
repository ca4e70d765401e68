"""Where the cycles of the bf16 GEMM kernels go, from rocprofv3 --pmc passes.
  run:      pmc_gemm_stalls.py run            (the workload: three training shapes, 20 launches each)
  analyse:  pmc_gemm_stalls.py show <dir> [<dir> ...]   (per kernel and grid: counter sums per launch and ratios)
Average VMEM latency = SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM (cycles an instruction is outstanding)."""
import collections, csv, glob, os, sys

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from specialk_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K in ((25639, 2048, 512), (25639, 512, 512), (25639, 512, 2048)):
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        B = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for _ in range(20):
            L.gemm(L.GEMM_NT, A, B, M, N, K, K, K, out=out, ldc=N)
        torch.cuda.synchronize()
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_kernel" not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"].split("::")[0].replace("void ", ""), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[key][r["Counter_Name"]] += 1
    for key in sorted(agg):
        c = {k: v / cnt[key][k] for k, v in agg[key].items()}
        print(f"{key[0]} blocks={key[1]}")
        for k in sorted(c):
            print(f"    {k:28s} {c[k]:16.0f}")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_MISC"):
                if k in c:
                    print(f"    {k + ' / WAVE_CYCLES':38s} {c[k] / wc:8.3f}")
        if c.get("SQ_INSTS_VMEM"):
            print(f"    avg VMEM latency (cycles)              {c.get('SQ_INST_LEVEL_VMEM', 0) / c['SQ_INSTS_VMEM']:8.0f}")
        if c.get("SQ_INSTS_LDS"):
            print(f"    avg LDS latency (cycles)               {c.get('SQ_INST_LEVEL_LDS', 0) / c['SQ_INSTS_LDS']:8.0f}")
