"""Memory-side counters of the 256-tile bf16 GEMM kernel on a short-K training shape against a long-K cube.
  run:   pmc_gemm_mem.py run        (the workload, under rocprofv3 --pmc ...)
  show:  pmc_gemm_mem.py show <dir> [<dir> ...]"""
import collections, csv, glob, os, sys

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update({"SKHIP_GEMM_BIG": "1", "SKHIP_GEMM_PERSIST": "0"})
    from specialk_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(0)
    bias = torch.randn(8192, device="cuda")
    for M, N, K in ((25639, 2048, 512), (25639, 512, 2048), (8192, 8192, 8192)):
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        B = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for _ in range(6):
            L.gemm(L.GEMM_NT, A, B, M, N, K, K, K, out=out, ldc=N, bias=bias[:N])
        torch.cuda.synchronize()
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_kernel" not in r["Kernel_Name"]:
                    continue
                key = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[key][r["Counter_Name"]] += 1
    names = sorted({k for a in agg.values() for k in a})
    keys = sorted(agg)
    print(f"{'counter (per launch)':40s}" + "".join(f"{'blocks=' + str(k):>18s}" for k in keys))
    for n in names:
        print(f"{n:40s}" + "".join(f"{agg[k][n] / max(cnt[k][n], 1):18.0f}" for k in keys))
    def ratio(a, b, label):
        vals = []
        for k in keys:
            x, y = agg[k].get(a), agg[k].get(b)
            vals.append(f"{x / y:18.3f}" if x is not None and y else f"{'-':>18s}")
        print(f"{label:40s}" + "".join(vals))
    ratio("TCC_HIT_sum", "TCC_REQ_sum", "L2 hit / req")
    ratio("TCC_MISS_sum", "TCC_REQ_sum", "L2 miss / req")
    ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "avg L1->L2 read latency (clk)")
    ratio("TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum", "avg L1->L2 write latency (clk)")
