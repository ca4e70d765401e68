#!/bin/bash
# Copies the summaries of gpurun_out/r03 (tools/profile_r03.sh) into profiles/ with the r03_ prefix.
set -e
O=gpurun_out/r03; P=profiles
cp $O/bench_cfg2.json $P/r03_bench_cfg2.json
cp $O/bench_cfg1.json $P/r03_bench_cfg1.json
cp $O/bench_cfg4.json $P/r03_bench_cfg4.json
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline (13 timed + 3 instrumented steps = 16)"; cat $O/bench_cfg2_summary.txt; echo; echo "# result printed by the same command:"; tail -1 $O/bench_under_rocprof.json; } > $P/r03_bench_cfg2_summary.txt
cp $(find $O/kt_train -name '*kernel_stats.csv' | head -1) $P/r03_bench_cfg2_kernel_stats.csv
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --decode-only (beam 5, 128 sentences, 150 steps, 4 batches = 600 steps)"; cat $O/decode_summary.txt; echo; echo "# result printed by the same command:"; tail -1 $O/decode_under_rocprof.json; } > $P/r03_decode_summary.txt
cp $(find $O/kt_decode -name '*kernel_stats.csv' | head -1) $P/r03_decode_kernel_stats.csv
cp $O/pmc_hbm_traffic.json $P/r03_pmc_hbm_traffic.json
{ echo "# rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-roofline ; the same with --pmc WRITE_SIZE (6 steps each); tools/pmc_hbm_traffic.py"; cat $O/pmc_hbm_traffic.txt; } > $P/r03_pmc_hbm_traffic.txt
cp $O/pmc_decode_hbm_traffic.json $P/r03_pmc_decode_hbm_traffic.json
{ echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-roofline ; tools/pmc_mfma_util.py"; python3 tools/pmc_mfma_util.py $O/pmc_mfma; } > $P/r03_pmc_mfma_util.txt
{ echo "# python3 tools/gemm_shapes.py cfg2 5 (every GEMM-family launch of a step bracketed by events; layout 0 = NT, 1 = NN, 2 = TN, grp = grouped weight gradients, panel = row-panel products)"; grep -v amdgpu.ids $O/gemm_shapes.txt; } > $P/r03_gemm_shapes.txt
{ echo "# python3 tools/bench_rowpanel.py --vendor"; grep -v amdgpu.ids $O/gemm_vendor_yardstick.txt; } > $P/r03_gemm_vendor_yardstick.txt
{ echo "# python3 tools/bench_attn.py"; grep -v amdgpu.ids $O/attn_bench.txt; } > $P/r03_attn_bench.txt
{ echo "# SKHIP_LIB=specialk_amd/libspecialk_hip_stamps.so python3 tools/rowpanel_stamps.py (clock = wave lifetime in s_memtime clocks / the same interval in s_memrealtime ticks of 10 ns, the stamped launch behind 40 launches of the same kernel)"; grep -v amdgpu.ids $O/inkernel_clock.txt; } > $P/r03_inkernel_clock.txt
ls -la $P | grep r03
