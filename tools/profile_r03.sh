#!/bin/bash
# Evidence run on one MI355X: the bench lines, per-kernel time of the training step and of the decode loop,
# HBM-side byte counters and matrix-pipe busy counters (each counter set in a --pmc pass of its own), the per-shape
# GEMM table, the vendor library on the same shapes and the in-kernel clock under load.  Run from the repository root
# through gpurun; results land in gpurun_out/r03/.  Every GPU step has its own time limit; the script stops at the
# first failure.   Usage: bash tools/profile_r03.sh [train|decode|all]
set -e
R=$(pwd)
O=$R/gpurun_out/r03
WHAT=${1:-all}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
if [ "$WHAT" != decode ]; then
step bench;        timeout -k 10 420 python3 $R/bench.py > $O/bench_cfg2.json 2> $O/bench.err
step kt_train;     timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python3 $R/bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt_train.err
python3 $R/tools/prof_summary.py $O/kt_train 16 > $O/bench_cfg2_summary.txt
step pmc_fetch;    timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-roofline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
step pmc_write;    timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-roofline > $O/pmc_write.json 2> $O/pmc_write.err
python3 $R/tools/pmc_hbm_traffic.py $O/pmc_fetch $O/pmc_write 6 $O/pmc_hbm_traffic.json > $O/pmc_hbm_traffic.txt
step pmc_mfma;     timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-roofline > $O/pmc_mfma.json 2> $O/pmc_mfma.err
python3 $R/tools/pmc_mfma_util.py $O/pmc_mfma > $O/pmc_mfma_util.txt
step shapes;       timeout -k 10 200 python3 $R/tools/gemm_shapes.py cfg2 5 > $O/gemm_shapes.txt 2> $O/shapes.err
step yardstick;    timeout -k 10 200 python3 $R/tools/bench_rowpanel.py --vendor > $O/gemm_vendor_yardstick.txt 2> $O/yard.err
step attn;         timeout -k 10 120 python3 $R/tools/bench_attn.py > $O/attn_bench.txt 2> $O/attn.err
step clock;        SKHIP_LIB=$R/specialk_amd/libspecialk_hip_stamps.so timeout -k 10 120 python3 $R/tools/rowpanel_stamps.py > $O/inkernel_clock.txt 2> $O/clock.err
step cfg1;         timeout -k 10 300 python3 $R/bench.py --config cfg1 --no-decode > $O/bench_cfg1.json 2> $O/cfg1.err
step cfg4;         timeout -k 10 300 python3 $R/bench.py --config cfg4 --no-decode > $O/bench_cfg4.json 2> $O/cfg4.err
fi
if [ "$WHAT" != train ]; then
step kt_decode;    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_decode -- python3 $R/bench.py --decode-only > $O/decode_under_rocprof.json 2> $O/kt_decode.err
python3 $R/tools/prof_summary.py $O/kt_decode 600 > $O/decode_summary.txt || true
export SK_DECODE_GRAPH=0   # counters are attributed per launch; the kernels are the same ones the graph replays
step pmc_dfetch;   timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_dfetch -- python3 $R/bench.py --decode-only > $O/pmc_dfetch.json 2> $O/pmc_dfetch.err
step pmc_dwrite;   timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_dwrite -- python3 $R/bench.py --decode-only > $O/pmc_dwrite.json 2> $O/pmc_dwrite.err
python3 $R/tools/pmc_decode_traffic.py $O/pmc_dfetch $O/pmc_dwrite 600 $O/pmc_decode_hbm_traffic.json > $O/pmc_decode_hbm_traffic.txt
fi
# the raw traces are large: keep the statistics and counter tables only
find $O -name '*kernel_trace.csv' -size +20M -delete
find $O -name '*.db' -delete
echo done
