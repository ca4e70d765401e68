#!/bin/bash
# The decode half of profile_r02.sh (after a change to the decode kernels only): bench line, per-kernel time of the
# training step and of the decode loop, HBM-side byte counters of the decode loop.  Results in gpurun_out/r02/.
set -e
R=$(pwd)
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
step bench;        timeout -k 10 420 python3 $R/bench.py > $O/bench_cfg2.json 2> $O/bench.err
step kt_train;     timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python3 $R/bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt_train.err
step kt_decode;    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_decode -- python3 $R/bench.py --decode-only > $O/decode_under_rocprof.json 2> $O/kt_decode.err
export SK_DECODE_GRAPH=0   # counters are attributed per launch; the kernels are the same ones the graph replays
step pmc_dfetch;   timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_dfetch -- python3 $R/bench.py --decode-only > $O/pmc_dfetch.json 2> $O/pmc_dfetch.err
step pmc_dwrite;   timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_dwrite -- python3 $R/bench.py --decode-only > $O/pmc_dwrite.json 2> $O/pmc_dwrite.err
find $O -name '*kernel_trace.csv' -size +20M -delete
echo done
