#!/bin/bash
# Round-2 evidence on one MI355X: the bench line, per-kernel time of the training step and of the decode loop,
# and HBM-side byte counters (separate --pmc passes).  Run from the repository root through gpurun; results land
# in gpurun_out/r02/.  Every GPU step has its own time limit and the script stops at the first failure.
set -e
R=$(pwd)
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $1"; }
step bench;        timeout -k 10 420 python3 $R/bench.py > $O/bench_cfg2.json 2> $O/bench.err
step kt_train;     timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python3 $R/bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt_train.err
step kt_decode;    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_decode -- python3 $R/bench.py --decode-only > $O/decode_under_rocprof.json 2> $O/kt_decode.err
step pmc_fetch;    timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-roofline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
step pmc_write;    timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-roofline > $O/pmc_write.json 2> $O/pmc_write.err
export SK_DECODE_GRAPH=0   # counters are attributed per launch; the kernels are the same ones the graph replays
step pmc_dfetch;   timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_dfetch -- python3 $R/bench.py --decode-only > $O/pmc_dfetch.json 2> $O/pmc_dfetch.err
step pmc_dwrite;   timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_dwrite -- python3 $R/bench.py --decode-only > $O/pmc_dwrite.json 2> $O/pmc_dwrite.err
# the raw traces are large: keep the statistics and counter tables only
find $O -name '*kernel_trace.csv' -size +20M -delete
echo done
