#!/bin/bash
# Per-kernel time of the training step on one MI355X: rocprofv3 kernel trace of the bench command, summarised.
# Usage (through gpurun, from the repository root): bash tools/profile_step.sh <tag> [bench.py arguments]
set -e
R=$(pwd)
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline "$@" > $O/bench_under_rocprof.json 2> $O/kt.err
python3 $R/tools/prof_summary.py $O/kt 16 > $O/summary.txt
cat $O/bench_under_rocprof.json >> $O/summary.txt
find $O -name '*kernel_trace.csv' -size +20M -delete
find $O -name '*.db' -delete
cat $O/summary.txt | cut -c1-170
