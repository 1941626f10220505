#!/bin/bash
# kernel-trace summary of the bench step only: bash tools/trace_bench.sh <tag> [extra bench args]
TAG=${1:-t}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-configs --no-variants "$@" > $OUT/bench_line.json 2> $OUT/bench.err
python tools/rocpd_stats.py $(ls $OUT/trace/*.db | head -1) 0 > $OUT/kernel_stats.txt 2>> $OUT/bench.err
rm -rf $OUT/trace
head -60 $OUT/kernel_stats.txt
