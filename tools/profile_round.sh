#!/bin/bash
# One profiling pass for a round: kernel-trace stats of the bench (hipGraph) and separate --pmc passes over eager steps of the
# bench configuration (C2) and of the skewed vx configuration (C3).  Runs ON the GPU box (gpurun -- 'bash tools/profile_round.sh r2x').
# Outputs land in gpurun_out/<tag>/; copy the summaries to profiles/.
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
P="timeout -s KILL 300 rocprofv3"
$P --kernel-trace -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-configs --no-variants > $OUT/bench_line.json 2> $OUT/bench.err
python tools/rocpd_stats.py $(ls $OUT/trace/*.db | head -1) 0 > $OUT/kernel_stats.txt 2>> $OUT/bench.err
# the launches of ONE hipGraph-replayed step, in order (dispatch count, per-kernel durations, gaps)
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace_seq -o seq -- python tools/eager_steps.py c2 6 graph > /dev/null 2>> $OUT/bench.err
python tools/step_sequence.py $(ls $OUT/trace_seq/*.db | head -1) > $OUT/step_sequence.txt 2>> $OUT/bench.err
rm -rf $OUT/trace_seq
# PMC passes: one counter group per run, eager steps, no other tracing (gpurun refuses --pmc with sys/hip traces)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  $P --kernel-trace --pmc $grp -d $OUT/pmc_c2_$name -o p -- python tools/eager_steps.py c2 3 > /dev/null 2>> $OUT/pmc.err
  $P --kernel-trace --pmc $grp -d $OUT/pmc_c3_$name -o p -- python tools/eager_steps.py c3 3 > /dev/null 2>> $OUT/pmc.err
done
python tools/pmc_report.py $OUT/pmc_c2.json $(ls $OUT/pmc_c2_*/*.db) >> $OUT/pmc.err 2>&1
python tools/pmc_report.py $OUT/pmc_c3.json $(ls $OUT/pmc_c3_*/*.db) >> $OUT/pmc.err 2>&1
$P --kernel-trace -d $OUT/trace_c3 -o c3 -- python tools/eager_steps.py c3 10 graph > $OUT/c3_line.json 2>> $OUT/bench.err
python tools/rocpd_stats.py $(ls $OUT/trace_c3/*.db | head -1) > $OUT/c3_kernel_stats.txt 2>> $OUT/bench.err
python tools/step_sequence.py $(ls $OUT/trace_c3/*.db | head -1) > $OUT/c3_step_sequence.txt 2>> $OUT/bench.err
rm -rf $OUT/trace $OUT/trace_c3 $OUT/pmc_c2_* $OUT/pmc_c3_*
ls -la $OUT
