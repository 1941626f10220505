cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/eager_steps.py c2 300 graph 2>&1 | grep -o "ms_per_step\": [0-9.]*"; done
mkdir -p gpurun_out/seq
timeout -s KILL 300 rocprofv3 --kernel-trace -d gpurun_out/seq/t -o seq -- python tools/eager_steps.py c2 6 graph > /dev/null 2>&1
python tools/step_sequence.py $(ls gpurun_out/seq/t/*.db | head -1) > gpurun_out/seq/step_sequence.txt 2>&1
rm -rf gpurun_out/seq/t
head -1 gpurun_out/seq/step_sequence.txt; grep "proj_\|lift_\|grouped\|attn_bwd" gpurun_out/seq/step_sequence.txt
