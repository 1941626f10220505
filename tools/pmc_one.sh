#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only) over ANY command; report for kernels matching a pattern.
# usage (on the GPU box): bash tools/pmc_one.sh <tag> <kernel substring> -- <command ...>
TAG=$1; PAT=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p_$i -o p -- "$@" > /dev/null 2>> $OUT/err.txt
done
python tools/pmc_report.py $OUT/pmc.json $(ls $OUT/p_*/*.db $OUT/p_*/*/*.db 2>/dev/null) >> $OUT/err.txt 2>&1
rm -rf $OUT/p_*
PAT="$PAT" OUTJ=$OUT/pmc.json python - <<'PY'
import json, os
d = json.load(open(os.environ["OUTJ"]))
for k, r in d.items():
    if os.environ["PAT"] not in k: continue
    a = r["raw_avg"]; cu = a.get("SQ_BUSY_CU_CYCLES", 0); wc = a.get("SQ_WAVE_CYCLES", 0)
    print(k, {x: (round(v, 3) if isinstance(v, float) else v) for x, v in r.items() if x != "raw_avg"})
    print("   raw:", a)
PY
