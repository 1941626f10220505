#!/bin/bash
# Cache-path PMC passes (L1 -> L2 request counts, L2 hit / miss, stall cycles) over ANY command; one counter group per run.
# usage (on the GPU box): bash tools/pmc_cache.sh <tag> <kernel substring> -- <command ...>
TAG=$1; PAT=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_CYCLE_sum TCC_BUSY_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
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
    print(k, "us", round(r.get("avg_us_under_pmc", 0), 1))
    for x, v in sorted(r["raw_avg"].items()): print(f"    {x:44s} {v:16.1f}")
PY
