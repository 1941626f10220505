#!/bin/bash
# instruction-mix / pipe-busy counters for the bench configuration's eager steps (one --pmc pass per group)
export TAG=${1:-mix}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-48)
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p_$name -o p -- python tools/eager_steps.py c2 2 > /dev/null 2>> $OUT/err.txt
done
python tools/pmc_report.py $OUT/mix_c2.json $(ls $OUT/p_*/*.db $OUT/p_*/*/*.db 2>/dev/null) >> $OUT/err.txt 2>&1
rm -rf $OUT/p_*
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "%s", "mix_c2.json") % os.environ["TAG"]))
for k, r in d.items():
    if not any(s in k for s in ("attn_", "gemm_split", "gemm_glds_kernel<64,64,2,T,F", "kernel_mlp")): continue
    a = r["raw_avg"]
    wc, cu = a.get("SQ_WAVE_CYCLES", 0), a.get("SQ_BUSY_CU_CYCLES", 0)
    f = lambda n: (a.get(n, 0) / cu) if cu else 0
    print(f"{k:40s} per busy CU cycle: VALU-active {f('SQ_ACTIVE_INST_VALU'):.2f} LDS-active {f('SQ_ACTIVE_INST_LDS'):.2f} VMEM-active {f('SQ_ACTIVE_INST_VMEM'):.2f} MFMA-busy/4 {f('SQ_VALU_MFMA_BUSY_CYCLES')/4:.2f} | insts/launch VALU {a.get('SQ_INSTS_VALU',0):.0f} MFMA {a.get('SQ_INSTS_MFMA',0):.0f} LDS {a.get('SQ_INSTS_LDS',0):.0f} VMEM_RD {a.get('SQ_INSTS_VMEM_RD',0):.0f}")
PY
