#!/bin/bash
# PMC passes (one rocprofv3 run per counter group, kernel trace only) of a command, summarised per kernel.
#   scripts/pmc_run.sh <out-subdir under gpurun_out> <kernel-name filter> -- <command ...>
# Counter groups are kept at <= 4 SQ counters per pass.
set -u
out=$1; filt=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g -d $R/gpurun_out/$out/p$i -o p --output-format csv -- "$@" > $R/gpurun_out/$out.p$i.log 2>&1 || tail -5 $R/gpurun_out/$out.p$i.log
  i=$((i+1))
done
cd $R
python scripts/pmc_kernels.py $(find gpurun_out/$out -name "*counter_collection.csv") | grep -A40 "$filt"
