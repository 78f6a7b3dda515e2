#!/bin/bash
# rocprofv3 evidence for the bf16 eval network (BASELINE configs[4]) at 216 tiles: kernel trace + one PMC pass per counter set
# (each in its own run; never combined with other trace domains).  Usage: bash tools/pmc_bf16.sh <out dir under the repo> [batch]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/$1; BATCH=${2:-216}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/tools/run_eval.py $BATCH bf16"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $CMD > $O/kt.log 2>&1 || echo "kt failed"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-48)
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p --output-format csv -- $CMD > $O/$tag.log 2>&1 || echo "failed $tag"
  echo "$tag done"
done
cd $R
python3 tools/pmc_table.py $O > $O/summary.txt
rm -rf $O/*/*/*.db
find $O -name "*kernel_trace.csv" -path "*SQ_*" -delete; find $O -name "*kernel_trace.csv" -path "*SIZE*" -delete
cat $O/summary.txt
