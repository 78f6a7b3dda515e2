set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sig; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p --output-format csv -- python3 $R/tools/run_signal_kernels.py > $O/$tag.log 2>&1 || echo "failed $tag"
done
cd $R; for d in $O/*/; do python3 tools/pmc_summary.py $d stft 2>/dev/null; done
