set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_mr; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/tools/run_mr.py > $O/kt.log 2>&1
cd $R; python3 - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/r3_mr/**/kt_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=re.sub(r"^void ","",r["Name"]).split("(")[0]
    if n.startswith("mr_"): print(f"{n:30s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
for set in "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40); cd /tmp
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace -d $O/$tag -o p --output-format csv -- python3 $R/tools/run_mr.py > $O/$tag.log 2>&1 || echo failed $tag
done
cd $R; mkdir -p $O/kt; cp $(find $O -maxdepth 2 -name "kt_kernel_trace.csv" | head -1) $O/kt/ 2>/dev/null; python3 tools/pmc_table.py $O | grep -E "kernel|mr_"
