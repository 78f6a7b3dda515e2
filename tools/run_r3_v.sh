set -o pipefail
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in "X=1" "SVS_BF16_CFG=6" "SVS_BF16_CFG=6 SVS_BF16_KSPLIT=1"; do
  tag=$(echo "t256 $cfg" | tr ' =' '__')
  O=$R/gpurun_out/r3_bf16_sweep/$tag; mkdir -p $O
  env $cfg timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/tools/run_eval.py 216 bf16 > $O.log 2>&1 || echo failed $tag
  echo "== $cfg"; python3 - <<PY
import csv,glob,re
f=glob.glob("$O/**/kt_kernel_stats.csv",recursive=True)
tot=0
for r in csv.DictReader(open(f[0])):
    n=re.sub(r"^void ","",r["Name"]).split("(")[0]
    if "bf16" in n and "pack" not in n:
        if "gemm" in n or "splitk" in n: print(f"  {n:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us")
        tot+=float(r['TotalDurationNs'])/8e3
print(f"  sum per forward {tot:.1f} us")
PY
done
