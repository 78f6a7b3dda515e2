#!/bin/bash
# Collects the round's rocprofv3 evidence for bench.py --mode train (batch 64): kernel trace (two-stream and one-stream) and the
# three PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy), each in its own run.  Usage: bash tools/final_profiles.sh <out dir>
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-layers"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.log 2>&1
export SVS_TRAIN_ONE_STREAM=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt1 --output-format csv -- $B > $O/kt1.log 2>&1
unset SVS_TRAIN_ONE_STREAM
B="python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-layers"
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f --output-format csv -- $B > $O/pmc_fetch.log 2>&1
echo fetch done
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w --output-format csv -- $B > $O/pmc_write.log 2>&1
echo write done
timeout -k 10 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -o m --output-format csv -- $B > $O/pmc_mfma.log 2>&1
echo mfma done
cd $R
python3 tools/prof_steps_csv.py $O/kt/kt_kernel_trace.csv 8 > $O/per_step.txt
python3 tools/prof_steps_csv.py $O/kt1/kt1_kernel_trace.csv 8 > $O/per_step_one_stream.txt
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
python3 tools/pmc_mfma.py $O/pmc_mfma > $O/pmc_mfma_busy.json
rm -rf $O/pmc_fetch/*/*.db $O/pmc_write/*/*.db $O/pmc_mfma/*/*.db
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
head -12 $O/per_step.txt
