#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_nn.sh <tag> [nn_probe args...]
# kernel trace + stats, then PMC passes in their own runs (never combined with other trace domains).
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/nn_probe.py "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc1 -- python3 $R/tools/nn_probe.py "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc2 -- python3 $R/tools/nn_probe.py "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc3 -- python3 $R/tools/nn_probe.py "$@" > $out/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc4 -- python3 $R/tools/nn_probe.py "$@" > $out/pmc4.log 2>&1
cd $R
python3 tools/prof_summary.py $out k_nn_brick k_nn_fallback k_brick > $out/summary.txt 2>&1
# keep only summaries + the stats csv (the raw per-dispatch csvs are large)
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -size +2M -delete
cat $out/summary.txt
