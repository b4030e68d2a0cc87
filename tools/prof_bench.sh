#!/bin/bash
# usage (GPU box, repo root): tools/prof_bench.sh <tag>
# rocprofv3 kernel trace + stats of the default bench command, then PMC passes (own runs, no other trace domain).
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/profbench_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py $ARGS > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py $ARGS > $out/bench_pmc1.json 2> $out/pmc1.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 $R/bench.py $ARGS > $out/bench_pmc2.json 2> $out/pmc2.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc_sq -- python3 $R/bench.py $ARGS > $out/bench_pmc3.json 2> $out/pmc3.err
cd $R
python3 tools/prof_summary.py $out k_nn_ k_ba_ k_brick k_associate k_prepare k_finalize > $out/summary.txt 2>&1
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete; find $out -name "*domain_stats.csv" -delete
head -c 3000 $out/summary.txt
