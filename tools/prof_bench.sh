#!/bin/bash
# usage (GPU box, repo root): tools/prof_bench.sh <tag>
# rocprofv3 kernel trace + stats of the default bench command, then PMC passes (own runs, no other trace domain),
# then profiles/traffic.json (PMC HBM bytes of the NN kernel, tagged with the hash of the kernel sources) and the
# summaries under profiles/<tag>_*.  Copy profiles/ back (it is inside gpurun_out/<tag> too).
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/profbench_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py $ARGS > $out/bench_trace.json 2> $out/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py $ARGS > $out/bench_pmc1.json 2> $out/pmc1.err || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 $R/bench.py $ARGS > $out/bench_pmc2.json 2> $out/pmc2.err || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/pmc_sq -- python3 $R/bench.py $ARGS > $out/bench_pmc3.json 2> $out/pmc3.err || exit 1
cd $R
python3 tools/prof_summary.py $out k_nn_ k_ba_ k_bk_ k_fb_ k_brick k_associate k_prepare k_finalize k_query > $out/summary.txt 2>&1
python3 tools/make_traffic.py $out 10000000 1000000 > $out/traffic.log 2>&1
cp profiles/traffic.json $out/traffic.json
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
find $out -name "*agent_info.csv" -delete; find $out -name "*domain_stats.csv" -delete
head -c 3000 $out/summary.txt
