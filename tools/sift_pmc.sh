#!/bin/bash
# usage (GPU box, repo root): tools/sift_pmc.sh <tag>   -- SQ counters of the SIFT scores kernel on the 50-image block
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/siftpmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PROBE_ONLY_BLOCK=1
run() { d=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/$d -- python3 $R/tools/sift_probe.py > $out/$d.log 2> $out/$d.err || echo "pass $d failed"; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run p2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run p0 GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_CYCLES
run p3 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/sift_probe.py > $out/trace.log 2> $out/trace.err
cd $R
find $out/trace -name "*kernel_stats.csv" -exec head -8 {} \;
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for d in ("p0", "p1", "p2", "p3"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sift_scores" not in k: continue
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for c, (v, n) in sorted(agg.items()):
        print("%-28s %16.0f per launch (%d launches)" % (c, v / max(n, 1), n))
PY
find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
