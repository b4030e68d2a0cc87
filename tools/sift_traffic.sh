#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/sifttraffic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PROBE_ONLY_BLOCK=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 $R/tools/sift_probe.py > $out/f.log 2> $out/f.err || echo fail1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/w -- python3 $R/tools/sift_probe.py > $out/w.log 2> $out/w.err || echo fail2
cd $R
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for d in ("f", "w"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "sift_scores_batch" not in k: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        for n, v in sorted(c.items()):
            print(k, n, "per launch:", [round(x) for x in v])
PY
find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete
