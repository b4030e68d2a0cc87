"""(GPU box) profiles/traffic.json from the PMC passes of tools/prof_bench.sh:
  python tools/make_traffic.py <profbench dir> <cloud points> <queries>
Per launch of the dominant NN kernel: bytes = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB; FETCH_SIZE
under-reports wide coalesced reads by 2 on gfx950 -- MI355X_MICROARCH.md section HBM), L2 hit rate.
The file carries the hash of the kernel sources it was measured on; bench.py quotes it only when that matches."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d, cloud, queries = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
# every kernel of the step whose roofline entry bench.py prints (substring of the kernel name)
KERNELS = ["k_nn_brick", "k_nn_fallback", "k_ba_images", "k_ba_points", "k_ba_cost", "k_associate", "k_bk_", "k_fb_compact"]

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if any(k in name for k in KERNELS):
            # (template instances stay apart: "void pcd::k_ba_points<4, true>" ...)
            agg[name.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_source": "tools/prof_bench.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum / --pmc SQ_* "
                  "passes of `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras`",
       "_correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), "
                      "WRITE_SIZE exact; both in KiB", "workload": [cloud, queries], "kernels": {}}
import bench  # noqa: E402  (source_hash only; does not touch the GPU)
out["source_hash"] = bench.source_hash()
for k, cs in sorted(agg.items()):
    mean = lambda n: sum(cs[n]) / len(cs[n]) if cs.get(n) else None
    fk, wk = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    hit, miss = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
    b = (2 * (fk or 0) + (wk or 0)) * 1024
    out["kernels"][k] = dict(fetch_size_kib=fk, write_size_kib=wk, bytes_per_launch=b if fk is not None else None,
                             l2_hit_rate=(hit / (hit + miss)) if hit is not None and miss else None,
                             valu_wave_instructions=mean("SQ_INSTS_VALU"), wave_cycles=mean("SQ_WAVE_CYCLES"),
                             wait_any=mean("SQ_WAIT_ANY"), wait_inst_any=mean("SQ_WAIT_INST_ANY"),
                             active_inst_any=mean("SQ_ACTIVE_INST_ANY"),
                             launches_seen=len(cs.get("FETCH_SIZE", [])))
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
