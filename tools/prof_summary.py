"""Summarise rocprofv3 CSV output: python tools/prof_summary.py <dir> [kernel-substring ...]
kernel trace -> per-kernel count / avg us;  counter collection -> per-kernel mean of each counter."""
import csv, glob, os, sys, collections
d = sys.argv[1]
filt = sys.argv[2:]
def keep(name): return (not filt) or any(f in name for f in filt)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# kernel trace", os.path.relpath(f, d))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if keep(k): print("%-60s n=%5d avg=%10.2f us min=%10.2f us total=%10.2f ms" % (k[:60], len(v), sum(v) / len(v), min(v), sum(v) / 1e3))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# counters", os.path.relpath(f, d))
    for k, cs in agg.items():
        if not keep(k): continue
        print(k[:70])
        for c, v in sorted(cs.items()): print("    %-28s mean=%16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
