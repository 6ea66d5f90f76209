"""Condense rocprofv3 CSV output into the small summaries committed under profiles/.
usage: summarize_rocprof.py <dir with *kernel_stats.csv | *counter_collection.csv> [--filter substr] [--all]
(--all: every kernel of the trace, not only the top of the list plus libmatpbr.so's own)"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else None
everything = "--all" in sys.argv


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# kernel stats ({os.path.relpath(f, d)}): total {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches")
    print("name,calls,avg_us,min_us,max_us,pct")
    OWN = ("lazy_", "shade_", "relight", "loss_", "adam", "light_grad", "colsum", "normals_from", "eval_brdf", "sample_brdf", "sh_eval", "jac_bwd", "env_", "diffuse_cache", "mlp_")
    for i, r in enumerate(rows):
        if everything or i < 14 or any(k in r["Name"] for k in OWN):       # the top of the list plus every kernel of libmatpbr.so
            print(f"{short(r['Name'])},{r['Calls']},{float(r['AverageNs']) / 1e3:.2f},{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},{r['Percentage']}")
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if flt and flt not in k:
            continue
        agg[(k, r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"# counters ({os.path.relpath(f, d)})")
    print("kernel,grid,vgpr,sgpr,scratch,counter,dispatches,mean,min,max,mean_of_last_40")
    for key, cs in agg.items():
        for c, v in cs.items():
            tail = v[-40:]      # rows are in dispatch order: the timed steps of a `--steps 40` run (after its warm-up)
            print(",".join(key) + f",{c},{len(v)},{sum(v) / len(v):.4f},{min(v):.4f},{max(v):.4f},{sum(tail) / len(tail):.4f}")
