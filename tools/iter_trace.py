"""Kernel sequence of one steady-state iteration from a rocprofv3 kernel trace.  usage: iter_trace.py <kernel_trace.csv> <marker substr> [index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 20
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[which], idx[which + 1]
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
busy = 0
print(f"span {span / 1e3:.1f} us, {b - a} kernels")
for r in rows[a:b]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:56]
    if d > 8000:
        print(f"{d / 1e3:8.1f}  {n}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
print(f"busy {busy / 1e3:.1f} us")
