"""Per-launch durations of one kernel from a rocprofv3 kernel trace: percentiles, and the sequence (with the gap to the kernel before).
usage: step_durations.py <dir with *_kernel_trace.csv> <kernel substr> [first] [count]"""
import csv
import glob
import sys

import numpy as np

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sub = sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 200
count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
dur, gap, prev = [], [], []
for i, r in enumerate(rows):
    if sub in r["Kernel_Name"]:
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        gap.append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i else 0.0)
        prev.append((int(rows[i - 1]["End_Timestamp"]) - int(rows[i - 1]["Start_Timestamp"])) / 1e3 if i else 0.0)
d = np.array(dur)
print(f"{len(d)} launches: mean {d.mean():.1f} us, percentiles 1/10/25/50/75/90/99 = " + " ".join(f"{np.percentile(d, p):.1f}" for p in (1, 10, 25, 50, 75, 90, 99)))
print("histogram (10 us bins):", dict(zip(*[x.tolist() for x in np.unique((d // 10 * 10).astype(int), return_counts=True)])))
print("sequence from launch", first, "(duration / gap before / duration of the kernel before):")
print(" ".join(f"{a:.0f}/{g:.1f}/{p:.0f}" for a, g, p in zip(dur[first:first + count], gap[first:first + count], prev[first:first + count])))
