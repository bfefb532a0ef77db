"""Collapse rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12:
            print("  ", ",".join(c[:60] for c in row))

for f in find("*counter_collection.csv"):
    print("== counters:", os.path.relpath(f, root))
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    rd = csv.DictReader(open(f))
    for row in rd:
        k = row.get("Kernel_Name", "?")[:70]
        c = row.get("Counter_Name")
        v = float(row.get("Counter_Value", 0) or 0)
        acc[k][c] += v
        cnt[k][c] += 1
    for k in acc:
        print("  kernel:", k)
        for c in sorted(acc[k]):
            print(f"     {c:36s} total {acc[k][c]:.4e}  per-dispatch {acc[k][c] / max(cnt[k][c], 1):.4e}  (n={cnt[k][c]})")
