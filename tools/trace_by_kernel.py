"""Groups a rocprofv3 kernel trace by (kernel, grid size): launches, average and total duration.
    python tools/trace_by_kernel.py <dir with *kernel_trace.csv> [min_total_ms]"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    acc[(name[:70], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
print(f"{'kernel':70s} {'blocks':>7s} {'n':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / 1e3 >= min_ms:
        print(f"{k:70s} {g:7d} {len(v):6d} {sum(v) / len(v):9.1f} {sum(v) / 1e3:9.2f} {100 * sum(v) / tot:6.1f}")
print(f"total {tot / 1e3:.2f} ms")
