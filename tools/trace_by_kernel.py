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
# device idle time inside the traced span: gaps between the end of everything launched so far and the next kernel start
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(f)))
if len(sys.argv) > 3:  # steady state only: from the third-last to the last launch of the named kernel (e.g. rollout_h2)
    marks = [a for a, b, nm in ev if sys.argv[3] in nm]
    if len(marks) >= 3:
        ev = [e for e in ev if marks[-3] <= e[0] < marks[-1]]
        print(f"window: {(marks[-1] - marks[-3]) / 1e6:.3f} ms = 2 periods of {sys.argv[3]}")
idle, gaps, hi = 0, [], ev[0][1]
for a, b, nm in ev[1:]:
    if a > hi:
        idle += a - hi
        gaps.append(((a - hi) / 1e3, nm))
    hi = max(hi, b)
span = (hi - ev[0][0]) / 1e6
gaps.sort(reverse=True)
print(f"span {span:.2f} ms, idle {idle / 1e6:.2f} ms in {len(gaps)} gaps; largest (us, next kernel): " + ", ".join(f"{g:.0f} {n}" for g, n in gaps[:8]))
print(f"{'kernel':70s} {'blocks':>7s} {'n':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / 1e3 >= min_ms:
        print(f"{k:70s} {g:7d} {len(v):6d} {sum(v) / len(v):9.1f} {sum(v) / 1e3:9.2f} {100 * sum(v) / tot:6.1f}")
print(f"total {tot / 1e3:.2f} ms")
