"""Launches of ONE steady-state step that cannot fill the chip: fewer workgroups than 2 per CU (512) and longer than a threshold, grouped
by kernel and grid, from a rocprofv3 --kernel-trace run of bench.py (same cut as tools/step_kernels.py).
usage: python tools/underfilled.py DIR [min_us=12]"""
import collections, csv, glob, os, sys

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "loss_main_kernel" in r["Kernel_Name"]]
step = rows[marks[-2]:marks[-1]]


def blocks(r):
    g = [int(r.get("Grid_Size_" + a, r.get("Grid_Size", 1)) or 1) for a in "XYZ"] if "Grid_Size_X" in r else [int(r["Grid_Size"]), 1, 1]
    w = [int(r.get("Workgroup_Size_" + a, 1) or 1) for a in "XYZ"] if "Workgroup_Size_X" in r else [int(r["Workgroup_Size"]), 1, 1]
    return (g[0] // max(w[0], 1)) * (g[1] // max(w[1], 1)) * (g[2] // max(w[2], 1))


acc = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    b = blocks(r)
    if b < 512 and us >= min_us:
        name = r["Kernel_Name"].replace("void ", "").split("(")[0][:70]
        a = acc[(name, b)]
        a[0] += 1
        a[1] += us
tot = sum(v[1] for v in acc.values())
print("launches of one step with < 512 workgroups and >= %.0f us: %.0f us in total" % (min_us, tot))
print("%-72s %7s %5s %9s %8s" % ("kernel", "blocks", "n", "total us", "avg us"))
for (name, b), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s %7d %5d %9.1f %8.1f" % (name, b, n, us, us / n))
