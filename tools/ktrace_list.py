"""Kernel durations in dispatch order from a rocprofv3 --kernel-trace run (csv), averaged over repeated identical dispatch runs.
usage: python tools/ktrace_list.py DIR [name substring ...]"""
import csv, glob, os, sys
d = sys.argv[1]; subs = sys.argv[2:]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"]
    if subs and not any(s in n for s in subs):
        continue
    short = n.split("(")[0].replace("void ", "")[:48]
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])) if "Grid_Size_X" in r else 0
    key = (short, g)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if out and out[-1][0] == key:
        out[-1][1].append(dur)
    else:
        out.append([key, [dur]])
for (short, g), ds in out:
    ds2 = ds[1:] if len(ds) > 2 else ds
    print("%-50s blocks %7d  n %3d  avg %8.1f us  min %8.1f" % (short, g, len(ds), sum(ds2) / len(ds2), min(ds)))
