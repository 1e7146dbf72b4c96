"""Per-dispatch SQ counters next to the kernel's duration, from one rocprofv3 run:
  rocprofv3 --pmc <counters...> --kernel-trace --output-format csv -d DIR -- python3 tools/bench_spconv.py --iters 1 ...
usage: python tools/pmc_sq.py DIR [name substring ...]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles summed over waves (MI355X_MICROARCH.md); they are printed as a share of
SQ_WAVE_CYCLES.  mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock), clock = SQ_BUSY_CYCLES / 32 SEs / duration."""
import collections, csv, glob, os, sys

d = sys.argv[1]
subs = sys.argv[2:]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    e = rows.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", ""), "vgpr": r.get("VGPR_Count", ""), "lds": r.get("LDS_Block_Size", "")})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = []
for e in rows.values():
    for k in e:
        if k not in ("name", "grid", "vgpr", "lds") and k not in names:
            names.append(k)
print("%-44s %9s %5s %6s %8s %6s %6s | %s" % ("kernel", "grid", "vgpr", "lds", "dur_us", "clkGHz", "mfma%", "  ".join(n.replace("SQ_", "") for n in names)))
for did, e in rows.items():
    if subs and not any(s in e["name"] for s in subs):
        continue
    t = dur.get(did, float("nan"))
    clk = e.get("SQ_BUSY_CYCLES", float("nan")) / 32.0 / (t * 1e3) if t == t else float("nan")
    mf = 100.0 * e.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (1024.0 * t * 1e3 * clk) if t == t else float("nan")
    wc = e.get("SQ_WAVE_CYCLES", 0.0)
    cols = []
    for n in names:
        v = e.get(n, float("nan"))
        if n.startswith("SQ_WAIT") or n.startswith("SQ_ACTIVE_INST"):
            cols.append("%5.1f%%" % (100.0 * v / wc) if wc else "nan")
        else:
            cols.append("%.3g" % v)
    short = e["name"].split("(")[0].replace("void ", "")[:44]
    print("%-44s %9s %5s %6s %8.1f %6.2f %6.1f | %s" % (short, e["grid"], e["vgpr"], e["lds"], t, clk, mf, "  ".join(cols)))
