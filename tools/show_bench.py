"""Short view of a bench.py JSON line.  usage: python tools/show_bench.py FILE [FILE ...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c, r = d["config"], d["roofline"]
    b1, ns = c.get("batch1_configs1_literal") or {}, c.get("nuscenes_shaped_configs2") or {}
    print("%s: %.1f frames/s (%.2f ms)  batch1 %s (%s ms)  nuscenes %s  frac %.4f" % (
        f, d["value"], d["ms_per_step"], b1.get("frames_per_sec"), b1.get("ms_per_step"), ns.get("frames_per_sec"), r["frac"]))
    for k, v in r["per_kernel"].items():
        print("   %-22s n %3d  %.3f ms  avg %.1f us" % (k, v["launches"], v["ms"], v["avg_us"]))
    for k, v in (r.get("other_kernel_groups") or {}).items():
        print("   %-22s %s %s  frac %.3f  %s" % (k.split(" ")[0], v["achieved"], v["unit"], v["frac"], {x: v[x] for x in ("ms", "launches", "us_per_block_fwd", "us_per_block_bwd") if x in v}))
