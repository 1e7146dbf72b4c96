"""Kernel launches of ONE steady-state training step, from a rocprofv3 --kernel-trace run of bench.py (csv).

The trace is cut at a marker kernel that runs exactly once per step (the fused loss kernel); the LAST complete interval between two
markers is one full step (backward of step n-1, optimizer, forward of step n), free of the one-time launches of the first steps
(Adam state, TunableOp, graph capture warm-ups).  Prints launches and GPU time per kernel name, per stream, and the GPU idle time.
usage: python tools/step_kernels.py DIR [marker substring, default loss_main_kernel]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "loss_main_kernel"
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit("fewer than 3 marker kernels (%s) in the trace" % marker)
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])


def short(n):
    n = n.replace("void ", "")
    if n.startswith("Cijk"):
        return "hipBLASLt " + n[:24]
    if "rocprim" in n:
        for k in ("merge_sort", "radix_sort", "onesweep", "scan", "partition", "transform", "histogram", "lookback", "reduce"):
            if k in n:
                return "rocprim " + k
        return "rocprim other"
    if "at::native" in n:
        for k in ("FillFunctor", "CUDAFunctor_add", "layer_norm", "GammaBeta", "Gelu", "multi_tensor", "reduce_kernel", "direct_copy", "MulFunctor", "CatArray", "index"):
            if k in n:
                return "torch " + k
        return "torch " + n.split("<")[0][-40:]
    return n.split("(")[0][:60]


cnt = collections.Counter()
tim = collections.Counter()
streams = collections.Counter()
busy = []
for r in step:
    k = short(r["Kernel_Name"])
    cnt[k] += 1
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tim[k] += e - s
    streams[r.get("Stream_Id", r.get("Queue_Id", "?"))] += 1
    busy.append((s, e))
busy.sort()
covered, cur_s, cur_e = 0, busy[0][0], busy[0][1]
for s, e in busy[1:]:
    if s > cur_e:
        covered += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
covered += cur_e - cur_s
print("step: %d launches, wall %.2f ms, GPU busy (union over streams) %.2f ms, kernel sum %.2f ms" % (len(step), (t1 - t0) / 1e6, covered / 1e6, sum(tim.values()) / 1e6))
print("launches per stream / queue:", dict(streams))
print("%-64s %6s %9s %8s" % ("kernel", "n", "total us", "avg us"))
for k, n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%-64s %6d %9.1f %8.1f" % (k, n, tim[k] / 1e3, tim[k] / 1e3 / n))
