"""Which HIP stream runs on which hardware queue: (Stream_Id, Queue_Id) pairs with their kernel counts and a few kernel names, from a
rocprofv3 --kernel-trace csv.  usage: python tools/probes/queue_map.py DIR"""
import collections, csv, glob, os, sys
kt = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
acc = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    acc[(r.get("Stream_Id"), r.get("Queue_Id"))][n] += 1
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    print("stream %s queue %s: %d launches; %s" % (k[0], k[1], sum(c.values()), ", ".join("%s x%d" % kv for kv in c.most_common(4))))
