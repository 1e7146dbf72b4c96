"""HBM traffic of the sparse-conv kernels per training step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: python tools/pmc_hbm.py <fetch counter_collection.csv> <write counter_collection.csv> <steps in each run> > profiles/rNN_pmc_hbm_spconv.json
Collect (GPU box):  rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tune-gemm
Counter values are KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)."""
import collections, csv, json, sys

KERNELS = ("pairs_gemm_kernel", "spconv_reduce", "pairs_wgrad_kernel", "wgrad_reduce_kernel", "spconv_ostat_kernel")


def per_kernel(path, counter, steps):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                tot[k][0] += 1
                tot[k][1] += float(r["Counter_Value"]) * 1024.0
    return {k: (v[0] / steps, v[1] / steps) for k, v in tot.items()}


fetch, write, steps = per_kernel(sys.argv[1], "FETCH_SIZE", float(sys.argv[3])), per_kernel(sys.argv[2], "WRITE_SIZE", float(sys.argv[3])), float(sys.argv[3])
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tune-gemm; "
              "per training step (all %d steps of the run incl. warm-up averaged), summed over the sparse-conv kernels" % int(steps),
    "units": "counter values are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; the same factor "
             "was measured on this library's row gathers, 128-B and 512-B rows: tools/probes/fetch_calibration.py, profiles/r02_fetch_calibration.txt)",
    "fetch_raw_bytes_per_step": sum(v[1] for v in fetch.values()),
    "fetch_corrected_bytes_per_step": 2 * sum(v[1] for v in fetch.values()),
    "write_bytes_per_step": sum(v[1] for v in write.values()),
    "per_kernel": {k: {"launches_per_step": fetch[k][0], "fetch_raw_bytes": fetch[k][1], "write_bytes": write.get(k, (0, 0.0))[1]} for k in fetch},
    "workload": {"batch": 4, "shape": "kitti", "kind": "middle"},
}
print(json.dumps(out, indent=1))
