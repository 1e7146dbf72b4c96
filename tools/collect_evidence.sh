#!/bin/bash
# Round evidence on the GPU box: tests, bench line, rocprofv3 kernel stats (two-stream and serial issue), HBM PMC passes.
# usage (from the repo root, via gpurun): bash tools/collect_evidence.sh <out dir under gpurun_out>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err && tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_two -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_two_bench.json 2>/dev/null
cp $(find /tmp/ev_two -name "*kernel_stats.csv" | head -1) $O/rocprof_two_stream_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_ser -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --serial-branches --no-tune-gemm > $O/prof_serial_bench.json 2>/dev/null
cp $(find /tmp/ev_ser -name "*kernel_stats.csv" | head -1) $O/rocprof_serial_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ev_f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tune-gemm > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ev_w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tune-gemm > /dev/null 2>&1
python3 $R/tools/pmc_hbm.py $(find /tmp/ev_f -name "*counter_collection.csv" | head -1) $(find /tmp/ev_w -name "*counter_collection.csv" | head -1) 3 > $O/pmc_hbm_spconv.json
ls -la $O
