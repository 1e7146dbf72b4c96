# kernel census of the step with the gradient-exchange machinery on (one-rank RCCL): which kernels does it add, and how long do they run?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt_on -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-nuscenes --no-attention-roofline --no-selfcheck --force-collectives > /dev/null 2>&1
cp $(find /tmp/rt_on -name "*kernel_stats.csv" | head -1) $O/reducer_on_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt_off -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-nuscenes --no-attention-roofline --no-selfcheck > /dev/null 2>&1
cp $(find /tmp/rt_off -name "*kernel_stats.csv" | head -1) $O/reducer_off_stats.csv
