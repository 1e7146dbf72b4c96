set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stepk; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/sk1 -- python3 $R/bench.py --batch 1 --steps 6 --warmup 3 --no-cpu-baseline --no-batch1 --no-nuscenes --no-selfcheck > $O/b1.json 2>/dev/null &&
python3 $R/tools/step_kernels.py /tmp/sk1 > $O/step_kernels_batch1.txt &&
rocprofv3 --kernel-trace --output-format csv -d /tmp/sk4 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-batch1 --no-nuscenes --no-selfcheck > $O/b4.json 2>/dev/null &&
python3 $R/tools/step_kernels.py /tmp/sk4 > $O/step_kernels_batch4.txt && head -5 $O/step_kernels_batch1.txt $O/step_kernels_batch4.txt
