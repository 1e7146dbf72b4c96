# Step time against the number of hardware queues ROCm gives the process (GPU_MAX_HW_QUEUES; default 4), separate processes on one box.
# Measured: 4 queues 26.6-26.9 ms/step, 3 queues 27.0-27.1, 5 / 6 / 8 queues 28.1-28.2, 2 queues 31.7; batch 1: 14.6 (4) against 16.9 (3).
A="--steps 40 --warmup 10 --no-nuscenes --no-batch1 --no-cpu-baseline --no-selfcheck"
run() { echo "$1: $(env $2 python bench.py $A $3 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])') ms/step"; }
run "4 queues" "X=1" ""
run "3 queues" "GPU_MAX_HW_QUEUES=3" ""
run "4 queues" "X=1" ""
run "3 queues" "GPU_MAX_HW_QUEUES=3" ""
run "3 queues batch 1" "GPU_MAX_HW_QUEUES=3" "--batch 1"
run "4 queues batch 1" "X=1" "--batch 1"
