# One-GPU cost of the gradient exchange over a one-rank RCCL communicator, every variant in its own process on the same box.
A="--steps 40 --warmup 10 --no-nuscenes --no-batch1 --no-cpu-baseline --no-selfcheck"
run() { echo "$1: $(env $2 python bench.py $A $3 2>gpurun_out/ab6.err | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])') ms/step; $(grep "reducer hooks" gpurun_out/ab6.err | cut -c20-110)"; }
run "no reducer" "X=1" ""
run "forced collectives: bucket i launched at the tail of bucket i+1 (shipped)" "X=1" "--force-collectives"
run "forced collectives: bucket i launched at its own tail (FTX_REDUCER_EAGER=1, round 3 until now)" "FTX_REDUCER_EAGER=1" "--force-collectives"
run "forced collectives: every bucket after the backward (FTX_REDUCER_LATE=1)" "FTX_REDUCER_LATE=1" "--force-collectives"
run "forced collectives (shipped), again" "X=1" "--force-collectives"
run "no reducer, again" "X=1" ""
