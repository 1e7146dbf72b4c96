# A/B of the gradient-exchange machinery on ONE GPU (one-rank RCCL communicator): what the N > 1 path costs before any byte moves.
A="--steps 20 --warmup 6 --no-nuscenes --no-batch1 --no-cpu-baseline --no-selfcheck"
python bench.py $A --force-collectives > gpurun_out/ab_forced.json 2> gpurun_out/ab_forced.err; echo "forced collectives: $(grep ms/step gpurun_out/ab_forced.err)"
FTX_REDUCER_LATE=1 python bench.py $A --force-collectives > gpurun_out/ab_late.json 2> gpurun_out/ab_late.err; echo "forced, every bucket launched after the backward: $(grep ms/step gpurun_out/ab_late.err)"
python bench.py $A > gpurun_out/ab_none.json 2> gpurun_out/ab_none.err; echo "no reducer: $(grep ms/step gpurun_out/ab_none.err)"
