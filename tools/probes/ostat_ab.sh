A="--steps 20 --warmup 6 --no-nuscenes --no-batch1 --no-cpu-baseline --no-selfcheck"
for o in 1 0 1 0; do FTX_OSTAT=$o python bench.py $A > /dev/null 2> gpurun_out/ab4.err; echo "two streams ostat=$o: $(grep ms/step gpurun_out/ab4.err)"; done
for o in 1 0; do FTX_OSTAT=$o python bench.py $A --serial-branches > /dev/null 2> gpurun_out/ab4.err; echo "serial ostat=$o: $(grep ms/step gpurun_out/ab4.err)"; done
