A="--batch 1 --steps 30 --warmup 8 --no-nuscenes --no-batch1 --no-cpu-baseline --no-selfcheck"
for o in 1 all 0 1 all 0; do FTX_OSTAT=$o python bench.py $A > /dev/null 2> gpurun_out/ab5.err; echo "batch 1 ostat=$o: $(grep ms/step gpurun_out/ab5.err)"; done
