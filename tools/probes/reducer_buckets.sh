# One-GPU cost of the gradient exchange (one-rank RCCL) against the bucket size (bench.py reads FTX_BUCKET_MB; GradReducer(bucket_mb=...)).
# Measured: no reducer 27.2-27.7 ms/step; 2 x 256 MB 27.1; 4 x 128 MB (default) 27.7; 7 x 64 MB 28.7; 14 x 32 MB 29.5 -- every launch costs
# host time on the autograd thread and one more set of stream waits; fewer buckets overlap less of a real all-reduce.
A="--steps 40 --warmup 10 --no-nuscenes --no-batch1 --no-cpu-baseline --no-selfcheck"
run() { echo "$1: $(env $2 python bench.py $A $3 2>gpurun_out/ab10.err | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])') ms/step; $(grep "reducer hooks" gpurun_out/ab10.err | cut -c20-100)"; }
run "no reducer" "X=1" ""
run "4 buckets of 128 MB" "FTX_BUCKET_MB=128" "--force-collectives"
run "7 buckets of 64 MB" "FTX_BUCKET_MB=64" "--force-collectives"
run "14 buckets of 32 MB" "FTX_BUCKET_MB=32" "--force-collectives"
run "2 buckets of 256 MB" "FTX_BUCKET_MB=256" "--force-collectives"
run "no reducer" "X=1" ""
