# A/B of one environment switch on ONE box: bench.py alternately without and with it, twice each.
# usage: bash tools/probes/ab_bench.sh VAR=VALUE [batch, default 4] [steps, default 30]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
B=${2:-4}; S=${3:-30}
for i in 1 2; do
  python bench.py --batch $B --steps $S --warmup 6 --no-cpu-baseline --no-nuscenes --no-batch1 > $O/default_b${B}_$i.json 2>/dev/null
  env "$1" python bench.py --batch $B --steps $S --warmup 6 --no-cpu-baseline --no-nuscenes --no-batch1 > $O/switch_b${B}_$i.json 2>/dev/null
done
echo "batch $B: default vs $1"; python tools/show_bench.py $O/default_b${B}_1.json $O/switch_b${B}_1.json $O/default_b${B}_2.json $O/switch_b${B}_2.json | grep frames
