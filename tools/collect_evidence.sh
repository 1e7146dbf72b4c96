#!/bin/bash
# Round evidence on the GPU box: tests, bench line, rocprofv3 kernel stats (two-stream and serial issue), HBM PMC passes, SQ counters and
# the per-layer micro-benchmark of the sparse-conv kernels, and the 2-rank data-parallel rehearsal of bench.py on the one GPU.
# usage (from the repo root, via gpurun): bash tools/collect_evidence.sh <out dir under gpurun_out> [steps: all | quick]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err && tail -c 600 $O/bench.json
python tools/bench_spconv.py > $O/spconv_layer_micro.txt 2>&1; tail -1 $O/spconv_layer_micro.txt
python tools/bench_bn.py > $O/bn_layer_micro.txt 2>&1; tail -1 $O/bn_layer_micro.txt
python tools/probes/eval_throughput.py > $O/eval_throughput.txt 2>&1; tail -2 $O/eval_throughput.txt
python tools/bench_attn.py > $O/attn_tilings.txt 2>&1; tail -3 $O/attn_tilings.txt
python tools/probes/loss_time.py > $O/loss_time.txt 2>&1; tail -2 $O/loss_time.txt
bash tools/probes/reducer_ab6.sh > $O/reducer_one_gpu_cost.txt 2>&1; cat $O/reducer_one_gpu_cost.txt
# N > 1 code path (bucketed overlapped all-reduce, per-rank batches) rehearsed as 2 ranks on this one GPU over gloo: NOT a scaling number
FTX_DIST_BACKEND=gloo FTX_FORCE_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_2ranks_one_gpu_gloo.json 2> $O/bench_2ranks.err; tail -c 300 $O/bench_2ranks_one_gpu_gloo.json
# the same path over the REAL backend: a one-rank RCCL communicator, bucketed async all-reduces issued from the autograd hooks inside every step
python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-batch1 --no-nuscenes --force-collectives > $O/bench_rccl_one_rank.json 2> $O/bench_rccl_one_rank.err; tail -c 400 $O/bench_rccl_one_rank.json
# self-launch: plain `python bench.py --gpus 2` with WORLD_SIZE unset becomes the launcher (here 2 ranks share the one GPU over gloo)
FTX_DIST_BACKEND=gloo FTX_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-selfcheck > $O/bench_self_launch_2ranks.json 2> $O/bench_self_launch.err; tail -c 300 $O/bench_self_launch_2ranks.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_two -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-nuscenes --no-attention-roofline --no-selfcheck > $O/prof_two_bench.json 2>/dev/null
cp $(find /tmp/ev_two -name "*kernel_stats.csv" | head -1) $O/rocprof_two_stream_stats.csv
# serial issue AND an eagerly executed trunk (FTX_VIT_GRAPHS=0: the same kernels, but no capture warm-up passes that would inflate the ViT
# kernels' per-step counts): every kernel's count and duration is its own
FTX_VIT_GRAPHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_ser -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-nuscenes --no-attention-roofline --no-selfcheck --serial-branches > $O/prof_serial_bench.json 2>/dev/null
cp $(find /tmp/ev_ser -name "*kernel_stats.csv" | head -1) $O/rocprof_serial_stats.csv
(echo "# rocprofv3 --kernel-trace --stats of: FTX_VIT_GRAPHS=0 bench.py --steps 5 --warmup 2 --serial-branches (7 steps; one stream, eager trunk)"; python3 $R/tools/prof_summary.py $O/rocprof_serial_stats.csv 7) > $O/rocprof_serial_summary.txt; head -22 $O/rocprof_serial_summary.txt
(echo "# rocprofv3 --kernel-trace --stats of: bench.py --steps 5 --warmup 2 (two streams, graphed trunk).  The graph capture replays the trunk 4 extra times, so the"; echo "# ViT rows (library GEMM, attention, layernorm, part of elementwise) are 11/7 of their per-step values here; the serial summary has the exact ones."; python3 $R/tools/prof_summary.py $O/rocprof_two_stream_stats.csv 7) > $O/rocprof_two_stream_summary.txt
# launches of ONE steady-state step (cut at the loss kernel), batch 1 and batch 4
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_k1 -- python3 $R/bench.py --batch 1 --steps 6 --warmup 3 --no-cpu-baseline --no-batch1 --no-nuscenes --no-selfcheck > /dev/null 2>&1
python3 $R/tools/step_kernels.py /tmp/ev_k1 > $O/step_kernels_batch1.txt; head -2 $O/step_kernels_batch1.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_k4 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-batch1 --no-nuscenes --no-selfcheck > /dev/null 2>&1
python3 $R/tools/step_kernels.py /tmp/ev_k4 > $O/step_kernels_batch4.txt; head -2 $O/step_kernels_batch4.txt
python3 $R/tools/underfilled.py /tmp/ev_k4 12 > $O/underfilled_batch4.txt; head -3 $O/underfilled_batch4.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ev_f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-nuscenes --no-selfcheck --no-attention-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ev_w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-nuscenes --no-selfcheck --no-attention-roofline > /dev/null 2>&1
python3 $R/tools/pmc_hbm.py $(find /tmp/ev_f -name "*counter_collection.csv" | head -1) $(find /tmp/ev_w -name "*counter_collection.csv" | head -1) 3 > $O/pmc_hbm_spconv.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/ev_sq -- python3 $R/tools/bench_spconv.py --iters 1 > /dev/null 2>&1
python3 $R/tools/pmc_sq.py /tmp/ev_sq pairs_ reduce > $O/pmc_spconv_sq.txt
ls -la $O
