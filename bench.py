#!/usr/bin/env python
"""Throughput bench of the fusion forward+backward(+Adam) step on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver
launches it under torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], per-GPU batch of configs[3]): synthetic
SemanticKITTI-shaped frames (370x1226 image, ~20k front-camera points each),
MiddleFusionTransformer with the full DeiT-base-distilled-384 trunk and SPVCNN, fp32,
forward + loss (CE x2 + 0.1*KL x2) + backward + Adam, inputs resident in HBM.  `--batch`
frames per GPU per step (default 4 = configs[3]'s 32 global / 8 GPUs), weak scaling.

A "step" is one pass of the hot path over one batch.  `value` = frames of all ranks per
second, timed over exactly K steps between barrier + synchronize pairs, max over ranks."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver (set before the runtime starts)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # exact-fp32 MFMA peak (same guide)


def build_inputs(cfg, batch, shape, rank, device, cycle=0):
    """One resident batch.  `cycle` selects a different set of frames (seeds), so the timed loop can alternate between
    DISTINCT batches: every per-batch structure (voxel hash, kernel maps, sorted segments) is rebuilt in every step."""
    from fusiontransformer_amd.data.synth import make_batch
    from fusiontransformer_amd.models.image_models_billinear import pack_img_indices
    from fusiontransformer_amd.sparse import SparseTensor
    b = make_batch([1000 * cycle + rank * batch + i for i in range(batch)], shape=shape)
    data = {
        "img": torch.from_numpy(b["img"]).to(device),
        "img_indices": pack_img_indices(b["img_indices"], device),
        "lidar": SparseTensor(torch.from_numpy(b["feats"]).to(device), torch.from_numpy(b["coords"]).int().to(device)),
        "seg_label": torch.from_numpy(b["seg_label"]).to(device),
    }
    return b, data


def spconv_roofline(log, workload=None):
    """Algorithmic bytes / HIP-event time over the sparse-conv launches of one timed step (the dominant hand-written kernel group),
    plus the same accounting for the other hand-written groups (BatchNorm: HBM, attention: MFMA) under "other_kernel_groups".

    Unit = one (in,out) pair: 4*(ca+co) bytes (gather ca floats + write co floats), plus
    the kernel weights 4*kvol*ca*co read once per convolution.  A convolution is two launches
    (pair gather-GEMM, ordered reduce), or one (weight gradient; bijective maps, whose GEMM epilogue writes the output
    itself); the bytes are attributed to the GEMM / wgrad launch, the reduce launch adds time only."""
    tot_bytes = tot_ms = tot_flops = tot_roof_ms = 0.0
    per_kind = {}
    bn = {"launches": 0, "ms": 0.0, "bytes": 0.0}
    attn = {"launches": 0, "ms": 0.0, "flops": 0.0}
    n_sp = 0
    for kind, e0, e1, m in log:
        ms = e0.elapsed_time(e1)
        if kind.startswith("bn_"):
            bn["launches"] += 1
            bn["ms"] += ms
            bn["bytes"] += 4.0 * m["n"] * m["c"] * (m["reads"] + m["writes"])
            continue
        if kind.startswith("attn_"):
            attn["launches"] += 1
            attn["ms"] += ms
            attn["flops"] += m["products"] * 2.0 * m["b"] * m["h"] * m["t"] * m["t"] * m["d"]
            continue
        n_sp += 1
        nbytes = flops = 0.0
        if kind != "spconv_reduce":
            nbytes = 4.0 * m["pairs"] * (m["ca"] + m["co"]) + 4.0 * m["kvol"] * m["ca"] * m["co"]
            flops = 2.0 * m["pairs"] * m["ca"] * m["co"]
        tot_bytes += nbytes
        tot_ms += ms
        tot_flops += flops
        # the roof that binds THIS launch: its algorithmic bytes at the HBM peak or its flops at the exact-fp32 MFMA peak, whichever is longer
        tot_roof_ms += 1e3 * max(nbytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_F32_PEAK_TFLOPS * 1e12))
        k = per_kind.setdefault(kind, [0, 0.0, 0.0, 0.0])
        k[0] += 1
        k[1] += ms
        k[2] += nbytes
        k[3] += flops
    if tot_ms <= 0:
        return None
    achieved = tot_bytes / (tot_ms * 1e-3) / 1e9
    n_conv = sum(v[0] for k, v in per_kind.items() if k != "spconv_reduce")
    traffic, traffic_note = None, None
    for name in ("r03_pmc_hbm_spconv.json", "r02_pmc_hbm_spconv.json", "r01_pmc_hbm_spconv.json"):
        try:  # HBM bytes of these kernels from the PMC passes committed under profiles/ (same workload)
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if pmc["workload"] == workload:
                traffic = int(pmc["fetch_corrected_bytes_per_step"] + pmc["write_bytes_per_step"])
                traffic_note = ("bytes per step over the same kernels: FETCH_SIZE x2 (gfx950 correction; checked on this library's row gathers: tools/probes/fetch_calibration.py) + WRITE_SIZE, "
                                "rocprofv3 --pmc in separate passes, profiles/" + name)
                break
        except (OSError, KeyError, ValueError):
            pass
    other = {}
    if bn["ms"] > 0:
        gbs = bn["bytes"] / (bn["ms"] * 1e-3) / 1e9
        other["batchnorm (bn_partial / bn_apply, fwd + bwd)"] = {
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "launches": bn["launches"], "ms": round(bn["ms"], 3), "algorithmic_bytes_per_step": int(bn["bytes"]),
            "bytes_rule": "4*N*C per row matrix read or written: forward x (+residual) in, y out -- x twice when the statistics are not produced by the "
                          "convolution's reduce pass; backward two passes over (gy, x) -- plus y for the ReLU mask only where a residual went into it, otherwise the mask is recomputed from x --, gx (+ residual gradient) out"}
    if attn["ms"] > 0:
        tf = attn["flops"] / (attn["ms"] * 1e-3) / 1e12
        other["attention (attn_fwd / attn_bwd_kv / attn_bwd_q)"] = {
            "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
            "launches": attn["launches"], "ms": round(attn["ms"], 3),
            "flops_rule": "2*B*H*T^2*64 per product; 2 products forward (S, O), 7 backward (S, dP, dV, dK; S^T, dP^T, dQ)"}
    return {
        "bound": "hbm", "kernel": "pairs_gemm_kernel + spconv_reduce(_stats)_kernel + pairs_wgrad_kernel + wgrad_reduce_kernel (sparse conv fwd / dgrad / wgrad)",
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        "binding_frac": round(tot_roof_ms / tot_ms, 4),
        "binding_frac_rule": "sum over the convolutions of max(algorithmic bytes / HBM peak, 2*pairs*ca*co flop / exact-fp32 MFMA peak) / sum of the HIP-event "
                             "time of all sparse-conv launches (reduce passes add time only): the fraction of each launch's own binding roof",
        "traffic": traffic, "traffic_from_committed_profile": traffic is not None, "traffic_note": traffic_note,
        "launches": n_sp, "convolutions": n_conv, "avg_conv_us": round(1e3 * tot_ms / max(n_conv, 1), 2),
        "algorithmic_bytes_per_step": int(tot_bytes), "useful_tflops": round(tot_flops / (tot_ms * 1e-3) / 1e12, 3),
        "mfma_f32_peak_tflops": MFMA_F32_PEAK_TFLOPS,
        "measured_on": "the last timed step, which issues both branches on one stream with a HIP event pair around every logged launch "
                       "(so a kernel's duration is its own); that step is part of `value`",
        "per_kernel": {k: {"launches": v[0], "ms": round(v[1], 3), "avg_us": round(1e3 * v[1] / v[0], 1),
                           "GB/s": round(v[2] / (v[1] * 1e-3) / 1e9, 1), "useful_TFLOP/s": round(v[3] / (v[1] * 1e-3) / 1e12, 2)}
                       for k, v in per_kind.items()},
        "other_kernel_groups": other,
    }


def host_cores():
    """CPU threads this process may really use: affinity mask and cgroup quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(cfg, np_batch):
    """The CPU restatement of the reference (oracle/), forward+backward on ONE frame of the same
    synthetic workload: 1 untimed warm-up + timed passes until ~12 s of CPU work."""
    from oracle import ft_oracle as O
    from fusiontransformer_amd.data.synth import make_batch
    nthreads = host_cores()
    torch.set_num_threads(nthreads)
    torch.manual_seed(0)
    model = O.build_model(dict(cfg.MODEL))
    model.train()
    b = make_batch([0])
    cw = torch.tensor(cfg.TRAIN.CLASS_WEIGHTS)
    lab = torch.from_numpy(b["seg_label"])

    def one():
        model.zero_grad()
        inp = {"img": torch.from_numpy(b["img"]), "img_indices": b["img_indices"], "lidar": O.SparseTensor(torch.from_numpy(b["feats"]), b["coords"])}
        out = model(inp)
        l2, l3 = O.fusion_losses(out, lab, cw, float(cfg.TRAIN.FusionTransformer.lambda_xm), True)
        (l2 + l3).backward()

    log("cpu_baseline: oracle warm-up pass on %d threads" % nthreads)
    one()
    times = []
    while sum(times) < 12.0 and len(times) < 16:   # ~12 s of CPU work
        t = time.perf_counter()
        one()
        times.append(time.perf_counter() - t)
        log("cpu_baseline: pass %.1f s" % times[-1])
    return {"value": round(1.0 / float(np.median(times)), 4), "unit": "frames/s", "cores": nthreads, "cpu_model": cpu_model(), "kind": "port",
            "sample": "1 synthetic SemanticKITTI frame (%d points), fwd+bwd, batch 1, 1 warm-up + %d timed passes (median), torch CPU fp32 (CPU restatement of the reference, oracle/ft_oracle.py)" % (b["coords"].shape[0], len(times))}


def launch_command(gpus, argv, port=None):
    """The command `python bench.py --gpus N` turns itself into when it was NOT started by a launcher: one fresh process per GPU
    under torch.distributed.run (the driver's own N > 1 command line; the reference's is `torchpack dist-run -np N`, torchpack_run.sh:3)."""
    if port is None:
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus, "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    """Parent of a self-launched N > 1 run: it never touches the GPU (importing torch does not initialise HIP), starts the ranks as
    CHILD processes -- no exec of this process -- and exits with their code; rank 0's JSON line reaches stdout through inheritance."""
    import subprocess
    cmd = launch_command(gpus, argv)
    log("WORLD_SIZE is not set: starting %d ranks: %s" % (gpus, " ".join(cmd)))
    return subprocess.run(cmd).returncode


def selfcheck(cfg, model, step, data, device, bf16, compare_grads):
    """Evidence that the timed configuration computes the right thing: one more step on `data` exactly as the timed steps ran it
    (HIP-graph trunk, two streams), then the SAME step from the same pre-step parameters on a twin model with the eager trunk
    and both branches issued serially on one stream.  Deterministic kernels => the two must agree to the last bit; a kernel
    skipped or fed a stale buffer inside a replayed graph shows up here instead of as a better number."""
    from fusiontransformer_amd.models.build import build_model
    from fusiontransformer_amd.trainer import TrainStep
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    lb = getattr(model, "lidar_backbone", None)
    torch.manual_seed(4321)      # dropout
    preds = step(data)
    torch.cuda.synchronize()
    logits = {k: v.detach().clone() for k, v in preds.items()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    losses = (float(step.last["loss_2d"].item()), float(step.last["loss_3d"].item()))
    twin, _, _ = build_model(cfg)
    twin.load_state_dict(state)
    twin = twin.to(device).train()
    twin.image_backbone.backbone.use_graphs = False      # eager trunk, same segment structure
    twin.overlap_branches = False
    if bf16:
        twin.image_backbone.backbone.set_bf16(True)
    step_t = TrainStep(cfg, twin, loss_mix=step.loss_mix)
    torch.manual_seed(4321)
    preds_t = step_t(data)
    torch.cuda.synchronize()
    dlogit = max(float((logits[k] - preds_t[k].detach()).abs().max().item()) for k in logits)
    out = {"what": "step K+1 as timed (graphed trunk, 2 streams) against an eager-trunk, one-stream twin from the same pre-step parameters",
           "max_abs_dlogit": dlogit, "loss_2d": losses[0], "loss_3d": losses[1],
           "twin_loss_2d": float(step_t.last["loss_2d"].item()), "twin_loss_3d": float(step_t.last["loss_3d"].item()),
           "finite": bool(all(torch.isfinite(v).all().item() for v in logits.values()))}
    if compare_grads:
        worst, worst_name, n_cmp = 0.0, None, 0
        for n, p in twin.named_parameters():
            if p.grad is None:
                continue
            g = grads.get(n)
            if g is None:
                worst, worst_name = float("inf"), n
                continue
            den = float(p.grad.abs().max().item())
            d = float((g - p.grad).abs().max().item())
            d = d / den if den > 0 else d
            if d > worst:
                worst, worst_name = d, n
            n_cmp += 1
        out["max_rel_dgrad"] = worst
        out["max_rel_dgrad_parameter"] = worst_name
        out["gradients_compared"] = n_cmp
    else:
        out["max_rel_dgrad"] = None
        out["gradients_compared"] = "no: p.grad holds the average over ranks"
    out["bit_identical"] = bool(dlogit == 0.0 and (not compare_grads or out["max_rel_dgrad"] == 0.0))
    del twin, step_t
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU per step")
    ap.add_argument("--cycle", type=int, default=2, help="distinct resident batches the steps alternate between (>= 2: nothing per-batch can be cached across steps)")
    ap.add_argument("--shape", default="kitti", choices=["kitti", "nuscenes"])
    ap.add_argument("--kind", default="middle", choices=["middle", "early", "late"])
    ap.add_argument("--attn", default=os.environ.get("FTX_ATTN", "ftx"), choices=["ftx", "torch"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16-forward", action="store_true",
                    help="BASELINE configs[4]: ViT GEMMs with bf16 operands (fp32 accumulate); NOT the headline configuration, the line says so in dtype")
    ap.add_argument("--no-nuscenes", action="store_true", help="skip the secondary measurement on NuScenes-shaped frames (BASELINE configs[2])")
    ap.add_argument("--no-batch1", action="store_true", help="skip the secondary measurement at batch 1 (the literal BASELINE configs[1] frame)")
    ap.add_argument("--serial-branches", action="store_true",
                    help="issue the image and LiDAR branches back to back on one stream in every step (profiling aid: under rocprofv3 "
                         "each kernel's duration is then its own, as in the roofline block's HIP-event timings)")
    ap.add_argument("--no-attention-roofline", action="store_true",
                    help="skip the standalone timing of the attention kernels after the timed region (profiling runs: keeps their launch counts per step exact)")
    ap.add_argument("--no-index-prefetch", dest="index_prefetch", action="store_false",
                    help="build the coordinate structures of every batch inside its own forward instead of starting the build of batch i+1 "
                         "during step i (TrainStep(batch, next_batch): non-blocking start + bounded polling, trainer.py)")
    ap.add_argument("--no-tune-gemm", dest="tune_gemm", action="store_false", help="skip TunableOp selection of the library GEMM kernels")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: create a one-rank RCCL communicator and run the gradient exchange of the N > 1 path (bucketed async all-reduces "
                         "issued from the autograd hooks) inside every step; the result is the identity, the code path is the multi-GPU one")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the graphed-vs-eager bit-identity check after the timed region")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (the reference's counterpart is torchpack_run.sh:3)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON record: RCCL prints a version banner to stdout when its first communicator comes up, so
    # file descriptor 1 points at stderr for the rest of the run and the record is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from fusiontransformer_amd import functional as spf
    from fusiontransformer_amd.config import fusion_cfg
    from fusiontransformer_amd.data.synth import SHAPES
    from fusiontransformer_amd.dist import GradReducer, init_process_group
    from fusiontransformer_amd.models.build import build_model
    from fusiontransformer_amd.trainer import TrainStep

    # FTX_DIST_BACKEND=gloo FTX_FORCE_DEVICE=0 rehearses the N>1 path with several ranks on ONE GPU
    force_coll = args.force_collectives or os.environ.get("FTX_FORCE_COLLECTIVES") == "1"
    rank, world, local_rank = init_process_group(os.environ.get("FTX_DIST_BACKEND"), force=force_coll)
    if os.environ.get("FTX_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["FTX_FORCE_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.tune_gemm:
        # The ViT's dense GEMMs are plain library calls (hipBLASLt / rocBLAS through torch); TunableOp keeps the fastest
        # solution per shape.  The results are read from the file committed next to libftx.so, so a job tunes once
        # (fusiontransformer_amd/gemm_tuning.py); shapes missing from it are tuned during the warm-up steps, before the timed region.
        from fusiontransformer_amd import gemm_tuning
        gemm_tuning.enable(rank)

    cfg = fusion_cfg(args.kind)
    cfg.MODEL.attn_impl = args.attn
    cfg.MODEL.lift_size = (SHAPES[args.shape]["H"], SHAPES[args.shape]["W"])
    torch.manual_seed(0)
    model, m2d, m3d = build_model(cfg)
    model = model.to(device).train()
    if args.bf16_forward:
        model.image_backbone.backbone.set_bf16(True)
    reducer = GradReducer(model, bucket_mb=float(os.environ.get("FTX_BUCKET_MB", "128")), force_collectives=force_coll) if (world > 1 or force_coll) else None
    if os.environ.get("FTX_NO_REDUCER") == "1" and world == 1:
        reducer = None      # measurement aid: the process group exists, the step runs without the reducer
    step = TrainStep(cfg, model, metrics=(m2d, m3d), grad_reducer=reducer)
    batches = [build_inputs(cfg, args.batch, args.shape, rank, device, cycle=c) for c in range(max(1, args.cycle))]
    np_batch = batches[0][0]
    datas = [d for _, d in batches]
    points = [int(b["coords"].shape[0]) for b, _ in batches]
    n_points = points[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log("model and %d resident batches (%s points) resident; warm-up" % (len(datas), points))
    # index prefetch (default): while step i runs, the index build of batch i+1 is started on a third stream (TrainStep(batch, next_batch)).
    # Every step then builds exactly one set -- the one the NEXT step consumes -- so the K timed steps contain K index builds either way.
    nxt = (lambda seq, i: seq[(i + 1) % len(seq)]) if args.index_prefetch else (lambda seq, i: None)
    for i in range(args.warmup):
        step(datas[i % len(datas)], nxt(datas, i))
    barrier()
    if rank == 0:
        log("timing %d steps" % args.steps)
    model.overlap_branches = not args.serial_branches
    t0 = time.perf_counter()
    launch_log = None
    loss_first = loss_last = None
    for i in range(args.steps):
        if rank == 0 and i == args.steps - 1:
            # HIP events around every sparse-conv launch of the last timed step.  That one step issues
            # the two branches back to back instead of on two streams, so the per-kernel durations are
            # the kernels' own and not inflated by the ViT GEMMs running beside them.
            launch_log = spf.LAUNCH_LOG = []
            model.overlap_branches = False
        j = args.warmup + i            # continue the warm-up's alternation: the batch of step j was prepared by step j - 1
        step(datas[j % len(datas)], nxt(datas, j))
        if i == 0:
            loss_first = step.last          # device tensors: read after the timed region
        loss_last = step.last
    spf.LAUNCH_LOG = None
    model.overlap_branches = not args.serial_branches
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    allreduce_ms = None
    if reducer is not None and rank == 0:
        hs = reducer.hook_stats
        log("reducer hooks: %s; per step %.3f ms of host time on the autograd thread" % (hs, hs["host_ms"] / max(reducer.step_idx, 1)))
    if reducer is not None:
        allreduce_ms = reducer.allreduce_ms(3)    # collective: every rank takes part; gradients are overwritten (next step zeroes them)
        barrier()
    check = None
    if not args.no_selfcheck:
        # every rank runs it (the step contains collectives when N > 1); rank 0 reports
        j = args.warmup + args.steps
        check = selfcheck(cfg, model, step, datas[j % len(datas)], device, args.bf16_forward, compare_grads=(world == 1))
        barrier()

    if rank == 0:
        log("%.1f ms/step" % (1e3 * elapsed / args.steps))
        frames = args.batch * world * args.steps
        roof = spconv_roofline(launch_log, {"batch": args.batch, "shape": args.shape, "kind": args.kind}) if launch_log else None
        out = {
            "metric": "frames/sec fwd+bwd (SemanticKITTI synth), whole job",
            "value": round(frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 ViT GEMM operands, f32 accumulate and everything else (configs[4] mode, not the reference precision)" if args.bf16_forward else "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] frame shape at configs[3] per-GPU batch: %s-shaped synthetic frames, %dx%d image, "
                                   "%d points/batch, %sFusionTransformer (DeiT-B/16-384 distilled + SPVCNN), fwd+loss+bwd+Adam, fp32, random-init weights"
                                   % (args.shape, SHAPES[args.shape]["H"], SHAPES[args.shape]["W"], n_points, args.kind.capitalize()),
                       "frames_per_gpu": args.batch, "global_batch": args.batch * world, "points_per_gpu_batch": n_points,
                       "resident_batches_cycled": len(datas), "points_of_each_resident_batch": points,
                       "attention": args.attn, "library_gemm_tuning": bool(args.tune_gemm), "branch_overlap": "off (--serial-branches)" if args.serial_branches else "2 HIP streams (image / LiDAR)",
                       "index_prefetch": ("index build of batch i+1 started during step i on a third stream without blocking the host, finished by its forward; one build per timed step" if args.index_prefetch
                                          else "off: every batch's coordinate structures are built inside its own forward"),
                       "parallelism": "dp%d" % world},
            "frames_per_sec_per_gpu": round(frames / elapsed / world, 3),
            "roofline": roof,
        }
        trunk = getattr(getattr(model, "image_backbone", None), "backbone", None)
        out["selfcheck"] = check
        out["allreduce_ms_per_step_standalone"] = None if allreduce_ms is None else round(allreduce_ms, 3)
        out["trunk_graphs"] = trunk.graph_state() if hasattr(trunk, "graph_state") else "n/a"
        if loss_first is not None:
            out["losses"] = {"first_timed_step": {k: round(float(v.item()), 6) for k, v in loss_first.items()},
                             "last_timed_step": {k: round(float(v.item()), 6) for k, v in loss_last.items()},
                             "note": "rank 0; additive mix CE + 0.1 KL (SemanticTrainer.py:158-178); random-init weights, Adam lr 1e-4"}
        import torch.distributed as _d
        out["rccl_ranks"] = (world if (_d.is_initialized() and _d.get_backend() == "nccl") else 0)
        out["multi_gpu"] = ("measured: %d ranks over RCCL" % world) if (world > 1 and out["rccl_ranks"] > 1) else "unmeasured in this run (one GPU)"
        out["collectives"] = ("none (single process, no process group)" if reducer is None else
                              "%s backend, %d rank(s), %d flat buckets of <= %d MB, async all-reduce on an exchange stream, each bucket issued from the autograd hook of the NEXT bucket's last gradient (the last one after the backward)%s"
                              % (_d.get_backend(), world, len(reducer.buckets), reducer.bucket_bytes >> 20, " (forced at world size 1)" if world == 1 else ""))
        if roof is not None and args.attn == "ftx" and not args.no_attention_roofline:
            # The ViT trunk replays as HIP graphs, whose kernels cannot be bracketed by events from the host; the attention kernels are
            # timed here, standalone, at the workload's shape (same launches as inside the graphs), after the timed region.
            # The library entry points are called directly on preallocated buffers, 20 launches between two events, so the figure is
            # kernel time (through the autograd node the host's ~30 us per call would be what is measured at this size).
            L = spf._lib.load()
            B_, T_, H_ = args.batch, 578, 12
            qkv = torch.randn(B_, T_, 3, H_, 64, device=device)
            go = torch.randn(B_, T_, H_ * 64, device=device)
            o = torch.empty(B_, T_, H_ * 64, device=device)
            lse = torch.empty(B_, H_, T_, device=device)
            gq = torch.empty_like(qkv)
            ws_bytes = int(L.ftx_attn_bwd_workspace_bytes(B_, T_, H_))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            st = spf.stream()
            fwd = lambda: spf._lib.check(L.ftx_attn_fwd(qkv.data_ptr(), B_, T_, H_, 64, 0.125, o.data_ptr(), lse.data_ptr(), st), "ftx_attn_fwd")
            bwd = lambda: spf._lib.check(L.ftx_attn_bwd(qkv.data_ptr(), o.data_ptr(), go.data_ptr(), lse.data_ptr(), B_, T_, H_, 64, 0.125, gq.data_ptr(),
                                                   ws.data_ptr(), ws_bytes, st), "ftx_attn_bwd")
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            fwd(); bwd()
            torch.cuda.synchronize()
            reps = 20
            ev[0].record()
            for _ in range(reps):
                fwd()
            ev[1].record()
            for _ in range(reps):
                bwd()
            ev[2].record()
            torch.cuda.synchronize()
            t_f, t_b = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
            prod = 2.0 * args.batch * 12 * 578 * 578 * 64
            tf_f, tf_b = 2 * prod / (t_f / reps * 1e-3) / 1e12, 7 * prod / (t_b / reps * 1e-3) / 1e12
            tf_all = 9 * prod / ((t_f + t_b) / reps * 1e-3) / 1e12
            roof["other_kernel_groups"]["attention (attn_fwd / attn_bwd_kv / attn_bwd_q)"] = {
                "bound": "mfma", "achieved": round(tf_all, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf_all / MFMA_F32_PEAK_TFLOPS, 4),
                "forward_TFLOP/s": round(tf_f, 2), "backward_TFLOP/s": round(tf_b, 2), "us_per_block_fwd": round(1e3 * t_f / reps, 1),
                "us_per_block_bwd": round(1e3 * t_b / reps, 1), "launches_per_step": 3 * 12,
                "flops_rule": "2*B*H*T^2*64 per product; 2 products forward (S, O), 7 backward (S, dP, dV, dK; S^T, dP^T, dQ); B=%d, H=12, T=578" % args.batch,
                "measured_on": "standalone launches of the library entry points at the workload's shape after the timed region, 20 back to back "
                               "between two events (inside the step the same kernels replay from HIP graphs)"}
            del qkv, go, o, lse, gq, ws
        if world == 1 and args.batch != 1 and not args.no_batch1:
            # secondary figure, outside the timed region above: the literal BASELINE configs[1] workload (ONE frame per step)
            ones = [build_inputs(cfg, 1, args.shape, rank, device, cycle=c)[1] for c in range(max(1, args.cycle))]
            one = ones[0]
            step.reset_prefetch()      # the self-pausing index prefetch learnt its setting on the batches above
            for i in range(4):
                step(ones[i % len(ones)], nxt(ones, i))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(4, 14):
                step(ones[i % len(ones)], nxt(ones, i))
            torch.cuda.synchronize()
            ms1 = 1e3 * (time.perf_counter() - t1) / 10
            out["config"]["batch1_configs1_literal"] = {"frames_per_sec": round(1e3 / ms1, 2), "ms_per_step": round(ms1, 3), "steps": 10,
                                                       "points": int(one["lidar"].F.shape[0])}
        if world == 1 and args.shape != "nuscenes" and not args.no_nuscenes and not args.bf16_forward:
            # secondary figure, outside the timed region: BASELINE configs[2], NuScenes-shaped frames (900x1600 image, ~27 k points
            # per frame) at the same per-GPU batch, through a second model built for that lift size
            del step, model
            torch.cuda.empty_cache()
            cfg_n = fusion_cfg(args.kind)
            cfg_n.MODEL.attn_impl = args.attn
            cfg_n.MODEL.lift_size = (SHAPES["nuscenes"]["H"], SHAPES["nuscenes"]["W"])
            torch.manual_seed(0)
            model_n, m2n, m3n = build_model(cfg_n)
            model_n = model_n.to(device).train()
            step_n = TrainStep(cfg_n, model_n, metrics=(m2n, m3n))
            nbs = [build_inputs(cfg_n, args.batch, "nuscenes", rank, device, cycle=c) for c in range(max(1, args.cycle))]
            nb = nbs[0][0]
            nds = [d for _, d in nbs]
            for i in range(4):
                step_n(nds[i % len(nds)], nxt(nds, i))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(4, 14):
                step_n(nds[i % len(nds)], nxt(nds, i))
            torch.cuda.synchronize()
            msn = 1e3 * (time.perf_counter() - t1) / 10
            out["config"]["nuscenes_shaped_configs2"] = {"frames_per_sec": round(args.batch * 1e3 / msn, 2), "ms_per_step": round(msn, 3), "steps": 10,
                                                        "frames_per_gpu": args.batch, "points": int(nb["coords"].shape[0]),
                                                        "image": "%dx%d" % (SHAPES["nuscenes"]["H"], SHAPES["nuscenes"]["W"])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, np_batch)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if args.tune_gemm and rank == 0:
        from fusiontransformer_amd import gemm_tuning
        if gemm_tuning.save(rank):
            log("library-GEMM selections written to %s" % gemm_tuning.SHARED)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
