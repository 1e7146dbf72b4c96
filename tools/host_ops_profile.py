"""Which ATen ops produce the small fill / copy kernels of a training step, and from where (torch.profiler with stacks).
usage: python tools/host_ops_profile.py [batch]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from fusiontransformer_amd.config import fusion_cfg
from fusiontransformer_amd.models.build import build_model
from fusiontransformer_amd.trainer import TrainStep
from torch.profiler import profile, ProfilerActivity

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = fusion_cfg("middle")
torch.manual_seed(0)
model, m2d, m3d = build_model(cfg)
model = model.cuda().train()
step = TrainStep(cfg, model, metrics=(m2d, m3d))
datas = [build_inputs(cfg, batch, "kitti", 0, torch.device("cuda"), cycle=c)[1] for c in range(2)]
for i in range(4):
    step(datas[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    for i in range(2):
        step(datas[i % 2])
    torch.cuda.synchronize()
counts = collections.Counter()
stacks = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::copy_", "aten::contiguous", "aten::clone", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::empty", "aten::zeros_like", "aten::_to_copy"):
        counts[ev.name] += 1
        st = [s for s in (ev.stack or []) if "fusiontransformer_amd" in s or "bench.py" in s or "trainer" in s]
        stacks[ev.name][st[0] if st else (ev.stack[0] if ev.stack else "?")] += 1
for name, c in counts.most_common():
    print("%-18s %6.1f per step" % (name, c / 2))
    for st, k in stacks[name].most_common(8):
        print("      %5.1f  %s" % (k / 2, st[:150]))
