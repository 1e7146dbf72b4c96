"""Full-size parity (BASELINE.json configs[1] and configs[2] shapes): the 12-block trunk, a whole ~20 k-point
SemanticKITTI-shaped frame and a whole 900x1600 / ~27 k-point NuScenes-shaped frame against the CPU oracle, plus
train-mode steps over DIFFERENT batches through one model (nothing about a batch may survive on the module).

Bars: per-point logits within 1e-3 (BASELINE.json north_star); voxel coordinates of every level, neighbour tables and
pair lists bit-exact."""
import numpy as np
import pytest
import torch

from tests.helpers import oracle_inputs, product_inputs, small_cfg

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _full_pair(kind="middle", lift_size=None, seed=0):
    from fusiontransformer_amd.config import fusion_cfg
    from fusiontransformer_amd.models.build import build_model
    from oracle import ft_oracle as O
    cfg = fusion_cfg(kind)              # depth 12, taps 5 / 11: the reference's middlefusion.yaml
    if lift_size is not None:
        cfg.MODEL.lift_size = lift_size
    torch.manual_seed(seed)
    oracle = O.build_model(dict(cfg.MODEL))
    model, _, _ = build_model(cfg)
    model.load_state_dict(oracle.state_dict())
    return cfg, oracle.eval(), model.cuda().eval()


def _check_index_artefacts(oracle, model):
    """Level coordinates, neighbour tables and pair lists of every map the U-Net used, bit for bit."""
    lo, lp = oracle.lidar_backbone.last_index, model.lidar_backbone.last_index
    for lvl in ("x0", "x1", "x2", "x3", "x4"):
        assert np.array_equal(lp[lvl].C.cpu().numpy(), lo[lvl].C), lvl
    cm = lp["x0"].cm
    okm = lo["x0"].kernel_maps
    checked = 0
    for (ks, cur, s), km in cm.kernel_maps.items():
        ref_idx, ref_out = okm["k%s_os%d_s%d_d1" % (ks, cur, s)]
        nbr = km.nbr.cpu().numpy()
        assert np.array_equal(nbr.astype(np.int64), ref_idx), (ks, cur, s)
        assert np.array_equal(km.out_coords.cpu().numpy(), ref_out), (ks, cur, s)
        kk, oo = np.nonzero(nbr >= 0)
        assert km.n_pairs == len(kk)
        assert np.array_equal(km.pair_out.cpu().numpy(), oo) and np.array_equal(km.pair_in.cpu().numpy(), nbr[kk, oo])
        assert np.array_equal(km.koff.cpu().numpy(), np.concatenate([[0], np.cumsum((nbr >= 0).sum(1))]))
        checked += 1
    assert checked == 9 and len(okm) == 9   # 5 submanifold maps + 4 strided maps


def test_full_size_kitti_frame_through_the_12_block_trunk():
    """configs[1]: one whole synthetic SemanticKITTI-shaped frame (no point cap), full DeiT-B trunk, taps 5 and 11."""
    from fusiontransformer_amd.data.synth import make_batch
    cfg, oracle, model = _full_pair("middle")
    batch = make_batch([0])
    assert batch["coords"].shape[0] > 18000
    with torch.no_grad():
        ref = oracle(oracle_inputs(batch))
        out = model(product_inputs(batch))
    for k in ref:
        err = (out[k].cpu() - ref[k]).abs().max().item()
        assert err <= TOL, (k, err)
    _check_index_artefacts(oracle, model)


def test_full_size_nuscenes_shaped_frame():
    """configs[2] shape: 900x1600 image, lift to (900, 1600), ~27 k points, full trunk."""
    from fusiontransformer_amd.data.synth import SHAPES, make_batch
    hw = (SHAPES["nuscenes"]["H"], SHAPES["nuscenes"]["W"])
    cfg, oracle, model = _full_pair("middle", lift_size=hw, seed=1)
    batch = make_batch([2], shape="nuscenes")
    assert batch["coords"].shape[0] > 22000 and batch["img"].shape[-2:] == hw
    with torch.no_grad():
        ref = oracle(oracle_inputs(batch))
        out = model(product_inputs(batch))
    for k in ref:
        err = (out[k].cpu() - ref[k]).abs().max().item()
        assert err <= TOL, (k, err)
    _check_index_artefacts(oracle, model)


def test_one_model_serves_kitti_and_nuscenes_shaped_batches():
    """configs[4] "mixed NuScenes+SemanticKITTI batches": lift_size="image" takes the lift size from each batch's own
    image, so ONE model runs both shapes back to back.  The oracle is the reference's fixed-size module built twice
    (the lift size is not a parameter, the state_dict is shared)."""
    from fusiontransformer_amd.data.synth import SHAPES, make_batch
    from fusiontransformer_amd.models.build import build_model
    from oracle import ft_oracle as O
    cfg = small_cfg("middle")
    cfg.MODEL.lift_size = "image"
    torch.manual_seed(2)
    model, _, _ = build_model(cfg)
    model = model.cuda().eval()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    refs = {}
    for shape in ("kitti", "nuscenes", "kitti"):
        hw = (SHAPES[shape]["H"], SHAPES[shape]["W"])
        ocfg = dict(cfg.MODEL)
        ocfg["lift_size"] = hw
        oracle = O.build_model(ocfg)
        oracle.load_state_dict(sd)
        oracle.eval()
        batch = make_batch([7], shape=shape, max_points=4000)
        with torch.no_grad():
            ref = oracle(oracle_inputs(batch))
            out = model(product_inputs(batch))
        for k in ref:
            err = (out[k].cpu() - ref[k]).abs().max().item()
            assert err <= TOL, (shape, k, err)
        refs[shape] = ref
    # an explicit per-batch size in the data dict overrides the model's
    pin = product_inputs(make_batch([7], shape="nuscenes", max_points=4000))
    pin["lift_size"] = (SHAPES["nuscenes"]["H"], SHAPES["nuscenes"]["W"])
    cfg2 = small_cfg("middle")
    torch.manual_seed(2)
    fixed, _, _ = build_model(cfg2)               # model built with the reference literal (370, 1226)
    fixed.load_state_dict(sd)
    with torch.no_grad():
        out2 = fixed.cuda().eval()(pin)
    for k in refs["nuscenes"]:
        assert (out2[k].cpu() - refs["nuscenes"][k]).abs().max().item() <= TOL, k


def test_bf16_forward_mode_on_mixed_full_size_frames():
    """BASELINE configs[4] at full size on one GPU: the 12-block trunk with bf16 GEMM operands (the "bf16 forward" mode), one model
    with lift_size="image" taking a whole SemanticKITTI-shaped frame and then a whole NuScenes-shaped frame.  The reference has no bf16
    mode, so the bar is the fp32 oracle with the looser tolerance stated in tests/test_model_gpu.py: per-point logits within 3e-2."""
    from fusiontransformer_amd.config import fusion_cfg
    from fusiontransformer_amd.data.synth import SHAPES, make_batch
    from fusiontransformer_amd.models.build import build_model
    from oracle import ft_oracle as O
    cfg = fusion_cfg("middle")
    cfg.MODEL.lift_size = "image"
    torch.manual_seed(4)
    model, _, _ = build_model(cfg)
    model = model.cuda().eval()
    model.image_backbone.backbone.set_bf16(True)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    for shape, seed in (("kitti", 1), ("nuscenes", 4)):
        ocfg = dict(cfg.MODEL)
        ocfg["lift_size"] = (SHAPES[shape]["H"], SHAPES[shape]["W"])
        oracle = O.build_model(ocfg)
        oracle.load_state_dict(sd)
        oracle.eval()
        batch = make_batch([seed], shape=shape)
        assert batch["coords"].shape[0] > 18000
        with torch.no_grad():
            ref = oracle(oracle_inputs(batch))
            out = model(product_inputs(batch))
        err = {k: (out[k].cpu() - ref[k]).abs().max().item() for k in ref}
        assert max(err.values()) <= 3e-2, (shape, err)
        assert err["img_seg_logit"] > 1e-6, "bf16 mode did not engage"
        del oracle


def _train_grads(model, pin, seed=0):
    from fusiontransformer_amd.trainer import fusion_losses
    model.zero_grad(set_to_none=True)
    torch.manual_seed(seed)          # dropout masks
    out = model(pin)
    l2, l3 = fusion_losses(out, pin["seg_label"], None, 0.1, True)
    (l2 + l3).backward()
    torch.cuda.synchronize()
    return ({k: v.detach().clone() for k, v in out.items()},
            {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})


@pytest.mark.parametrize("second", ["different_n", "same_n_other_indices"])
def test_two_different_batches_through_one_model_in_train_mode(second):
    """Step 2 of a model that has already trained on another batch must equal step 1 of a FRESH model on that batch,
    bit for bit (the kernels are deterministic): nothing of batch 1 -- sorted lift segments, kernel maps, graphs'
    static inputs -- may leak into batch 2.  The freed index tensors of batch 1 are recycled by the caching allocator,
    which is exactly the situation an address-keyed cache gets wrong."""
    from fusiontransformer_amd.data.synth import make_batch
    from fusiontransformer_amd.models.build import build_model
    cfg = small_cfg("middle")

    def fresh():
        torch.manual_seed(11)
        m, _, _ = build_model(cfg)
        return m.cuda().train()

    b1 = make_batch([0, 1], max_points=2500)
    if second == "different_n":
        b2 = make_batch([2, 3], max_points=1800)
    else:
        # same frames, same N, same coordinates -- only the pixel each point reads differs (rolled within each frame)
        b2 = {k: (list(v) if isinstance(v, list) else v.copy()) for k, v in b1.items()}
        b2["img_indices"] = [np.roll(a, 37, axis=0).copy() for a in b1["img_indices"]]
        assert not np.array_equal(b2["img_indices"][0], b1["img_indices"][0])

    ref_model = fresh()
    ref_out, ref_grads = _train_grads(ref_model, product_inputs(b2))

    model = fresh()
    pin1 = product_inputs(b1)
    _train_grads(model, pin1)
    # restore parameters / buffers touched by step 1 (running stats, counters): compare like with like
    model.load_state_dict(fresh().state_dict())
    del pin1                              # batch 1's device tensors go back to the allocator
    out, grads = _train_grads(model, product_inputs(b2))
    for k in ref_out:
        assert torch.equal(out[k], ref_out[k]), k
    assert grads.keys() == ref_grads.keys()
    for n in grads:
        assert torch.equal(grads[n], ref_grads[n]), n


def test_step_two_gradients_match_the_oracle():
    """The same property against the checker: gradients of a SECOND, different batch (train mode, injected dropout
    masks) within the train-step tolerance of the float64 oracle."""
    from fusiontransformer_amd.data.synth import make_batch
    from fusiontransformer_amd.models.build import build_model
    from fusiontransformer_amd.trainer import fusion_losses
    from oracle import ft_oracle as O
    cfg = small_cfg("middle")
    torch.manual_seed(21)
    oracle = O.build_model(dict(cfg.MODEL)).double().train()
    model, _, _ = build_model(cfg)
    model.load_state_dict({k: v.float() for k, v in oracle.state_dict().items()})
    model = model.cuda().train()
    cw = torch.tensor(cfg.TRAIN.CLASS_WEIGHTS)
    b1, b2 = make_batch([0, 1], max_points=1500), make_batch([4, 5], max_points=2000)

    # batch 1 through the product only (no optimizer step: parameters stay equal to the oracle's)
    model.lidar_backbone.dropout_masks = None
    pin = product_inputs(b1)
    out = model(pin)
    l2, l3 = fusion_losses(out, pin["seg_label"], cw.cuda(), 0.1, True)
    (l2 + l3).backward()
    model.zero_grad(set_to_none=True)
    del pin, out, l2, l3

    # batch 2 through both
    with torch.no_grad():
        oracle.eval(); i64 = oracle_inputs(b2); i64["img"], i64["lidar"].F = i64["img"].double(), i64["lidar"].F.double()
        oracle(i64); oracle.train()
    li = oracle.lidar_backbone.last_index
    g = torch.Generator().manual_seed(5)
    masks = {"y1": (torch.rand(li["x4"].C.shape[0], 256, generator=g) > 0.3).double(),
             "y3": (torch.rand(li["x2"].C.shape[0], 128, generator=g) > 0.3).double()}
    oracle.lidar_backbone.dropout_masks = masks
    model.lidar_backbone.dropout_masks = {k: v.float().cuda() for k, v in masks.items()}
    i64 = oracle_inputs(b2); i64["img"], i64["lidar"].F = i64["img"].double(), i64["lidar"].F.double()
    a, b = O.fusion_losses(oracle(i64), torch.from_numpy(b2["seg_label"]), cw.double(), 0.1, True)
    (a + b).backward()
    pin = product_inputs(b2)
    out = model(pin)
    l2, l3 = fusion_losses(out, pin["seg_label"], cw.cuda(), 0.1, True)
    (l2 + l3).backward()
    assert abs(l2.item() - a.item()) < 1e-4 and abs(l3.item() - b.item()) < 1e-4
    p64, pp = dict(oracle.named_parameters()), dict(model.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in p64.values() if p.grad is not None)
    worst = (0.0, None)
    for name, p in p64.items():
        if p.grad is None:
            continue
        gp = pp[name].grad.cpu().double()
        floor = 1e-4 * gmax * p.numel() ** 0.5
        rel = (gp - p.grad).norm().item() / max(p.grad.norm().item(), floor)
        if rel > worst[0]:
            worst = (rel, name)
    assert worst[0] < 5e-2, worst
