"""Internal-consistency checks of the CPU oracle for the parts no reference fixture pins
(torchsparse / timm arithmetic): sparse convolution against torch's dense conv3d, adjointness and
float64 gradchecks of voxelize / devoxelize, hash known answers, ViT block against the DeiT
layer of HuggingFace transformers (a secondary, config-constructed sanity source)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ft_oracle as O


def test_hash_known_answer():
    def fnv(c):
        h = 14695981039346656037
        for v in c:
            h ^= v & 0xFFFFFFFF
            h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFF)
    cs = np.array([[0, 0, 0, 0], [1, 2, 3, 0], [-1, 5, 7, 1], [4095, 4095, 4095, 31]], dtype=np.int32)
    assert [int(v) for v in O.sphash(cs)] == [fnv(c.tolist()) for c in cs]
    off = O.kernel_offsets(2, 4)
    assert off.tolist() == [[0, 0, 0], [0, 0, 4], [0, 4, 0], [0, 4, 4], [4, 0, 0], [4, 0, 4], [4, 4, 0], [4, 4, 4]]
    assert O.kernel_offsets(3, 1)[:4].tolist() == [[-1, -1, -1], [0, -1, -1], [1, -1, -1], [-1, 0, -1]]
    assert int(O.sphash(cs, off)[5, 1]) == fnv([1 + 4, 2, 3 + 4, 0])


def test_hashquery_and_count_edge_cases():
    assert O.sphashquery(np.array([5, 6]), np.array([], dtype=np.int64)).tolist() == [-1, -1]
    t = np.array([9, 3, 7], dtype=np.int64)
    assert O.sphashquery(np.array([7, 9, 4, 3]), t).tolist() == [2, 0, -1, 1]
    assert O.spcount(np.array([0, 0, -1, 2]), 4).tolist() == [2, 0, 1, 0]


def _dense_setup(rng, D=8, cin=3, p=0.35):
    occ = rng.random((D, D, D)) < p
    xyz = np.argwhere(occ).astype(np.int32)
    coords = np.concatenate([xyz, np.zeros((len(xyz), 1), np.int32)], 1)
    feats = torch.from_numpy(rng.standard_normal((len(xyz), cin))).double()
    dense = torch.zeros(1, cin, D, D, D, dtype=torch.float64)
    dense[0, :, xyz[:, 0], xyz[:, 1], xyz[:, 2]] = feats.T
    return coords, xyz, feats, dense


def test_sparse_conv_equals_dense_conv3d():
    rng = np.random.default_rng(0)
    coords, xyz, feats, dense = _dense_setup(rng)
    cin, cout = 3, 5
    # submanifold 3x3x3: kernel index k = (z+1)*9 + (y+1)*3 + (x+1) (x fastest)
    w = torch.from_numpy(rng.standard_normal((27, cin, cout))).double()
    idx, out_coords = O.build_kernel_map(coords, 1, 3, 1)
    sp = O.sparseconv_op(feats, w, idx, len(coords), False)
    wd = w.view(3, 3, 3, cin, cout).permute(4, 3, 2, 1, 0).contiguous()  # [o,i,dx,dy,dz] from [dz,dy,dx,i,o]
    ref = F.conv3d(dense, wd, padding=1)[0][:, xyz[:, 0], xyz[:, 1], xyz[:, 2]].T
    assert torch.allclose(sp, ref, atol=1e-12)
    # strided 2x2x2: k = x*4 + y*2 + z (z fastest)
    w2 = torch.from_numpy(rng.standard_normal((8, cin, cout))).double()
    idx2, oc2 = O.build_kernel_map(coords, 1, 2, 2)
    sp2 = O.sparseconv_op(feats, w2, idx2, len(oc2), False)
    wd2 = w2.view(2, 2, 2, cin, cout).permute(4, 3, 0, 1, 2).contiguous()  # [o,i,dx,dy,dz]
    d2 = F.conv3d(dense, wd2, stride=2)[0]
    ref2 = d2[:, oc2[:, 0] // 2, oc2[:, 1] // 2, oc2[:, 2] // 2].T
    assert torch.allclose(sp2, ref2, atol=1e-12)
    # output coordinates = occupied 2^3 cells, ordered by ascending hash
    cells = np.unique(xyz // 2 * 2, axis=0)
    assert len(cells) == len(oc2) and np.all(np.diff(O.sphash(oc2)) > 0)
    # transposed conv on the same map == conv_transpose3d evaluated at the fine occupied voxels
    coarse = torch.from_numpy(rng.standard_normal((len(oc2), cout))).double()
    wt = torch.from_numpy(rng.standard_normal((8, cout, cin))).double()
    up = O.sparseconv_op(coarse, wt, idx2, len(coords), True)
    dc = torch.zeros(1, cout, 4, 4, 4, dtype=torch.float64)
    dc[0, :, oc2[:, 0] // 2, oc2[:, 1] // 2, oc2[:, 2] // 2] = coarse.T
    wtd = wt.view(2, 2, 2, cout, cin).permute(3, 4, 0, 1, 2).contiguous()  # conv_transpose weight [in,out,dx,dy,dz]
    refu = F.conv_transpose3d(dc, wtd, stride=2)[0][:, xyz[:, 0], xyz[:, 1], xyz[:, 2]].T
    assert torch.allclose(up, refu, atol=1e-12)


def test_voxelize_devoxelize_gradcheck_and_adjoint():
    rng = np.random.default_rng(1)
    n, m, c = 40, 9, 3
    idx = rng.integers(-1, m, n)
    counts = O.spcount(idx, m)
    x = torch.from_numpy(rng.standard_normal((n, c))).double().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda t: O.spvoxelize(t, idx, counts), (x,), atol=1e-8)
    idx8 = rng.integers(-1, m, (n, 8))
    w8 = rng.random((n, 8))
    f = torch.from_numpy(rng.standard_normal((m, c))).double().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda t: O.spdevoxelize(t, idx8, w8), (f,), atol=1e-8)
    # voxelize of a constant is that constant on non-empty voxels
    ones = O.spvoxelize(torch.ones(n, 1, dtype=torch.float64), idx, counts)
    assert torch.allclose(ones[counts > 0], torch.ones(int((counts > 0).sum()), 1, dtype=torch.float64))


def test_trilinear_weights_properties():
    rng = np.random.default_rng(2)
    pc = np.concatenate([rng.integers(0, 64, (200, 3)), np.zeros((200, 1))], 1).astype(np.float32)
    idx = np.zeros((8, 200), dtype=np.int64)
    w1 = O.calc_ti_weights(pc, idx, 1)
    assert np.allclose(w1[0], 1.0) and np.allclose(w1[1:], 0.0)  # integer coordinates at stride 1: corner 0 only
    w4 = O.calc_ti_weights(pc, idx, 4)
    assert np.allclose(w4.sum(0), 1.0, atol=1e-6)
    idx[3] = -1
    w4m = O.calc_ti_weights(pc, idx, 4)
    assert np.all(w4m[3] == 0)


def test_initial_voxelize_orders_by_hash_and_dedupes():
    rng = np.random.default_rng(3)
    c = rng.integers(0, 6, (300, 4)).astype(np.int64)
    c[:, 3] = rng.integers(0, 2, 300)
    z = O.PointTensor(torch.from_numpy(rng.standard_normal((300, 4)).astype(np.float32)), c.astype(np.float32))
    x0 = O.initial_voxelize(z, 1, 1)
    assert len(np.unique(c, axis=0)) == x0.C.shape[0]
    assert np.all(np.diff(O.sphash(x0.C)) > 0)
    iq = z.additional_features["idx_query"][1]
    assert np.array_equal(x0.C[iq], c.astype(np.int32))
    # ragged / empty frame in the batch: a batch index with no points simply does not appear
    assert set(np.unique(x0.C[:, 3])) <= {0, 1}


def test_vit_block_matches_huggingface_deit_layer():
    tr = pytest.importorskip("transformers")
    from transformers.models.deit.configuration_deit import DeiTConfig
    from transformers.models.deit.modeling_deit import DeiTLayer
    cfg = DeiTConfig(hidden_size=64, num_attention_heads=4, intermediate_size=256, layer_norm_eps=1e-6, hidden_act="gelu",
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, qkv_bias=True)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    torch.manual_seed(0)
    hf = DeiTLayer(cfg).eval()
    blk = O.Block(64, 4).eval()
    mods = dict(hf.named_modules())
    # module names differ between transformers releases
    if "attention.q_proj" in mods:
        q, k, v, o = (mods["attention." + n] for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
        fc1, fc2 = mods["mlp.fc1"], mods["mlp.fc2"]
    else:
        att = hf.attention.attention
        q, k, v, o = att.query, att.key, att.value, hf.attention.output.dense
        fc1, fc2 = hf.intermediate.dense, hf.output.dense
    with torch.no_grad():
        blk.norm1.load_state_dict(hf.layernorm_before.state_dict())
        blk.norm2.load_state_dict(hf.layernorm_after.state_dict())
        blk.attn.qkv.weight.copy_(torch.cat([q.weight, k.weight, v.weight], 0))
        blk.attn.qkv.bias.copy_(torch.cat([q.bias, k.bias, v.bias], 0))
        blk.attn.proj.load_state_dict(o.state_dict())
        blk.mlp.fc1.load_state_dict(fc1.state_dict())
        blk.mlp.fc2.load_state_dict(fc2.state_dict())
        x = torch.randn(2, 10, 64)
        ref = hf(x)
        ref = ref[0] if isinstance(ref, (tuple, list)) else ref
        assert torch.allclose(blk(x), ref, atol=1e-5)
