"""Tiny stand-in for the yacs config tree of the reference (yacs is not installed).

Only the keys the hot path reads are given defaults here, with the reference's
values (config/FusionTransformerConfig.py:8-10,124-139, common/config/base.py:
26-50); the reference's YAML files (configs/semantic_kitti/*.yaml) load as-is
through `cfg.merge_from_file`.  A CfgNode is a Mapping, so `SPVCNN(**cfg.MODEL)`
(models/build.py:32-36) works as it does with yacs."""
from __future__ import annotations

from collections.abc import Mapping

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, Mapping) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def merge_from_dict(self, other):
        for k, v in other.items():
            if isinstance(v, Mapping) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, Mapping) else v

    def merge_from_file(self, path):
        with open(path, "r") as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = yaml.safe_load(val) if isinstance(val, str) else val

    def clone(self):
        return CfgNode(self)

    def freeze(self):
        return self


def get_cfg_defaults() -> CfgNode:
    return CfgNode({
        "MODEL": {
            "TYPE": "", "SAVE": True, "CKPT_PATH": "", "NUM_CLASSES": 20, "DUAL_HEAD": False,
            "USE_IMAGE": False, "USE_LIDAR": False, "USE_FUSION": False, "IMAGE_PRETRAINED_PATH": "",
            "middle_feat_block_number": None, "late_feat_block_number": None,
        },
        "OPTIMIZER": {"TYPE": "", "BASE_LR": 0.001, "WEIGHT_DECAY": 0.0, "Adam": {"betas": (0.9, 0.999)}},
        "TRAIN": {"BATCH_SIZE": 0, "CLASS_WEIGHTS": [], "FusionTransformer": {"lambda_xm": 0.0}},
        "DATALOADER": {"NUM_WORKERS": 0, "DROP_LAST": True},
    })


# The live fusion configs of the reference (configs/semantic_kitti/{middle,early,late}fusion.yaml),
# MODEL + loss + optimizer parts, so the bench does not need the reference tree at run time.
_CLASS_WEIGHTS = [0., 1.58003993, 3.69774469, 3.2460013, 2.65342029, 2.61079801, 3.27744058, 3.48282471, 3.45874555, 1.,
                  2.07298878, 1.26831551, 2.65889542, 1.37436805, 1.4891881, 1.03083152, 2.25629999, 1.51838281, 2.51986332,
                  3.08564901]


def fusion_cfg(kind: str = "middle") -> CfgNode:
    cfg = get_cfg_defaults()
    model = {"middle": dict(TYPE="MiddleFusionTransformer", middle_feat_block_number=5),
             "early": dict(TYPE="EarlyFusionTransformer", middle_feat_block_number=0),
             "late": dict(TYPE="LateFusionTransformer")}[kind]
    cfg.merge_from_dict({
        "MODEL": dict(DUAL_HEAD=True, NUM_CLASSES=20, late_feat_block_number=11, USE_IMAGE=True, USE_LIDAR=True, USE_FUSION=True, **model),
        "OPTIMIZER": {"TYPE": "Adam", "BASE_LR": 1e-4, "WEIGHT_DECAY": 0.0005},
        "TRAIN": {"BATCH_SIZE": 10, "CLASS_WEIGHTS": _CLASS_WEIGHTS, "FusionTransformer": {"lambda_xm": 0.1}},
    })
    return cfg
