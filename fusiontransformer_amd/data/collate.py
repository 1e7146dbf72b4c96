"""collate_scn_base: mirror of FusionTransformer/data/collate.py:6-92 for the keys the model
consumes (`lidar`, `seg_label`, `img`, `img_indices`); the bookkeeping keys
(`seq`, `filename`, `voxel_coords`, `orig_*`) are carried when present."""
from functools import partial

import numpy as np
import torch

from ..sparse import SparseTensor


def collate_scn_base(input_dict_list, output_orig, output_image=True):
    locs, feats, labels, imgs, img_idxs = [], [], [], [], []
    extras = {k: [] for k in ("seq", "filename", "voxel_coords", "orig_seg_label", "sparse_orig_points_idx", "inverse_map")}
    for idx, input_dict in enumerate(input_dict_list):
        coords = torch.from_numpy(np.asarray(input_dict["coords"]))
        batch_idxs = torch.LongTensor(coords.shape[0], 1).fill_(idx)
        locs.append(torch.cat([coords.long(), batch_idxs], 1))  # locs = [coords, batch]
        feats.append(torch.from_numpy(input_dict["feats"]))
        if "seg_label" in input_dict.keys():
            labels.append(torch.from_numpy(input_dict["seg_label"]))
        if output_image:
            imgs.append(torch.from_numpy(input_dict["img"]))
            img_idxs.append(input_dict["img_indices"])
        for k in extras:
            if k in input_dict and (output_orig or k in ("seq", "filename", "voxel_coords")):
                extras[k].append(input_dict[k])
    locs = torch.cat(locs, 0)
    feats = torch.cat(feats, 0)
    out_dict = {"lidar": SparseTensor(coords=locs.int(), feats=feats)}
    if labels:
        out_dict["seg_label"] = torch.cat(labels, 0)
    if output_image:
        out_dict["img"] = torch.stack(imgs)
        out_dict["img_indices"] = img_idxs
    for k, v in extras.items():
        if v:
            out_dict[k] = v
    return out_dict


def get_collate_scn(is_train):
    return partial(collate_scn_base, output_orig=not is_train)
