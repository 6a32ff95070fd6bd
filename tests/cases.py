"""The seeded parity cases shared by the oracle tests (CPU) and the HIP parity tests (GPU).

Each case is (architecture, B, T, config overrides); the same table drives
`oracle/gen_golden.py`, which ran the reference itself on these inputs to produce
`tests/golden/<case>.npz`.
"""
import os

import numpy as np

from dfd_clip_amd.config import ConfigNode, default_detector_config
from dfd_clip_amd.weights import ARCHS, random_state_dict, resolve_layer_indices, synthetic_clips

CASES = {
    "tiny": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1])),
    "tiny_stride": ("tiny", 2, 4, dict()),
    "tiny_adapter_nln": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                            adapter__frozen=0, adapter__struct={"type": "768-x-768-nln", "x": 32})),
    "tiny_adapter_ln": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                           adapter__frozen=0, adapter__struct={"type": "768-x-768-ln", "x": 32})),
    "tiny_adapter_gl": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                           adapter__frozen=0, adapter__struct={"type": "768-x-768", "x": 32})),
    "tiny_adapter_legacy": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                               adapter__frozen=0, adapter__struct={"type": "legacy-768-x-768", "x": 32})),
    "tiny_global": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__global_prediction=1)),
    "tiny_attnmode": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__attn_mode="frame+temporal")),
    "tiny_nopos": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__temporal_position=0)),
    "tiny_augq": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__aug_query=1)),
    "small": ("small", 2, 3, dict(decode_mode="index", decode_indices=[1, 2])),
    "small14": ("small14", 2, 3, dict(decode_mode="index", decode_indices=[0, 1])),
    "vitb16_cfg1": ("ViT-B/16", 2, 8, dict(decode_mode="index", decode_indices=[6, 7, 8, 9, 10, 11])),
    "vitl14": ("ViT-L/14", 2, 2, dict(decode_mode="stride", decode_stride=2)),
    "tiny_ema": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__ema_frame=0.3,
                                    op_mode__temporal_position=0)),
    "tiny_rank": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], train_mode__temporal="ranking")),
    "tiny_pmask": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1],
                                      train_mode__patch_mask={"type": "batch", "ratio": 0.5})),
}
EXTRA_INPUTS = dict(comp=["raw", "c23"], speed=[1.0, 2.5], np_seed=11)


def make_config(arch, **over):
    cfg = default_detector_config()
    cfg.architecture = arch
    cfg.out_dim = [2]
    cfg.losses = ["auc_roc"]
    for k, v in over.items():
        node = cfg
        parts = k.split("__")
        for p in parts[:-1]:
            if p not in node:
                node[p] = ConfigNode()
            node = node[p]
        node[parts[-1]] = v
    return cfg


def build_case(name):
    arch, B, T, over = CASES[name]
    res, patch, width, layers, heads, _ = ARCHS[arch]
    cfg = make_config(arch, **over)
    sd = random_state_dict(cfg, T, seed=0)
    x, m, y = synthetic_clips(B, T, res, seed=1234, masked_tail=("attn_mode" not in str(over)))
    return dict(name=name, arch=arch, B=B, T=T, cfg=cfg, sd=sd, x=x, m=m, y=y, res=res, patch=patch,
                width=width, layers=layers, heads=heads, layer_indices=resolve_layer_indices(cfg, layers))


def oracle_kwargs(case):
    cfg = case["cfg"]
    op = cfg.op_mode
    return dict(
        heads=case["heads"], patch=case["patch"], layer_indices=case["layer_indices"], out_dims=list(cfg.out_dim),
        num_frames=case["T"],
        adapter_struct=(cfg.adapter.struct.type if cfg.adapter.type != "none" else None),
        attn_mode=tuple(op.attn_mode.split("+")) if "attn_mode" in op else (),
        global_prediction=bool("global_prediction" in op and op.global_prediction))


def load_golden(name):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    return np.load(path, allow_pickle=False)
