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
    # GELU-first adapters at the real width (768 -> 256 -> 768 on ViT-B/16 keys / values, 1 clip x 2 frames, layers 10 and
    # 11 tapped): the LayerNorm behind the GELU normalises 256 values per row here, not the tiny model's 32
    "vitb16_adapter_gl": ("ViT-B/16", 1, 2, dict(decode_mode="index", decode_indices=[10, 11], adapter__type="normal",
                                                 adapter__frozen=0, adapter__struct={"type": "768-x-768", "x": 256})),
    "vitb16_adapter_legacy": ("ViT-B/16", 1, 2, dict(decode_mode="index", decode_indices=[10, 11], adapter__type="normal",
                                                     adapter__frozen=0, adapter__struct={"type": "legacy-768-x-768", "x": 256})),
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


def synthetic_clip_checkpoint(arch="tiny", seed=3, dtype=None):
    """A complete synthetic CLIP state_dict in the published key layout (`visual.`-prefixed ViT tower of `arch` plus a
    one-block text tower of width 64, the keys reference clip/model.py:453-496 `build_model` insists on), fp32, seeded.
    `oracle/gen_golden.py loader` ran the reference's own `build_model(sd).visual.float()` on exactly this dictionary to
    produce tests/golden/clip_loader_<arch>.npz; tests/test_host_cpu.py feeds it to `load_clip_visual`."""
    import torch
    from dfd_clip_amd.weights import _fill, encoder_schema
    rng = np.random.default_rng(seed)
    sd = {"visual." + k: _fill(rng, k, shp) for k, shp in encoder_schema(arch).items()}
    tw, ctx, vocab, embed = 64, 4, 16, ARCHS[arch][5]

    def rnd(*shape, scale=0.05):
        return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))

    sd.update({"token_embedding.weight": rnd(vocab, tw), "positional_embedding": rnd(ctx, tw), "text_projection": rnd(tw, embed),
               "logit_scale": torch.tensor(2.5), "ln_final.weight": 1 + rnd(tw), "ln_final.bias": rnd(tw)})
    pre = "transformer.resblocks.0."
    sd.update({pre + "attn.in_proj_weight": rnd(3 * tw, tw), pre + "attn.in_proj_bias": rnd(3 * tw),
               pre + "attn.out_proj.weight": rnd(tw, tw), pre + "attn.out_proj.bias": rnd(tw),
               pre + "ln_1.weight": 1 + rnd(tw), pre + "ln_1.bias": rnd(tw), pre + "ln_2.weight": 1 + rnd(tw), pre + "ln_2.bias": rnd(tw),
               pre + "mlp.c_fc.weight": rnd(4 * tw, tw), pre + "mlp.c_fc.bias": rnd(4 * tw),
               pre + "mlp.c_proj.weight": rnd(tw, 4 * tw), pre + "mlp.c_proj.bias": rnd(tw)})
    if dtype is not None:
        sd = {k: (v.to(dtype) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
    return sd
