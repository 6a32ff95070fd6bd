"""Weight schema, seeded random initialisation and device-side preparation.

The state_dict key names and shapes are the reference's (`Detector.state_dict()`, schema
listed in SURVEY.md §8b; producers: reference `src/clip/model.py:255-274`, `:176-181`,
`:203-213` and `src/models.py:130-131`, `:155-166`, `:287-318`), so a reference
`best_weights.pt` loads unchanged and a checkpoint written here loads into the reference.

Pretrained CLIP weights cannot be fetched offline (reference `src/clip/clip.py:30-40`), so
benchmarks and parity tests use `random_state_dict`: a numpy PCG64 stream (stable across
machines and library versions, unlike `torch.manual_seed` streams) scaled so that
activations stay O(1) through the stack.
"""
from collections import OrderedDict

import numpy as np
import torch

# name -> (input_resolution, patch_size, width, layers, heads, output_dim)
# (reference `src/clip/model.py:453-496` infers the same numbers from checkpoint shapes).
ARCHS = {
    "ViT-B/16": (224, 16, 768, 12, 12, 512),
    "ViT-B/32": (224, 32, 768, 12, 12, 512),
    "ViT-L/14": (224, 14, 1024, 24, 16, 768),
    # test-only shapes: every layout rule of the real model at a fraction of the size
    "tiny": (32, 16, 128, 2, 2, 64),
    "small": (224, 16, 256, 3, 4, 64),
    "small14": (224, 14, 128, 2, 2, 64),  # ViT-L/14's token geometry: 14x14 patches (K = 588), 257 tokens
}


def encoder_schema(arch):
    res, patch, width, layers, heads, out_dim = ARCHS[arch]
    tokens = (res // patch) ** 2 + 1
    s = OrderedDict()
    s["class_embedding"] = (width,)
    s["positional_embedding"] = (tokens, width)
    s["proj"] = (width, out_dim)
    s["conv1.weight"] = (width, 3, patch, patch)
    s["ln_pre.weight"] = (width,)
    s["ln_pre.bias"] = (width,)
    for l in range(layers):
        p = f"transformer.resblocks.{l}."
        s[p + "attn.in_proj_weight"] = (3 * width, width)
        s[p + "attn.in_proj_bias"] = (3 * width,)
        s[p + "attn.out_proj.weight"] = (width, width)
        s[p + "attn.out_proj.bias"] = (width,)
        s[p + "ln_1.weight"] = (width,)
        s[p + "ln_1.bias"] = (width,)
        s[p + "mlp.c_fc.weight"] = (4 * width, width)
        s[p + "mlp.c_fc.bias"] = (4 * width,)
        s[p + "mlp.c_proj.weight"] = (width, 4 * width)
        s[p + "mlp.c_proj.bias"] = (width,)
        s[p + "ln_2.weight"] = (width,)
        s[p + "ln_2.bias"] = (width,)
    s["ln_post.weight"] = (width,)
    s["ln_post.bias"] = (width,)
    return s


def decoder_schema(arch, num_frames, n_blocks, out_dims, temporal_position=True,
                   aug_query=False, global_prediction=False, layer_indices=None):
    res, patch, width, layers, heads, _ = ARCHS[arch]
    s = OrderedDict()
    s["class_embedding"] = (width,)
    if temporal_position:
        s["positional_embedding"] = (num_frames, 1, heads, width // heads)
    s["ln_pre.weight"] = (width,)
    s["ln_pre.bias"] = (width,)
    if aug_query:
        for i in range(n_blocks - 1):
            s[f"transformer.augment_query_{i}"] = (width,)
    for b in range(n_blocks):
        p = f"transformer.resblocks.{b}."
        s[p + "attn.in_proj.weight"] = (2 * width, width)
        s[p + "attn.in_proj.bias"] = (2 * width,)
        s[p + "attn.out_proj.weight"] = (width, width)
        s[p + "attn.out_proj.bias"] = (width,)
        s[p + "ln_1.weight"] = (width,)
        s[p + "ln_1.bias"] = (width,)
        s[p + "mlp.c_fc.weight"] = (4 * width, width)
        s[p + "mlp.c_fc.bias"] = (4 * width,)
        s[p + "mlp.c_proj.weight"] = (width, 4 * width)
        s[p + "mlp.c_proj.bias"] = (width,)
        s[p + "ln_2.weight"] = (width,)
        s[p + "ln_2.bias"] = (width,)
    s["ln_post.weight"] = (width,)
    s["ln_post.bias"] = (width,)
    for i, od in enumerate(out_dims):
        if global_prediction:
            for l in layer_indices:
                s[f"proj{i}x{od}_L{l}"] = (width, od)
        else:
            s[f"proj{i}x{od}"] = (width, od)
    return s


def adapter_schema(arch, n_blocks, struct_type, x):
    """`CompInvAdapter` parameters (reference `src/models.py:783-928`), the LayerNorm variants."""
    res, patch, width, layers, heads, _ = ARCHS[arch]
    patches = (res // patch) ** 2
    s = OrderedDict()
    for i in range(n_blocks):
        for j in ("k", "v"):
            p = f"l{i}_{j}."
            if struct_type == "768-x-768-nln":
                s[p + "0.weight"] = (x, width)
                s[p + "1.weight"] = (patches, x)
                s[p + "1.bias"] = (patches, x)
                s[p + "4.weight"] = (width, x)
            elif struct_type in ("768-x-768-ln", "768-x-768-z0"):
                s[p + "0.weight"] = (x, width)
                s[p + "1.weight"] = (x,)
                s[p + "1.bias"] = (x,)
                s[p + "4.weight"] = (width, x)
            elif struct_type in ("768-x-768", "legacy-768-x-768"):  # Linear, GELU, LayerNorm, [Dropout,] Linear
                s[p + "0.weight"] = (x, width)
                s[p + "2.weight"] = (x,)
                s[p + "2.bias"] = (x,)
                s[p + ("4.weight" if struct_type == "768-x-768" else "3.weight")] = (width, x)
            else:
                raise NotImplementedError(struct_type)
    return s


def _fill(rng, name, shape):
    """One tensor of the seed recipe.  Linear/conv weights ~ N(0, fan_in^-1) keep activations
    O(1); LayerNorm gains near 1 and all biases small but non-zero so that every gain/bias
    path is exercised by the parity tests."""
    leaf = name.split(".")[-1]
    is_ln = ".ln_" in name or name.startswith("ln_") or (name.split(".")[-2:-1] == ["1"]) or \
        (name.split(".")[-2:-1] == ["2"] and len(shape) == 1)
    if is_ln and leaf == "weight":
        a = 1.0 + 0.05 * rng.standard_normal(shape)
    elif leaf == "bias" or leaf.endswith("in_proj_bias"):
        a = 0.02 * rng.standard_normal(shape)
    elif "augment_query" in name:
        a = 0.02 * rng.standard_normal(shape)
    elif leaf in ("class_embedding", "positional_embedding", "proj") or leaf.startswith("proj"):
        width = shape[0] if leaf.startswith("proj") else shape[-1]
        if leaf == "positional_embedding" and len(shape) == 4:
            width = shape[2] * shape[3]
        a = (width ** -0.5) * rng.standard_normal(shape)
    else:  # linear / conv weight [out, in, ...]
        fan_in = int(np.prod(shape[1:]))
        a = (fan_in ** -0.5) * rng.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def resolve_layer_indices(config, n_layers):
    """Reference `src/models.py:458-463`."""
    if config.decode_mode == "stride":
        return list(range(0, n_layers, config.decode_stride))
    if config.decode_mode == "index":
        return list(config.decode_indices)
    raise Exception(f"Unknown decode type: {config.decode_mode}")


def random_state_dict(config, num_frames, seed=0):
    """Seeded fp32 `Detector` state_dict (reference key names).  The decoder's ln_1 / ln_2 /
    mlp start as copies of encoder layer `layer_indices[i]`, as the reference's
    `_apply_reference` does (`src/models.py:226-229`, concat_ref == 0)."""
    arch = config.architecture
    res, patch, width, layers, heads, _ = ARCHS[arch]
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, shp in encoder_schema(arch).items():
        sd["encoder." + k] = _fill(rng, k, shp)
    lidx = resolve_layer_indices(config, layers)
    op = config.op_mode
    dsch = decoder_schema(
        arch, num_frames, len(lidx), list(config.out_dim),
        temporal_position=bool("temporal_position" in op and op.temporal_position),
        aug_query=bool("aug_query" in op and op.aug_query),
        global_prediction=bool("global_prediction" in op and op.global_prediction),
        layer_indices=lidx)
    for k, shp in dsch.items():
        sd["decoder." + k] = _fill(rng, k, shp)
    if not ("concat_ref" in config and config.concat_ref):
        for b, l in enumerate(lidx):
            for part in ("ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight",
                         "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias"):
                sd[f"decoder.transformer.resblocks.{b}.{part}"] = \
                    sd[f"encoder.transformer.resblocks.{l}.{part}"].clone()
    if "temporal" in config.train_mode and config.train_mode.temporal == "ranking":
        sd["ranking_transform_param"] = _fill(rng, "proj", (width, 1))
    if config.adapter.type != "none":
        st = config.adapter.struct
        for k, shp in adapter_schema(arch, len(lidx), st.type, int(st.x)).items():
            sd["adapter." + k] = _fill(rng, k, shp)
    return sd


def synthetic_clips(b, t, res, seed=1234, masked_tail=True):
    """Synthetic post-`Normalize` frames and padding mask: x ~ N(0,1) fp32;
    with `masked_tail`, clip 1 has its last ceil(T/4) frames marked as padding."""
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((b, t, 3, res, res), dtype=np.float32))
    m = torch.ones(b, t, dtype=torch.bool)
    if masked_tail and b > 1:
        m[1, t - (t + 3) // 4:] = False
    y = torch.arange(b) % 2
    return x, m, y
