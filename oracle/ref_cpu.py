"""ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing here is product code.

A CPU (PyTorch fp32) restatement of the reference's hot path: CLIP-ViT per-layer K/V
extraction, optional CompInvAdapter, the cross-attention temporal decoder, logit
normalisation and the per-sample losses.  Written from the reference's behaviour as a set
of pure functions over a flat weight dict that uses the reference's state_dict key names;
each function cites the reference lines it follows (paths under /root/reference).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / the timed CPU baseline.  The product package
(`dfd-clip_amd/`) never imports it and fails loudly when its HIP library is missing.

Pinning: the reference ships no tests or golden vectors for this path (SURVEY.md §4,
§8c: "parity unpinned by the reference"), so this restatement is pinned against outputs of
the reference itself, imported and run in the build container by `oracle/gen_golden.py`;
those outputs are committed under `tests/golden/` and `tests/test_oracle_golden.py` checks
this file against them.
"""
import math

import torch
import torch.nn.functional as F


def layer_norm(x, w, b, eps=1e-5):
    """fp32 LayerNorm over the last dim (reference `src/clip/model.py:157-163`, `src/models.py:58-68`)."""
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps).to(x.dtype)


def quick_gelu(x):
    """x * sigmoid(1.702 x) (reference `src/clip/model.py:166-168`)."""
    return x * torch.sigmoid(1.702 * x)


def encoder_attention(x, w, p, heads):
    """Self-attention that also returns q, k, v (reference `src/clip/model.py:185-199`).
    in_proj rows are ordered q | k | v; head h owns channels 64h..64h+63; q is scaled by
    d^-0.5 before the product; softmax runs over the key axis."""
    qkv = F.linear(x, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    n, s, _ = q.shape
    q = q.reshape(n, s, heads, -1)
    k = k.reshape(n, s, heads, -1)
    v = v.reshape(n, s, heads, -1)
    aff = torch.einsum("nqhc,nkhc->nqkh", q / math.sqrt(q.shape[-1]), k).softmax(dim=-2)
    mix = torch.einsum("nqlh,nlhc->nqhc", aff, v)
    out = F.linear(mix.flatten(-2), w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])
    return q, k, v, out


def encoder_forward(w, x, heads, patch, prefix="encoder.", with_out=False, with_q=False):
    """`VisionTransformer.forward` (reference `src/clip/model.py:276-294`, `:236-251`,
    `:220-226`): patch conv (no bias) -> [N, P, D]; CLS row = class_embedding; + positional
    embedding; ln_pre; every block fully; returns one dict per block with k, v
    [N, tokens, heads, d] (bias included, un-scaled, CLS row included); ln_post / proj are
    never applied."""
    g = lambda k: w[prefix + k]
    y = F.conv2d(x, g("conv1.weight"), None, stride=patch)
    y = y.reshape(y.shape[0], y.shape[1], -1).permute(0, 2, 1)
    cls = g("class_embedding").to(y.dtype) + torch.zeros(y.shape[0], 1, y.shape[-1], dtype=y.dtype)
    y = torch.cat([cls, y], dim=1) + g("positional_embedding")
    y = layer_norm(y, g("ln_pre.weight"), g("ln_pre.bias"))
    n_layers = 0
    while f"{prefix}transformer.resblocks.{n_layers}.ln_1.weight" in w:
        n_layers += 1
    kvs = []
    for l in range(n_layers):
        p = f"{prefix}transformer.resblocks.{l}."
        h = layer_norm(y, w[p + "ln_1.weight"], w[p + "ln_1.bias"])
        q, k, v, out = encoder_attention(h, w, p, heads)
        y = y + out
        h2 = layer_norm(y, w[p + "ln_2.weight"], w[p + "ln_2.bias"])
        u = quick_gelu(F.linear(h2, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"]))
        y = y + F.linear(u, w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])
        d = {"k": k, "v": v}
        if with_q:
            d["q"] = q
        if with_out:
            d["out"] = y
        kvs.append(d)
    return kvs


def adapter_forward(w, kvs, struct_type, prefix="adapter.", drop=None):
    """`CompInvAdapter.forward` (reference `src/models.py:930-940`); `drop(site, tensor, div)` (train mode, see
    oracle/dropout_mask.py) stands for the nn.Dropout layers: p/10 (p/5 for `768-x-768`, absent in `legacy-`)
    before the second Linear and p after it, sites 1000 + 4(2i + {k: 0, v: 1}) + {0, 1}; None = eval mode.  LayerNorm-then-GELU
    structs (`:823-875`): Linear(D->x, no bias) -> LayerNorm (over (P, x) jointly for `nln`, over x
    for `ln`/`z0`) -> exact-erf GELU -> Linear(x->D, no bias), residual.  GELU-then-LayerNorm structs
    `768-x-768` (`:795-808`, output Linear at Sequential index 4) and `legacy-768-x-768` (`:809-821`,
    index 3): Linear -> GELU -> LayerNorm(x) -> Linear, residual."""
    gelu_first = struct_type in ("768-x-768", "legacy-768-x-768")
    ln_idx = 2 if gelu_first else 1
    out_idx = 3 if struct_type == "legacy-768-x-768" else 4
    out = []
    for i, kv in enumerate(kvs):
        new = {}
        for name in ("k", "v"):
            t = kv[name]
            b, tt, p, h, d = t.shape
            f = t.reshape(b, tt, p, h * d)
            a = F.linear(f, w[f"{prefix}l{i}_{name}.0.weight"])
            lw, lb = w[f"{prefix}l{i}_{name}.{ln_idx}.weight"], w[f"{prefix}l{i}_{name}.{ln_idx}.bias"]
            if gelu_first:
                a = F.layer_norm(F.gelu(a), tuple(lw.shape), lw, lb, 1e-5)
            else:
                a = F.gelu(F.layer_norm(a, tuple(lw.shape), lw, lb, 1e-5))
            site = 1000 + 4 * (2 * i + (0 if name == "k" else 1))
            if drop is not None and struct_type != "legacy-768-x-768":
                a = drop(site, a, 5 if struct_type == "768-x-768" else 10)
            a = F.linear(a, w[f"{prefix}l{i}_{name}.{out_idx}.weight"])
            if drop is not None:
                a = drop(site + 1, a)
            new[name] = t + a.reshape(b, tt, p, h, d)
        out.append(new)
    return out


def decoder_attention(q_in, k, v, m, w, p, heads, num_frames, attn_mode=()):
    """Two-branch single-query attention (reference `src/models.py:136-146`; softmax branch
    `:99-115`, CoDA branch `:117-125`).  in_proj output is viewed [B, 1, heads, 2d]: per head
    the first d channels feed the softmax branch and the next d the CoDA branch.  `m` is the
    per-key validity mask [B, S]."""
    b = q_in.shape[0]
    d = k.shape[-1]
    qs = F.linear(q_in, w[p + "attn.in_proj.weight"], w[p + "attn.in_proj.bias"]).view(b, 1, heads, 2 * d)
    q_s, q_c = qs[..., :d], qs[..., d:]
    mm = m.unsqueeze(1).unsqueeze(-1)  # [B, 1, S, 1]
    norm = math.sqrt(d)
    aff = torch.einsum("nqhc,nkhc->nqkh", q_s / norm, k).masked_fill(~mm, float("-inf"))
    if len(attn_mode) == 0:
        smax = aff.softmax(dim=-2)
    else:
        n, qq, kk, hh = aff.shape
        a5 = aff.view(n, qq, num_frames, -1, hh)
        parts = []
        if "frame" in attn_mode:
            parts.append(a5.softmax(dim=-2))
        if "temporal" in attn_mode:
            parts.append(a5.softmax(dim=-3))
        smax = sum(parts).view(n, qq, kk, hh)
    coda = torch.einsum("nqhc,nkhc->nqkh", q_c / norm, k).tanh()
    gate = -(q_c - k).abs().sum(-1).unsqueeze(1) / norm
    gate = 2 * gate.sigmoid().masked_fill(~mm, 0.0)
    aff = smax / 2 + (coda * gate) / 2
    mix = torch.einsum("nqlh,nlhc->nqhc", aff, v)
    return F.linear(mix.flatten(-2), w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])


def decoder_forward(w, kvs, m, heads, out_dims, num_frames, layer_indices, prefix="decoder.",
                    attn_mode=(), global_prediction=False, drop=None):
    """`Decoder.forward` (reference `src/models.py:323-361`, block `:173-176`, stack `:259-269`); `drop(site,
    tensor)` stands for drop_pre (site 0, `:337`), the MLP dropout of block i (site 1 + i, `:163`) and drop_post
    (site 250 + head index, `:342`) in train mode; None = eval mode.  kvs: per selected layer k, v [B, T, P, heads, d]; m [B, T] bool.
    The temporal positional embedding [T, 1, heads, d] is added to BOTH k and v."""
    g = lambda k: w[prefix + k]
    patches = kvs[0]["k"].shape[2]
    mk = m.repeat_interleave(patches, dim=-1)
    pos = w.get(prefix + "positional_embedding")
    flat = []
    for kv in kvs:
        k, v = kv["k"], kv["v"]
        if pos is not None:
            k, v = k + pos, v + pos
        flat.append((k.flatten(1, 2), v.flatten(1, 2)))
    b = flat[0][0].shape[0]
    x = g("class_embedding").view(1, 1, -1).repeat(b, 1, 1)
    x = layer_norm(x, g("ln_pre.weight"), g("ln_pre.bias"))
    if drop is not None:
        x = drop(0, x)
    results = []
    for i, (k, v) in enumerate(flat):
        p = f"{prefix}transformer.resblocks.{i}."
        h = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"])
        x = x + decoder_attention(h, k, v, mk, w, p, heads, num_frames, attn_mode)
        h2 = layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"])
        u = quick_gelu(F.linear(h2, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"]))
        if drop is not None:
            u = drop(1 + i, u)
        x = x + F.linear(u, w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])
        results.append(x)
        aq = w.get(f"{prefix}transformer.augment_query_{i}")
        if aq is not None and i != len(flat) - 1:
            x = x + aq
    x = torch.cat(results, dim=1)
    if not global_prediction:
        x = x[:, -1]
    x = layer_norm(x, g("ln_post.weight"), g("ln_post.bias"))
    if drop is not None:  # one nn.Dropout call in the reference; the product numbers the heads' slices separately
        x = drop(250, x) if x.dim() == 2 else torch.stack([drop(250 + j, x[:, j]) for j in range(x.shape[1])], dim=1)
    video_feature = x.squeeze(1)
    logits = []
    for i, od in enumerate(out_dims):
        if global_prediction:
            n = len(layer_indices)
            logits.append(sum((video_feature[:, j] @ g(f"proj{i}x{od}_L{l}")) * (j + 1) / ((1 + n) * n / 2)
                              for j, l in enumerate(layer_indices)))
        else:
            logits.append(video_feature @ g(f"proj{i}x{od}"))
    return logits, video_feature


def normalise_logits(z):
    """5 z / (||z||_2 + 1e-10) (reference `src/models.py:551-553`)."""
    return 5 * z / (torch.norm(z, dim=-1, keepdim=True) + 1e-10)


def cross_entropy_per_sample(logits, y, weight=None, label_smoothing=0.0):
    """The `auc_roc` loss factory's driver (reference `src/models.py:34-45`)."""
    wt = torch.tensor(weight) if weight else None
    return F.cross_entropy(logits, y, weight=wt, label_smoothing=label_smoothing, reduction="none")


def detector_predict(w, x, m, *, heads, patch, layer_indices, out_dims, num_frames,
                     adapter_struct=None, attn_mode=(), global_prediction=False, return_kvs=False, drop=None):
    """`Detector.predict` in eval mode (reference `src/models.py:498-566`): frozen encoder on
    the (B*T)-flattened frames, CLS row dropped, temporal axis restored, layers selected,
    optional adapter, decoder, logits rescaled to L2 norm 5."""
    b, t = x.shape[:2]
    with torch.no_grad():
        kvs = encoder_forward(w, x.flatten(0, 1), heads, patch)
        kvs = [{n: kvs[l][n][:, 1:].unflatten(0, (b, t)) for n in ("k", "v")} for l in layer_indices]
    if adapter_struct is not None:
        kvs = adapter_forward(w, kvs, adapter_struct, drop=drop)
    logits, feat = decoder_forward(w, kvs, m, heads, out_dims, num_frames, layer_indices,
                                   attn_mode=attn_mode, global_prediction=global_prediction, drop=drop)
    logits = [normalise_logits(z) for z in logits]
    if return_kvs:
        return logits, feat, kvs
    return logits, feat


def ema_frames(x, m, ratio):
    """`op_mode.ema_frame` (reference `src/models.py:572-578`): exponential moving average over the
    frames, started from zeros -> one frame per clip; the mask keeps its first column."""
    b, t, c, h, wd = x.shape
    acc = torch.zeros((b, 1, c, h, wd))
    for i in range(t):
        acc = acc * ratio + x[:, i].unsqueeze(1) * (1 - ratio)
    return acc, m[:, 0].unsqueeze(1)


def detector_forward_eval(w, x, y_list, m, single_task=None, ema_frame=0, **kw):
    """`Detector.forward(train=False)` (reference `src/models.py:568-596`): per-task
    unreduced losses (0 for unselected tasks) and logits."""
    if ema_frame:
        x, m = ema_frames(x, m, ema_frame)
        kw = dict(kw, num_frames=1)
    logits, _ = detector_predict(w, x, m, **kw)
    losses = [cross_entropy_per_sample(z, y) if (single_task is None or i == single_task) else 0
              for i, (z, y) in enumerate(zip(logits, y_list))]
    return losses, logits
