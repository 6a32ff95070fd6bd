"""Temporal cross-attention decoder on HIP kernels.

Host-side mirror of the reference's `Decoder` (reference `src/models.py:272-361`) with its
block (`:149-176`), stack (`:232-269`) and two-branch attention (`:81-146`): same parameter
names and shapes, same `forward(kvs, m) -> (task_logits, video_feature)` contract.  One
learned CLS query per clip attends to the T*P exported keys/values of one encoder layer per
block.  In `train()` mode with `config.dropout` > 0 the reference's nn.Dropout layers act as there: drop_pre
after ln_pre (`:337`), one in every block's MLP after QuickGELU (`:163`), drop_post after ln_post (`:342`);
masks are counter-based, drawn inside the kernels and regenerated in the backward (csrc/dropout.hpp).

Kernel sequence per block, rows = clips (B), everything f32 except the K/V stream:
  LayerNorm -> dfd_linear_rows(in_proj) -> dfd_decoder_attn_fwd (streams K and V once)
  -> dfd_linear_rows(out_proj, +residual) -> LayerNorm -> dfd_linear_rows(c_fc, QuickGELU)
  -> dfd_linear_rows(c_proj, +residual); then dfd_head_fwd (ln_post, projection, 5·z/‖z‖).
"""
import collections
import weakref

import logging

import torch
from torch import nn

from . import capi
from .encoder import RuntimeStateMixin, _Holder, _Mlp

# decoder -> {signature: captured graphs}; kept outside the module so that deepcopy / state_dict never see them
_GRAPHS = weakref.WeakKeyDictionary()


class _DecAttnParams(_Holder):
    def __init__(self, d):
        super().__init__()
        self.in_proj = nn.Linear(d, 2 * d)  # per head: [softmax query 64 | CoDA query 64]
        self.out_proj = nn.Linear(d, d)


class DecoderBlock(_Holder):
    def __init__(self, d):
        super().__init__()
        self.attn = _DecAttnParams(d)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = _Mlp(d)
        self.ln_2 = nn.LayerNorm(d)


class DecoderTransformer(_Holder):
    def __init__(self, width, n_blocks, aug_query):
        super().__init__()
        self.width = width
        self.augment_query_embeddings = []
        if aug_query:
            for i in range(n_blocks - 1):
                name = f"augment_query_{i}"
                setattr(self, name, nn.Parameter(torch.zeros(width)))
                self.augment_query_embeddings.append(getattr(self, name))
        self.resblocks = nn.Sequential(*[DecoderBlock(width) for _ in range(n_blocks)])


class Decoder(RuntimeStateMixin, nn.Module):
    _RUNTIME_STATE = {"_wt_cache": {}, "_after_backward": None, "_param_list": None, "_graphs_failed": None, "_mirror_ids": None}

    def invalidate_caches(self):
        """After parameters were rewritten in place behind autograd's back."""
        self._wt_cache = {}
        _GRAPHS.pop(self, None)

    def __init__(self, detector, config, num_frames):
        super().__init__()
        enc = detector.encoder
        width, heads = enc.width, enc.heads
        self.width, self.heads, self.num_frames = width, heads, num_frames
        self.op_mode = config.op_mode
        self.dropout_p = float(config.dropout) if "dropout" in config else 0.0
        self.out_dims = list(config.out_dim)
        self.layer_indices = list(detector.layer_indices)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        if "temporal_position" in config.op_mode and config.op_mode.temporal_position:
            self.positional_embedding = nn.Parameter(scale * torch.randn(num_frames, 1, heads, width // heads))
        else:
            self.positional_embedding = None
        # attn_mode "frame", "temporal" or "frame+temporal": grouped softmaxes in the softmax branch (models.py:97, :107-115)
        self.attn_modes = 0
        if "attn_mode" in config.op_mode and config.op_mode.attn_mode:
            for name in config.op_mode.attn_mode.split("+"):
                self.attn_modes |= capi.ATTN_MODE_BITS.get(name, 0)  # unknown names add nothing, as in the reference
            if self.attn_modes == 0:
                raise ValueError(f"op_mode.attn_mode={config.op_mode.attn_mode!r}: the reference sums an empty list here")
        self.global_prediction = bool("global_prediction" in config.op_mode and config.op_mode.global_prediction)
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = DecoderTransformer(width, len(self.layer_indices),
                                              bool("aug_query" in config.op_mode and config.op_mode.aug_query))
        self.ln_post = nn.LayerNorm(width)
        self.task_projections = []
        for i, od in enumerate(self.out_dims):
            # one projection per task, or one per tapped layer with `global_prediction` (models.py:306-321)
            names = [f"proj{i}x{od}_L{l}" for l in self.layer_indices] if self.global_prediction else [f"proj{i}x{od}"]
            for name in names:
                setattr(self, name, nn.Parameter(scale * torch.randn(width, od)))
            self.task_projections.append([getattr(self, name) for name in names])
        self._wt_cache = {}  # name -> (parameter version, transposed f32 copy) for the row-streaming linear kernel
        self._mirror_ids = None
        self._param_list = None
        # Opt-in (Detector.static_graphs): the ~450 small launches of one training step of the decoder are
        # captured into two HIP graphs (forward kernels, backward kernels) per input signature and replayed,
        # so the host enqueues a step in a few milliseconds however loaded its cores are.  Needs K/V at
        # stable addresses (Detector keeps persistent export buffers in this mode).
        self.use_graphs = False
        self._graphs_failed = None  # set when the runtime refused a capture: eager launches from then on
        self._after_backward = None  # one-shot callback for the next autograd node (Detector's encoder pipelining)
        # decoder blocks start from the encoder layer they read (models.py:226-229)
        for b, l in enumerate(self.layer_indices):
            src, dst = enc.transformer.resblocks[l], self.transformer.resblocks[b]
            dst.ln_1.load_state_dict(src.ln_1.state_dict())
            dst.ln_2.load_state_dict(src.ln_2.state_dict())
            dst.mlp.load_state_dict(src.mlp.state_dict())

    def temporal_pos(self):
        """[T, D] f32 view of the temporal positional embedding, or None."""
        if self.positional_embedding is None:
            return None
        return self.positional_embedding.detach().reshape(self.num_frames, self.width).to(torch.float32).contiguous()

    def _pack(self, kvs):
        """Reference-layout input: list of {k, v: [B, T, P, heads, 64]} -> ([L, B*S, D], same),
        positional embedding added (models.py:326-334).  Glue for callers that hand over
        reference-style kvs; `Detector.predict` exports the packed form directly."""
        ks, vs = [], []
        for kv in kvs:
            k, v = kv["k"], kv["v"]
            if self.positional_embedding is not None:
                pos = self.positional_embedding.detach().to(k.dtype)
                k, v = k + pos, v + pos
            b, t, p, h, d = k.shape
            ks.append(k.reshape(b * t * p, h * d))
            vs.append(v.reshape(b * t * p, h * d))
        return torch.stack(ks).contiguous(), torch.stack(vs).contiguous()

    def forward(self, kvs, m):
        """kvs: packed (k, v) tensors [L, B*T*P, D] with the positional embedding already added
        (the encoder's export), or the reference's list of dicts.  m: [B, T] bool.
        Returns (task_logits list of [B, out_dim] — NOT yet rescaled, as in the reference — and
        video_feature [B, D]).  Differentiable w.r.t. the decoder parameters when grad is enabled."""
        raw, feat, _ = self.run(kvs, m)
        return raw, feat

    # ---- plumbing ------------------------------------------------------------------------------
    def _unpack(self, kvs, m):
        """-> (k_all, v_all, kv_pos, mask, B, T, P).  K/V come as the dense export [L, B*T*P, D] (positional embedding
        already added; kv_pos None), or IN PLACE: (k, v, pos) with k, v strided views [L, B*T, P, D] of the encoder's
        q|k|v activations and pos [T, D] f32 (or None) added by the attention kernels as they read the rows."""
        kv_pos = None
        if isinstance(kvs, list):
            k_all, v_all = self._pack(kvs)
        elif len(kvs) == 3:
            k_all, v_all, kv_pos = kvs
        else:
            k_all, v_all = kvs
        if not k_all.is_cuda:
            raise capi.DfdError("the decoder runs on HIP kernels only: pass device tensors")
        B, T = m.shape
        if k_all.dim() == 4:
            assert k_all.shape[1] == B * T and v_all.shape == k_all.shape and v_all.stride() == k_all.stride()
            P = k_all.shape[2]
        else:
            S = k_all.shape[1] // B
            assert S % T == 0
            P = S // T
        assert k_all.shape[0] == len(self.layer_indices) and k_all.shape[-1] == self.width
        mask = m.to(device=k_all.device, dtype=torch.uint8).contiguous()
        return k_all, v_all, kv_pos, mask, B, T, P

    def _wt(self, name, w):
        """Transposed [K, N] copy of Linear weight `name`, refreshed when the parameter changed (its
        autograd version counter moves on optimizer.step / load_state_dict / .to()).  The copy keeps its ADDRESS across
        refreshes (captured graphs read it), and `FusedSGD` rewrites it inside its own launch (`mirrors_written`)."""
        p = w[name]
        key = (p._version, p.data_ptr(), p.device)
        hit = self._wt_cache.get(name)
        if hit is None or hit[0] != key:
            src = p.to(torch.float32).contiguous()
            dst = hit[1] if (hit is not None and hit[1].shape == (src.shape[1], src.shape[0]) and hit[1].device == src.device) else \
                torch.empty(src.shape[1], src.shape[0], device=src.device, dtype=torch.float32)
            capi.transpose(src, dst)
            self._wt_cache[name] = hit = (key, dst)
        return hit[1]

    # ---- transposed copies kept by the optimizer (optim.FusedSGD) -------------------------------------------------
    _MIRRORED = ("attn.in_proj.weight", "attn.out_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")

    def _mirror_names(self):
        if self._mirror_ids is None:  # the module tree is static
            self._mirror_ids = {id(p): n for n, p in self.named_parameters() if n.endswith(self._MIRRORED) and p.dim() == 2}
        return self._mirror_ids

    def current_mirror(self, p):
        """The transposed copy `_wt` would hand out for `p` right now (no refresh), or None."""
        hit = self._wt_cache.get(self._mirror_names().get(id(p)))
        return None if hit is None else hit[1]

    def mirror_for(self, p):
        """The transposed copy of Linear weight `p` that `_wt` hands to the kernels, created (and filled) on demand; None for
        parameters the forward does not read transposed."""
        name = self._mirror_names().get(id(p))
        if name is None or not p.is_cuda:
            return None
        return self._wt(name, {name: p.detach()})

    def mirrors_written(self, params):
        """`FusedSGD` has just rewritten these parameters AND their transposed copies: the copies are current."""
        names = self._mirror_names()
        for p, mirror in params:
            n = names.get(id(p))
            hit = self._wt_cache.get(n)
            if hit is not None and hit[1] is mirror:  # (a copy replaced since the optimizer planned is refreshed by `_wt` instead)
                self._wt_cache[n] = ((p._version, p.data_ptr(), p.device), hit[1])

    def _refresh_mirrors(self, w):
        """Before a captured forward is replayed: the graph reads the transposed copies at fixed addresses and contains no
        transpose, so whatever changed the weights without maintaining them (another optimizer, a checkpoint load) is caught
        up with here, eagerly."""
        for n in w:
            if n.endswith(self._MIRRORED) and w[n].dim() == 2:
                self._wt(n, w)

    def _lin_ws(self, B, dev):
        D = self.width
        shapes = [(2 * D, D), (D, 2 * D), (D, D), (4 * D, D), (D, 4 * D)]
        nbytes = max(capi.linear_rows_t_workspace_bytes(B, n, k) for n, k in shapes)
        return torch.empty(nbytes // 4, device=dev, dtype=torch.float32)

    def _splits(self, B, S):
        # enough workgroups to fill 256 CUs a few times over, few enough partial states to merge cheaply
        return max(1, min(S // 64, max(1, 768 // max(B, 1))))

    def _drop(self, drop_rng, site):
        """Dropout descriptor of one site of this step (0 = drop_pre, 1 + i = block i's MLP, 250 + j = drop_post of
        head j), or None."""
        if drop_rng is None or self.dropout_p <= 0:
            return None
        return capi.Dropout(drop_rng, site, self.dropout_p)

    def run(self, kvs, m, drop_rng=None):
        """-> (raw logits list, video_feature, normalised logits list).  With grad enabled and any
        trainable parameter, goes through `_DecoderFn` so `loss.backward()` reaches the parameters.
        `drop_rng`: this step's dropout state (device int64 {seed, step}) in train mode, None = no dropout."""
        k_all, v_all, kv_pos, mask, B, T, P = self._unpack(kvs, m)
        if self.dropout_p <= 0:
            drop_rng = None
        if self._param_list is None:  # the module tree is static: walk it once, not every step
            self._param_list = list(self.named_parameters())
        names = [n for n, p in self._param_list]
        params = [p for n, p in self._param_list]
        if torch.is_grad_enabled() and (any(p.requires_grad for p in params) or k_all.requires_grad):
            out = _DecoderFn.apply(self, k_all, v_all, mask, (B, T, P), names, (drop_rng, kv_pos), *params)
            n = len(self.out_dims)
            return list(out[1:1 + n]), out[0], list(out[1 + n:1 + 2 * n])
        w = {n: p.detach() for n, p in zip(names, params)}
        raws, feat, outs, _ = self._forward_kernels(w, k_all, v_all, mask, B, T, P, save=False, drop_rng=drop_rng, kv_pos=kv_pos)
        return raws, feat, outs

    # ---- forward on HIP kernels --------------------------------------------------------------
    def _forward_kernels(self, w, k_all, v_all, mask, B, T, P, save, drop_rng=None, kv_pos=None):
        dev = k_all.device
        D, H, L = self.width, self.heads, k_all.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        new = lambda *shape: torch.empty(*shape, **f32)
        g = lambda name: w[name].to(torch.float32).contiguous()
        splits = self._splits(B, T * P)
        ws = new(capi.decoder_attn_workspace_bytes(B, H, 64, splits) // 4)
        lws = self._lin_ws(B, dev)

        def lin(x_, pre_, y_, epi=capi.EPI_BIAS, residual=None):
            capi.linear_rows_t(x_, self._wt(pre_ + "weight", w), g(pre_ + "bias"), y_, lws, epi, residual=residual)

        x0 = g("class_embedding").view(1, D).repeat(B, 1).contiguous()
        x = new(B, D)
        capi.layernorm(x0, g("ln_pre.weight"), g("ln_pre.bias"), x)
        if self._drop(drop_rng, 0) is not None:
            capi.dropout(x, x, self._drop(drop_rng, 0))  # drop_pre (models.py:337)
        saved = dict(x0=x0, blocks=[], drop_rng=drop_rng, kv_pos=kv_pos)
        xs = []  # per-block outputs, read by the per-layer heads of `global_prediction`
        mode_ws = None
        h, q, mix, stats, u = new(B, D), new(B, 2 * D), new(B, D), new(B, H, 2), new(B, 4 * D)
        for i in range(L):
            pre = f"transformer.resblocks.{i}."
            if save:  # keep every intermediate of the block for the backward pass
                h1, q, mix, mix_s, stats = new(B, D), new(B, 2 * D), new(B, D), new(B, D), new(B, H, 2)
                x_in = x
                capi.layernorm(x_in, g(pre + "ln_1.weight"), g(pre + "ln_1.bias"), h1)
                lin(h1, pre + "attn.in_proj.", q)
                sc = aw = None
                if self.attn_modes:
                    sc, aw = new(B, H, T * P), new(B, H, T * P)
                    capi.decoder_attn_modes_fwd(q, k_all[i], mask, self.attn_modes, sc, aw, B, T, P, H, pos=kv_pos)
                capi.decoder_attn_fwd(q, k_all[i], v_all[i], mask, mix, stats, ws, splits, B, T, P, H, mix_softmax=mix_s,
                                      ext_weights=aw, pos=kv_pos)
                x_mid = new(B, D)  # x_in stays for the backward pass: the residual is a separate input, not a clone
                lin(mix, pre + "attn.out_proj.", x_mid, capi.EPI_BIAS_RESIDUAL, residual=x_in)
                h2, u_pre, uu = new(B, D), new(B, 4 * D), new(B, 4 * D)
                capi.layernorm(x_mid, g(pre + "ln_2.weight"), g(pre + "ln_2.bias"), h2)
                lin(h2, pre + "mlp.c_fc.", u_pre)
                capi.quickgelu(u_pre, uu, drop=self._drop(drop_rng, 1 + i))
                x = new(B, D)
                lin(uu, pre + "mlp.c_proj.", x, capi.EPI_BIAS_RESIDUAL, residual=x_mid)
                saved["blocks"].append(dict(x_in=x_in, h1=h1, q=q, mix=mix, mix_s=mix_s, stats=stats, x_mid=x_mid, h2=h2,
                                            u_pre=u_pre, u=uu, sc=sc, aw=aw))
            else:
                capi.layernorm(x, g(pre + "ln_1.weight"), g(pre + "ln_1.bias"), h)
                lin(h, pre + "attn.in_proj.", q)
                aw = None
                if self.attn_modes:
                    if mode_ws is None:
                        mode_ws = (new(B, H, T * P), new(B, H, T * P))
                    aw = capi.decoder_attn_modes_fwd(q, k_all[i], mask, self.attn_modes, mode_ws[0], mode_ws[1], B, T, P, H, pos=kv_pos)
                capi.decoder_attn_fwd(q, k_all[i], v_all[i], mask, mix, stats, ws, splits, B, T, P, H, ext_weights=aw, pos=kv_pos)
                lin(mix, pre + "attn.out_proj.", x, capi.EPI_BIAS_RESIDUAL)
                capi.layernorm(x, g(pre + "ln_2.weight"), g(pre + "ln_2.bias"), h)
                lin(h, pre + "mlp.c_fc.", u, capi.EPI_BIAS_QUICKGELU)
                if self._drop(drop_rng, 1 + i) is not None:  # train() under no_grad: same masks as the autograd path
                    capi.dropout(u, u, self._drop(drop_rng, 1 + i))
                lin(u, pre + "mlp.c_proj.", x, capi.EPI_BIAS_RESIDUAL)
            if self.global_prediction:
                xs.append(x if save else x.clone())
            aq = f"transformer.augment_query_{i}"
            if aq in w and i != L - 1:
                # result.append(x) precedes the add in the reference (models.py:263-267); only the last
                # block's x is read when there is one projection per task (models.py:340-341)
                x = x + g(aq)
        raws, outs = [], []
        if self.global_prediction:
            # every block's output goes through ln_post and its own projection; the logits are the
            # (j+1)/(L(L+1)/2)-weighted sum (models.py:345-357).  video_feature is [B, L, D].
            feat = new(B, L, D)
            cw = [(j + 1) / ((1 + L) * L / 2) for j in range(L)]
            for i, od in enumerate(self.out_dims):
                z = torch.zeros(B, od, **f32)
                for j, l in enumerate(self.layer_indices):
                    fj, rj, tmp = new(B, D), new(B, od), new(B, od)
                    capi.head_fwd(xs[j], g("ln_post.weight"), g("ln_post.bias"), g(f"proj{i}x{od}_L{l}"), fj, rj, tmp,
                                  drop=self._drop(drop_rng, 250 + j))
                    feat[:, j] = fj
                    z += cw[j] * rj
                raws.append(z)
                outs.append(5.0 * z / (z.norm(dim=-1, keepdim=True) + 1e-10))  # models.py:551-553
            saved["xs"] = xs
        else:
            feat = new(B, D)
            for i, od in enumerate(self.out_dims):
                raw, logits = new(B, od), new(B, od)
                capi.head_fwd(x, g("ln_post.weight"), g("ln_post.bias"), g(f"proj{i}x{od}"), feat, raw, logits,
                              drop=self._drop(drop_rng, 250))
                raws.append(raw)
                outs.append(logits)
        saved["x_last"] = x
        saved["feat"] = feat
        saved["raws"] = raws
        return raws, feat, outs, saved

    # ---- backward on HIP kernels -------------------------------------------------------------
    def _backward_kernels(self, w, saved, k_all, v_all, mask, B, T, P, d_feat, d_raws, d_logits, want_dkv=False):
        """Gradients of every decoder parameter (dict name -> tensor), given dL/d(video_feature),
        dL/d(raw logits) and dL/d(normalised logits) (any may be None).  With `want_dkv` the full
        key/value gradients [L, B*S, D] (in the K/V dtype) come back under "__dk" / "__dv"."""
        dev = k_all.device
        D, H, L = self.width, self.heads, k_all.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        new = lambda *shape: torch.empty(*shape, **f32)
        g = lambda name: w[name].to(torch.float32).contiguous()
        grads = {}
        xhat = new(B, D)
        lws = self._lin_ws(B, dev)
        drop_rng = saved.get("drop_rng")
        kv_pos = saved.get("kv_pos")

        def lin_bwd(pre, dy, x_act, want_dx=True):
            """dy [B,N] -> grads of Linear `pre` (weight [N,K], bias) and, optionally, dx [B,K]."""
            W = g(pre + "weight")
            N, K = W.shape
            dW, db = new(N, K), new(N)
            capi.linear_rows_bwd_weight(dy, x_act, dW, db)
            grads[pre + "weight"], grads[pre + "bias"] = dW, db
            if not want_dx:
                return None
            dx = new(B, K)
            capi.linear_rows_t(dy, W, None, dx, lws)  # dx = dy @ W: the weight [N, K] is already "K-major" for this product
            return dx

        # ---- head: logits = 5 z/(|z|+eps), z = feat @ proj, feat = ln_post(x_last)
        feat, x_last = saved["feat"], saved["x_last"]
        head_dx = None  # global_prediction: gradient entering each block's output from its own head
        if self.global_prediction:
            # z = Σ_j c_j feat_j @ proj_j.  head_bwd is linear in dlogits, so calling it with c_j·dlogits on the
            # summed z yields dz_j = c_j dz, dproj_j = feat_jᵀ dz_j and dfeat_j = dz_j proj_jᵀ (+ external dfeat_j)
            cw = [(j + 1) / ((1 + L) * L / 2) for j in range(L)]
            dfeat_layers = [d_feat[:, j].contiguous().clone() if d_feat is not None else None for j in range(L)]
            for i, od in enumerate(self.out_dims):
                z = saved["raws"][i]
                dl = d_logits[i].contiguous() if d_logits[i] is not None else torch.zeros(B, od, **f32)
                for j, l in enumerate(self.layer_indices):
                    name = f"proj{i}x{od}_L{l}"
                    proj, fj = g(name), feat[:, j].contiguous()
                    dz, dfj, dproj = new(B, od), new(B, D), new(D, od)
                    capi.head_bwd(z, cw[j] * dl, proj, fj, dfeat_layers[j], dz, dfj, dproj)
                    if d_raws[i] is not None:  # raw logits consumed directly (rare): plain glue
                        dproj = dproj + fj.t().contiguous() @ (cw[j] * d_raws[i])
                        dfj = dfj + (cw[j] * d_raws[i]) @ proj.t()
                    grads[name] = dproj
                    dfeat_layers[j] = dfj
            head_dx = []
            dgam_t, dbet_t = torch.zeros(D, **f32), torch.zeros(D, **f32)
            for j in range(L):
                dxj, dgam, dbet = new(B, D), new(D), new(D)
                dfj = dfeat_layers[j] if dfeat_layers[j] is not None else torch.zeros(B, D, **f32)
                if self._drop(drop_rng, 250 + j) is not None:  # video_feature_j = drop_post(ln_post(x_j))
                    capi.dropout(dfj, dfj, self._drop(drop_rng, 250 + j))
                capi.layernorm_bwd(saved["xs"][j], g("ln_post.weight"), dfj, dxj, dgam, dbet, xhat)
                head_dx.append(dxj)
                dgam_t += dgam
                dbet_t += dbet
            grads["ln_post.weight"], grads["ln_post.bias"] = dgam_t, dbet_t
            dx = head_dx[L - 1]
        else:
            dfeat = d_feat.contiguous().clone() if d_feat is not None else None
            for i, od in enumerate(self.out_dims):
                raw = saved["raws"][i]
                proj = g(f"proj{i}x{od}")
                dz, dfi, dproj = new(B, od), new(B, D), new(D, od)
                dl = d_logits[i]
                if dl is None:  # gradient only through the raw (un-normalised) logits, or none at all
                    dz = d_raws[i].contiguous() if d_raws[i] is not None else torch.zeros(B, od, **f32)
                    capi.head_bwd(raw, torch.zeros(B, od, **f32), proj, feat, dfeat, new(B, od), dfi, dproj)
                    dproj = feat.t().contiguous() @ dz  # rare path (raw logits consumed directly): plain glue
                    dfi = dfi + dz @ proj.t()
                else:
                    capi.head_bwd(raw, dl.contiguous(), proj, feat, dfeat, dz, dfi, dproj)
                    if d_raws[i] is not None:
                        dproj = dproj + feat.t().contiguous() @ d_raws[i]
                        dfi = dfi + d_raws[i] @ proj.t()
                grads[f"proj{i}x{od}"] = dproj
                dfeat = dfi
            if dfeat is None:
                dfeat = torch.zeros(B, D, **f32)
            if self._drop(drop_rng, 250) is not None:  # video_feature = drop_post(ln_post(x))
                capi.dropout(dfeat, dfeat, self._drop(drop_rng, 250))
            dx = new(B, D)
            dgam, dbet = new(D), new(D)
            capi.layernorm_bwd(x_last, g("ln_post.weight"), dfeat, dx, dgam, dbet, xhat)
            grads["ln_post.weight"], grads["ln_post.bias"] = dgam, dbet

        ws = new(capi.decoder_attn_bwd_workspace_bytes(B, T, H) // 4)
        has_pos = "positional_embedding" in w
        dpos_total = torch.zeros(T, D, **f32) if has_pos else None
        assert not (want_dkv and k_all.dim() == 4), "K/V gradients are produced for the dense export only"
        dk_all = torch.empty_like(k_all) if want_dkv else None
        dv_all = torch.empty_like(v_all) if want_dkv else None
        for i in reversed(range(L)):
            pre = f"transformer.resblocks.{i}."
            sv = saved["blocks"][i]
            aq = f"transformer.augment_query_{i}"
            if aq in w and i != L - 1:
                grads[aq] = dx.sum(dim=0)  # x_{i+1,in} = x_{i,out} + aq_i
            if head_dx is not None and i != L - 1:
                dx = dx + head_dx[i]  # block i's output also feeds its own head
            # x_out = x_mid + c_proj(u)
            du = lin_bwd(pre + "mlp.c_proj.", dx, sv["u"])
            du_pre = new(B, 4 * D)
            capi.quickgelu(sv["u_pre"], du_pre, du=du, drop=self._drop(drop_rng, 1 + i))
            dh2 = lin_bwd(pre + "mlp.c_fc.", du_pre, sv["h2"])
            dgam, dbet = new(D), new(D)
            capi.layernorm_bwd(sv["x_mid"], g(pre + "ln_2.weight"), dh2, dx, dgam, dbet, xhat, accumulate_dx=True)
            grads[pre + "ln_2.weight"], grads[pre + "ln_2.bias"] = dgam, dbet
            # x_mid = x_in + out_proj(mix)
            dmix = lin_bwd(pre + "attn.out_proj.", dx, sv["mix"])
            dq = new(B, 2 * D)
            dpos = new(T, D) if has_pos else None
            dsc = None
            if self.attn_modes:
                dsc = new(B, H, T * P)
                capi.decoder_attn_modes_bwd(sv["sc"], v_all[i], dmix, self.attn_modes, new(B, H, T * P), dsc, B, T, P, H, pos=kv_pos)
            capi.decoder_attn_bwd(sv["q"], k_all[i], v_all[i], mask, dmix, sv["mix_s"], sv["stats"], dq, dpos, ws, B, T, P, H,
                                  dk=dk_all[i] if want_dkv else None, dv=dv_all[i] if want_dkv else None,
                                  ext_weights=sv["aw"], ext_dscores=dsc, pos=kv_pos)
            if has_pos:
                dpos_total += dpos
            dh1 = lin_bwd(pre + "attn.in_proj.", dq, sv["h1"])
            dgam, dbet = new(D), new(D)
            capi.layernorm_bwd(sv["x_in"], g(pre + "ln_1.weight"), dh1, dx, dgam, dbet, xhat, accumulate_dx=True)
            grads[pre + "ln_1.weight"], grads[pre + "ln_1.bias"] = dgam, dbet
        # x_0 = drop_pre(ln_pre(class_embedding)) for every clip
        if self._drop(drop_rng, 0) is not None:
            capi.dropout(dx, dx, self._drop(drop_rng, 0))
        dcls_rows = new(B, D)
        dgam, dbet = new(D), new(D)
        capi.layernorm_bwd(saved["x0"], g("ln_pre.weight"), dx, dcls_rows, dgam, dbet, xhat)
        grads["ln_pre.weight"], grads["ln_pre.bias"] = dgam, dbet
        grads["class_embedding"] = dcls_rows.sum(dim=0)
        if has_pos:
            grads["positional_embedding"] = dpos_total.view(T, 1, H, D // H)
        if want_dkv:
            grads["__dk"], grads["__dv"] = dk_all, dv_all
        return grads


    # ---- HIP-graph replay of the training-step kernel sequences --------------------------------
    max_graphs = 8

    def drop_graphs_for(self, data_ptrs):
        """Forget the captured graphs that read K/V buffers at these addresses (the buffers are being released)."""
        graphs = _GRAPHS.get(self)
        if not graphs:
            return
        stale = [k for k in graphs if k[0] in data_ptrs or k[1] in data_ptrs]
        if stale:
            torch.cuda.synchronize()
            for k in stale:
                del graphs[k]

    def _graph_key(self, k_all, v_all, mask, dims, params, dropping, kv_pos=None):
        return (k_all.data_ptr(), v_all.data_ptr(), str(k_all.dtype), tuple(k_all.shape), tuple(k_all.stride()), dims, tuple(mask.shape),
                None if kv_pos is None else kv_pos.data_ptr(),
                tuple(p.data_ptr() for p in params), self.attn_modes, self.global_prediction, dropping, self.dropout_p)

    def _graph_forward(self, w, k_all, v_all, mask, dims, params, drop_rng=None, kv_pos=None):
        B, T, P = dims
        if self._graphs_failed:
            return None
        key = self._graph_key(k_all, v_all, mask, dims, params, drop_rng is not None, kv_pos)
        graphs = _GRAPHS.setdefault(self, collections.OrderedDict())
        ent = graphs.get(key)
        if ent is not None:
            graphs.move_to_end(key)
        if ent is None:
            # least recently used out: a training run cycles through a handful of signatures (full and last batch of an
            # epoch, two K/V sets each when the encoder is pipelined); an entry whose K/V buffers were released never comes
            # back and must not pin the cache (round 2 capped it at four and then stayed eager for good)
            while len(graphs) >= self.max_graphs:
                torch.cuda.synchronize()  # a replay of the entry may still be executing
                graphs.popitem(last=False)
            # the dropout state is read from device memory by the kernels: the graph owns a static copy that is
            # refreshed before every replay, so each step draws new masks and its backward regenerates them
            ent = dict(mask=mask.clone(), bwd={}, rng=None if drop_rng is None else drop_rng.clone())
            self._forward_kernels(w, k_all, v_all, ent["mask"], B, T, P, save=True, drop_rng=ent["rng"], kv_pos=kv_pos)  # eager once: lazy initialisations
            # the transposed weight copies are NOT nodes of the graph: they sit at fixed addresses, `FusedSGD` rewrites them
            # with the weights, and `_refresh_mirrors` (below, before every replay) catches up after anything else
            self._refresh_mirrors(w)
            torch.cuda.synchronize()
            try:
                # thread_local: other threads of the process (the RCCL watchdog, a data loader) may keep calling into
                # the runtime while this thread captures
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    raws, feat, outs, saved = self._forward_kernels(w, k_all, v_all, ent["mask"], B, T, P, save=True, drop_rng=ent["rng"], kv_pos=kv_pos)
            except Exception as e:  # capture is an optimisation: a runtime that refuses it leaves the eager launches
                self._graphs_failed = f"{type(e).__name__}: {e}"
                logging.warning("decoder: HIP graph capture failed, staying on eager launches (%s)", self._graphs_failed)
                torch.cuda.synchronize()
                return None
            ent.update(fwd=g, raws=raws, feat=feat, outs=outs, saved=saved)
            graphs[key] = ent
        ent["mask"].copy_(mask)
        if drop_rng is not None:
            ent["rng"].copy_(drop_rng)
        self._refresh_mirrors(w)
        ent["fwd"].replay()
        return ent

    def _graph_backward(self, ent, w, k_all, v_all, dims, d_feat, d_raws, d_logits, want_dkv):
        B, T, P = dims
        sig = (d_feat is not None, tuple(t is not None for t in d_raws), tuple(t is not None for t in d_logits), want_dkv)
        b = ent["bwd"].get(sig)
        if b is None:
            b = dict(d_feat=None if d_feat is None else d_feat.contiguous().clone(),
                     d_raws=[None if t is None else t.contiguous().clone() for t in d_raws],
                     d_logits=[None if t is None else t.contiguous().clone() for t in d_logits])
            self._backward_kernels(w, ent["saved"], k_all, v_all, ent["mask"], B, T, P, b["d_feat"], b["d_raws"], b["d_logits"],
                                   want_dkv)  # eager once
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    grads = self._backward_kernels(w, ent["saved"], k_all, v_all, ent["mask"], B, T, P, b["d_feat"], b["d_raws"],
                                                   b["d_logits"], want_dkv)
            except Exception as e:  # as in the forward: fall back to the eager kernels on the forward graph's saved tensors
                self._graphs_failed = f"{type(e).__name__}: {e}"
                logging.warning("decoder: HIP graph capture of the backward failed, staying on eager launches (%s)", self._graphs_failed)
                torch.cuda.synchronize()
                return self._backward_kernels(w, ent["saved"], k_all, v_all, ent["mask"], B, T, P, d_feat, d_raws, d_logits, want_dkv)
            b.update(graph=g, grads=grads)
            ent["bwd"][sig] = b
        if d_feat is not None:
            b["d_feat"].copy_(d_feat)
        for dst, src in zip(b["d_raws"] + b["d_logits"], list(d_raws) + list(d_logits)):
            if src is not None:
                dst.copy_(src)
        b["graph"].replay()
        # parameter gradients leave as copies (autograd may keep or accumulate into what it is handed), made by ONE
        # multi-tensor copy instead of a copy per parameter; the K/V gradients (adapter training, 1.7 GB each) are
        # consumed immediately and stay in place
        names = [k for k in b["grads"] if not k.startswith("__")]
        src = [b["grads"][k] for k in names]
        dst = [torch.empty_like(t) for t in src]
        torch._foreach_copy_(dst, src)
        out = dict(zip(names, dst))
        out.update({k: v for k, v in b["grads"].items() if k.startswith("__")})
        return out


class _DecoderFn(torch.autograd.Function):
    """autograd node around the decoder's HIP forward / backward kernels.
    Outputs: (video_feature, *raw_logits, *normalised_logits)."""

    @staticmethod
    def forward(ctx, dec, k_all, v_all, mask, dims, names, extra, *params):
        B, T, P = dims
        drop_rng, kv_pos = extra
        w = {n: p.detach() for n, p in zip(names, params)}
        ent = dec._graph_forward(w, k_all, v_all, mask, dims, params, drop_rng, kv_pos) if dec.use_graphs else None
        if ent is not None:  # replayed graph: outputs live in the graph's static buffers, hand out copies
            raws, outs = [t.clone() for t in ent["raws"]], [t.clone() for t in ent["outs"]]
            feat, saved = ent["feat"].clone(), ent["saved"]
        else:
            raws, feat, outs, saved = dec._forward_kernels(w, k_all, v_all, mask, B, T, P, save=True, drop_rng=drop_rng, kv_pos=kv_pos)
        ctx.graph_entry = ent
        ctx.after_backward, dec._after_backward = dec._after_backward, None
        ctx.dec, ctx.w, ctx.saved, ctx.dims, ctx.names = dec, w, saved, dims, names
        ctx.kv = (k_all, v_all, mask)
        ctx.n_out = len(raws)
        ctx.req = [p.requires_grad for p in params]
        return (feat, *raws, *outs)

    @staticmethod
    def backward(ctx, d_feat, *d_rest):
        n = ctx.n_out
        d_raws, d_logits = list(d_rest[:n]), list(d_rest[n:2 * n])
        k_all, v_all, mask = ctx.kv
        B, T, P = ctx.dims
        want_dkv = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if ctx.graph_entry is not None:
            grads = ctx.dec._graph_backward(ctx.graph_entry, ctx.w, k_all, v_all, ctx.dims, d_feat, d_raws, d_logits, want_dkv)
        else:
            grads = ctx.dec._backward_kernels(ctx.w, ctx.saved, k_all, v_all, mask, B, T, P, d_feat, d_raws, d_logits, want_dkv)
        out = [grads.get(nm) if rq else None for nm, rq in zip(ctx.names, ctx.req)]
        if ctx.after_backward is not None:
            ctx.after_backward()
        return (None, grads.get("__dk"), grads.get("__dv"), None, None, None, None, *out)
