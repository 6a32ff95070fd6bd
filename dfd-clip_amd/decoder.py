"""Temporal cross-attention decoder on HIP kernels.

Host-side mirror of the reference's `Decoder` (reference `src/models.py:272-361`) with its
block (`:149-176`), stack (`:232-269`) and two-branch attention (`:81-146`): same parameter
names and shapes, same `forward(kvs, m) -> (task_logits, video_feature)` contract.  One
learned CLS query per clip attends to the T*P exported keys/values of one encoder layer per
block.  Eval-mode semantics (dropout = identity).

Kernel sequence per block, rows = clips (B), everything f32 except the K/V stream:
  LayerNorm -> dfd_linear_rows(in_proj) -> dfd_decoder_attn_fwd (streams K and V once)
  -> dfd_linear_rows(out_proj, +residual) -> LayerNorm -> dfd_linear_rows(c_fc, QuickGELU)
  -> dfd_linear_rows(c_proj, +residual); then dfd_head_fwd (ln_post, projection, 5·z/‖z‖).
"""
import torch
from torch import nn

from . import capi
from .encoder import _Holder, _Mlp


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


class Decoder(nn.Module):
    def __init__(self, detector, config, num_frames):
        super().__init__()
        enc = detector.encoder
        width, heads = enc.width, enc.heads
        self.width, self.heads, self.num_frames = width, heads, num_frames
        self.op_mode = config.op_mode
        self.out_dims = list(config.out_dim)
        self.layer_indices = list(detector.layer_indices)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        if "temporal_position" in config.op_mode and config.op_mode.temporal_position:
            self.positional_embedding = nn.Parameter(scale * torch.randn(num_frames, 1, heads, width // heads))
        else:
            self.positional_embedding = None
        if "attn_mode" in config.op_mode and config.op_mode.attn_mode:
            raise NotImplementedError("op_mode.attn_mode (frame/temporal softmax factorisation) is not built yet")
        self.global_prediction = bool("global_prediction" in config.op_mode and config.op_mode.global_prediction)
        if self.global_prediction:
            raise NotImplementedError("op_mode.global_prediction is not built yet")
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = DecoderTransformer(width, len(self.layer_indices),
                                              bool("aug_query" in config.op_mode and config.op_mode.aug_query))
        self.ln_post = nn.LayerNorm(width)
        self.task_projections = []
        for i, od in enumerate(self.out_dims):
            name = f"proj{i}x{od}"
            setattr(self, name, nn.Parameter(scale * torch.randn(width, od)))
            self.task_projections.append([getattr(self, name)])
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

    @torch.no_grad()
    def forward(self, kvs, m):
        """kvs: packed (k, v) tensors [L, B*T*P, D] with the positional embedding already added
        (the encoder's export), or the reference's list of dicts.  m: [B, T] bool.
        Returns (task_logits list of [B, out_dim] — NOT yet rescaled, as in the reference — and
        video_feature [B, D])."""
        raw, feat, _ = self._forward_impl(kvs, m)
        return raw, feat

    def _forward_impl(self, kvs, m):
        if isinstance(kvs, (list,)):
            k_all, v_all = self._pack(kvs)
        else:
            k_all, v_all = kvs
        if not k_all.is_cuda:
            raise capi.DfdError("the decoder runs on HIP kernels only: pass device tensors")
        dev = k_all.device
        B, T = m.shape
        D, H = self.width, self.heads
        L = k_all.shape[0]
        S = k_all.shape[1] // B
        P = S // T
        assert L == len(self.layer_indices) and S == T * P and k_all.shape[2] == D
        mask = m.to(device=dev, dtype=torch.uint8).contiguous()
        f32 = dict(device=dev, dtype=torch.float32)
        # enough workgroups to fill 256 CUs a few times over, few enough partial states to merge cheaply
        splits = max(1, min(S // 64, max(1, 768 // max(B, 1))))
        ws = torch.empty(capi.decoder_attn_workspace_bytes(B, H, 64, splits) // 4, **f32)
        g = lambda t: t.detach().to(torch.float32).contiguous()

        x0 = self.class_embedding.detach().to(torch.float32).view(1, D).repeat(B, 1).contiguous()
        x = torch.empty(B, D, **f32)
        capi.layernorm(x0, g(self.ln_pre.weight), g(self.ln_pre.bias), x)
        h = torch.empty(B, D, **f32)
        q = torch.empty(B, 2 * D, **f32)
        mix = torch.empty(B, D, **f32)
        stats = torch.empty(B, H, 2, **f32)
        u = torch.empty(B, 4 * D, **f32)
        for i, blk in enumerate(self.transformer.resblocks):
            capi.layernorm(x, g(blk.ln_1.weight), g(blk.ln_1.bias), h)
            capi.linear_rows(h, g(blk.attn.in_proj.weight), g(blk.attn.in_proj.bias), q)
            capi.decoder_attn_fwd(q, k_all[i], v_all[i], mask, mix, stats, ws, splits, B, T, P, H)
            capi.linear_rows(mix, g(blk.attn.out_proj.weight), g(blk.attn.out_proj.bias), x, capi.EPI_BIAS_RESIDUAL)
            capi.layernorm(x, g(blk.ln_2.weight), g(blk.ln_2.bias), h)
            capi.linear_rows(h, g(blk.mlp.c_fc.weight), g(blk.mlp.c_fc.bias), u, capi.EPI_BIAS_QUICKGELU)
            capi.linear_rows(u, g(blk.mlp.c_proj.weight), g(blk.mlp.c_proj.bias), x, capi.EPI_BIAS_RESIDUAL)
            aq = self.transformer.augment_query_embeddings
            if len(aq) > 0 and i != L - 1:
                # result.append(x) precedes the add in the reference (models.py:263-267); only the
                # last block's x is read when there is one projection per task (models.py:340-341)
                x += aq[i].detach().to(torch.float32)
        feat = torch.empty(B, D, **f32)
        raws, outs = [], []
        for i, od in enumerate(self.out_dims):
            raw = torch.empty(B, od, **f32)
            logits = torch.empty(B, od, **f32)
            capi.head_fwd(x, g(self.ln_post.weight), g(self.ln_post.bias), g(self.task_projections[i][-1]), feat, raw, logits)
            raws.append(raw)
            outs.append(logits)
        return raws, feat, outs
