"""CompInvAdapter on HIP kernels (reference `src/models.py:783-940`).

Per selected layer and per tensor (k, v): a residual bottleneck
    kv <- kv + Linear_{x->D}( GELU( LayerNorm( Linear_{D->x}(kv) ) ) )
applied to the exported K/V (CLS row dropped) BEFORE the decoder adds its temporal positional
embedding.  Parameter names follow the reference's `nn.Sequential` indices
(`l{i}_{k|v}.0.weight`, `.1.weight`, `.1.bias`, `.4.weight`), so reference checkpoints load.

Kernel sequence per (layer, tensor) on the packed export `[N*P, D]`:
    dfd_gemm(BIAS, bias = NULL)                    a1 = kv · W0ᵀ           [N*P, x]
    dfd_adapter_norm_gelu                          a2 = GELU(LN(a1))        "nln": stats over (P, x); "ln"/"z0": over x
    dfd_gemm(RESIDUAL_POS)                         kv += a2 · W4ᵀ + pos[frame % T]   (in place, one rounding)
Eval-mode semantics (dropout = identity).  Forward only for now: training a non-frozen adapter
needs the K/V gradients pushed through these three stages.
"""
import torch
from torch import nn

from . import capi

_SUPPORTED = ("768-x-768-nln", "768-x-768-ln", "768-x-768-z0")


class CompInvAdapter(nn.Module):
    def __init__(self, config, detector, num_frames=50):
        super().__init__()
        enc = detector.encoder
        width = enc.width
        self.patches = (enc.input_resolution // enc.patch_size) ** 2
        self.struct = config.adapter.struct.type
        if self.struct not in _SUPPORTED:
            raise NotImplementedError(f"adapter struct {self.struct} is not built (supported: {_SUPPORTED})")
        self.inner = int(config.adapter.struct.x)
        self.width = width
        self.residual = True
        self.n_layers = len(detector.layer_indices)
        for i in range(self.n_layers):
            for j in ("k", "v"):
                ln_shape = (self.patches, self.inner) if self.struct.endswith("nln") else self.inner
                seq = nn.Sequential(nn.Linear(width, self.inner, bias=False), nn.LayerNorm(ln_shape), nn.GELU(),
                                    nn.Dropout(config.dropout / 10), nn.Linear(self.inner, width, bias=False),
                                    nn.Dropout(config.dropout))
                if self.struct.endswith("z0"):  # starts as the identity map (models.py:867-869)
                    seq[1].weight.data.zero_()
                    seq[4].weight.data.zero_()
                setattr(self, f"l{i}_{j}", seq)
        self._prep = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._prep = None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._prep = None
        return out

    def _weights(self, act):
        key = (act, tuple(p._version for p in self.parameters()))
        if self._prep is None or self._prep[0] != key:
            w = {}
            for i in range(self.n_layers):
                for j in ("k", "v"):
                    seq = getattr(self, f"l{i}_{j}")
                    w[(i, j)] = (seq[0].weight.detach().to(act).contiguous(), seq[1].weight.detach().float().contiguous(),
                                 seq[1].bias.detach().float().contiguous(), seq[4].weight.detach().to(act).contiguous())
            self._prep = (key, w)
        return self._prep[1]

    @torch.no_grad()
    def apply_packed(self, k_all, v_all, num_frames, temporal_pos):
        """In place on the packed exports [L, N*P, D] (raw encoder K/V, no positional embedding yet):
        afterwards they hold adapter(kv) + pos, i.e. exactly what the decoder attends to."""
        if not k_all.is_cuda:
            raise capi.DfdError("the adapter runs on HIP kernels only: pass device tensors")
        act = k_all.dtype
        w = self._weights(act)
        L, rows, D = k_all.shape
        P, x = self.patches, self.inner
        frames = rows // P
        a1 = torch.empty(rows, x, device=k_all.device, dtype=act)
        joint = self.struct.endswith("nln")
        for i in range(L):
            for j, t in (("k", k_all), ("v", v_all)):
                w0, lw, lb, w4 = w[(i, j)]
                capi.gemm(t[i], w0, a1, None, capi.EPI_BIAS)
                capi.adapter_norm_gelu(a1, a1, lw, lb, frames, P, x, joint)
                capi.gemm(a1, w4, t[i], None, capi.EPI_RESIDUAL_POS, pos=temporal_pos, tokens=P + 1, frames_per_clip=num_frames)
        return k_all, v_all
