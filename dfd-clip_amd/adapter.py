"""CompInvAdapter on HIP kernels (reference `src/models.py:783-940`).

Per selected layer and per tensor (k, v): a residual bottleneck
    kv <- kv + Linear_{x->D}( GELU( LayerNorm( Linear_{D->x}(kv) ) ) )        structs 768-x-768-{nln, ln, z0}
    kv <- kv + Linear_{x->D}( LayerNorm( GELU( Linear_{D->x}(kv) ) ) )        structs 768-x-768, legacy-768-x-768
applied to the exported K/V (CLS row dropped) BEFORE the decoder adds its temporal positional
embedding.  Parameter names follow the reference's `nn.Sequential` indices
(`l{i}_{k|v}.0.weight`, `.1.weight`, `.1.bias`, `.4.weight`), so reference checkpoints load.

Kernel sequence per (layer, tensor) on the packed export `[N*P, D]`:
    dfd_gemm(BIAS, bias = NULL)                    a1 = kv · W0ᵀ           [N*P, x]
    dfd_adapter_norm_gelu                          a2 = GELU(LN(a1))        "nln": stats over (P, x); "ln"/"z0": over x
    dfd_gemm(RESIDUAL_POS)                         kv += a2 · W4ᵀ + pos[frame % T]   (in place, one rounding)
In `train()` mode with `config.dropout` > 0 the reference's two nn.Dropout layers act as there: p/10 (p/5 for
`768-x-768`, none for `legacy-`) on a2, and p on a2 · W4ᵀ before the residual add (the GEMM epilogue draws the
mask); masks are counter-based and regenerated in the backward (csrc/dropout.hpp).

Training (`run`, with grad enabled and trainable parameters) goes through `_AdapterFn`: the forward is
out of place and keeps a1; the backward takes dK/dV from the decoder's attention backward and runs
    a2  = GELU(LN(a1))                      (recomputed)            dfd_adapter_norm_gelu
    dW4 = dOutᵀ · a2                        [D, x]                  dfd_gemm_at_b   (split-K over B·T·P rows)
    dA2 = dOut · W4                         [rows, x]               dfd_gemm
    da1, dγ, dβ = LN/GELU backward                                  dfd_adapter_norm_gelu_bwd
    dW0 = da1ᵀ · X                          [x, D]                  dfd_gemm_at_b
The encoder output X needs no gradient (frozen encoder, reference models.py:440).
"""
import collections
import logging
import weakref

import torch
from torch import nn

from . import capi
from .encoder import RuntimeStateMixin

# captured forward / backward kernel sequences per adapter instance and input signature (as the decoder's: decoder.py)
_GRAPHS = weakref.WeakKeyDictionary()

# struct -> (index of the LayerNorm and of the output Linear inside the reference's nn.Sequential, kernel mode:
# 0 = GELU(LN_row(a)), 1 = GELU(LN_joint(a)), 2 = LN_row(GELU(a)))   (reference models.py:795-875)
_STRUCTS = {"768-x-768-nln": (1, 4, 1), "768-x-768-ln": (1, 4, 0), "768-x-768-z0": (1, 4, 0),
            "768-x-768": (2, 4, 2), "legacy-768-x-768": (2, 3, 2)}
_SUPPORTED = tuple(_STRUCTS)


class CompInvAdapter(RuntimeStateMixin, nn.Module):
    _RUNTIME_STATE = {"_prep": None, "_after_backward": None, "_graphs_failed": None}
    max_graphs = 4

    def invalidate_caches(self):
        self._prep = None
        _GRAPHS.pop(self, None)

    def __init__(self, config, detector, num_frames=50):
        super().__init__()
        enc = detector.encoder
        width = enc.width
        self.patches = (enc.input_resolution // enc.patch_size) ** 2
        self.struct = config.adapter.struct.type
        if self.struct not in _SUPPORTED:
            raise NotImplementedError(f"adapter struct {self.struct} is not built (supported: {_SUPPORTED})")
        self.ln_idx, self.out_idx, self.mode = _STRUCTS[self.struct]
        self.inner = int(config.adapter.struct.x)
        self.width = width
        p = float(config.dropout) if "dropout" in config else 0.0
        self.drop_outer = p  # every struct ends in nn.Dropout(p) (models.py:807, :820, :836, :852, :864)
        self.drop_inner = 0.0 if self.struct == "legacy-768-x-768" else (p / 5 if self.struct == "768-x-768" else p / 10)
        self.residual = True
        self.n_layers = len(detector.layer_indices)
        for i in range(self.n_layers):
            for j in ("k", "v"):
                ln_shape = (self.patches, self.inner) if self.struct.endswith("nln") else self.inner
                if self.struct == "768-x-768":
                    seq = nn.Sequential(nn.Linear(width, self.inner, bias=False), nn.GELU(), nn.LayerNorm(self.inner),
                                        nn.Dropout(config.dropout / 5), nn.Linear(self.inner, width, bias=False),
                                        nn.Dropout(config.dropout))
                elif self.struct == "legacy-768-x-768":
                    seq = nn.Sequential(nn.Linear(width, self.inner, bias=False), nn.GELU(), nn.LayerNorm(self.inner),
                                        nn.Linear(self.inner, width, bias=False), nn.Dropout(config.dropout))
                else:
                    seq = nn.Sequential(nn.Linear(width, self.inner, bias=False), nn.LayerNorm(ln_shape), nn.GELU(),
                                        nn.Dropout(config.dropout / 10), nn.Linear(self.inner, width, bias=False),
                                        nn.Dropout(config.dropout))
                if self.struct.endswith("z0"):  # starts as the identity map (models.py:867-869)
                    seq[1].weight.data.zero_()
                    seq[4].weight.data.zero_()
                setattr(self, f"l{i}_{j}", seq)
        self._prep = None
        self._after_backward = None  # one-shot callback for the next autograd node (Detector's encoder pipelining)
        # training: replay the forward / backward kernel sequences as HIP graphs (`Detector.static_graphs`).  The ~170
        # launches and ~100 small tensor ops of an adapter step then cost the host two graph launches; the outputs
        # (k, v) = adapter(raw) + pos live in the graph's static buffers — what a call returned is overwritten by the
        # next call with the same input buffers, and the decoder's own graphs find them at fixed addresses
        self.use_graphs = False
        self._graphs_failed = None

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
                    ln, out = seq[self.ln_idx], seq[self.out_idx]
                    w[(i, j)] = (seq[0].weight.detach().to(act).contiguous(), ln.weight.detach().float().contiguous(),
                                 ln.bias.detach().float().contiguous(), out.weight.detach().to(act).contiguous())
            self._prep = (key, w)
        return self._prep[1]

    def run(self, k_raw, v_raw, num_frames, temporal_pos, drop_rng=None):
        """(k, v) = adapter(raw export) + pos.  In place without autograd; out of place through
        `_AdapterFn` when gradients can flow to the adapter's parameters.  `drop_rng`: this step's
        dropout state (device int64 {seed, step}) in train mode, None = no dropout."""
        names = [n for n, p in self.named_parameters()]
        params = [p for n, p in self.named_parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _AdapterFn.apply(self, k_raw, v_raw, num_frames, temporal_pos, names, drop_rng, *params)
        return self.apply_packed(k_raw, v_raw, num_frames, temporal_pos, drop_rng)

    def _drops(self, drop_rng, i, jj):
        """(inner, outer) dropout descriptors of layer i, tensor jj (0 = k, 1 = v); None where inactive."""
        if drop_rng is None:
            return None, None
        base = 1000 + 4 * (2 * i + jj)
        inner = capi.Dropout(drop_rng, base, self.drop_inner) if self.drop_inner > 0 else None
        outer = capi.Dropout(drop_rng, base + 1, self.drop_outer) if self.drop_outer > 0 else None
        return inner, outer

    def _stage_weights(self, w, act):
        out = {}
        for i in range(self.n_layers):
            for j in ("k", "v"):
                pre = f"l{i}_{j}."
                out[(i, j)] = (w[pre + "0.weight"].to(act).contiguous(), w[pre + f"{self.ln_idx}.weight"].float().contiguous(),
                               w[pre + f"{self.ln_idx}.bias"].float().contiguous(),
                               w[pre + f"{self.out_idx}.weight"].to(act).contiguous())
        return out

    def _forward_train(self, w, k_raw, v_raw, num_frames, temporal_pos, drop_rng=None):
        act = k_raw.dtype
        sw = self._stage_weights(w, act)
        L, rows, D = k_raw.shape
        P, x = self.patches, self.inner
        frames = rows // P
        joint = self.mode
        k_out, v_out = torch.empty_like(k_raw), torch.empty_like(v_raw)
        # GELU-first structs keep the projection output in f32 (LayerNorm after the non-linearity amplifies
        # the rounding of its input ~4x more than the other order)
        a1_all = torch.empty(L, 2, rows, x, device=k_raw.device, dtype=torch.float32 if self.mode == 2 else act)
        # the second Linear's input (after the inner dropout) is kept as well: 48 MB per (layer, tensor) at B16xT30 buys the
        # backward one normalisation pass less each
        a2_all = torch.empty(L, 2, rows, x, device=k_raw.device, dtype=act)
        for i in range(L):
            for jj, (j, src, dst) in enumerate((("k", k_raw, k_out), ("v", v_raw, v_out))):
                w0, lw, lb, w4 = sw[(i, j)]
                inner, outer = self._drops(drop_rng, i, jj)
                a2 = a2_all[i, jj]
                capi.gemm(src[i], w0, a1_all[i, jj], None, capi.EPI_BIAS)
                capi.adapter_norm_gelu(a1_all[i, jj], a2, lw, lb, frames, P, x, joint)
                if inner is not None:
                    capi.dropout(a2, a2, inner)
                capi.gemm(a2, w4, dst[i], None, capi.EPI_RESIDUAL_POS, pos=temporal_pos, tokens=P + 1, frames_per_clip=num_frames,
                          residual=src[i], drop=outer)
        return k_out, v_out, (a1_all, a2_all)

    def _backward_train(self, w, k_raw, v_raw, saved, dk, dv, drop_rng=None):
        a1_all, a2_all = saved
        act = k_raw.dtype
        sw = self._stage_weights(w, act)
        L, rows, D = k_raw.shape
        P, x = self.patches, self.inner
        frames = rows // P
        joint = self.mode
        dev = k_raw.device
        f32 = dict(device=dev, dtype=torch.float32)
        da2 = torch.empty(rows, x, device=dev, dtype=act)
        da1 = torch.empty(rows, x, device=dev, dtype=act)
        nb = max(capi.gemm_at_b_workspace_bytes(rows, D, x, act), capi.gemm_at_b_workspace_bytes(rows, x, D, act))
        ws_ab = torch.empty(nb // 4 + 64, **f32)
        ws_ln = torch.empty(capi.adapter_norm_gelu_bwd_workspace_bytes(frames, P, x, joint) // 4 + 4, **f32)
        grads = {}
        d_masked = None
        for i in range(L):
            for jj, (j, src, dout) in enumerate((("k", k_raw, dk), ("v", v_raw, dv))):
                pre = f"l{i}_{j}."
                w0, lw, lb, w4 = sw[(i, j)]
                inner, outer = self._drops(drop_rng, i, jj)
                if outer is not None:  # gradient entering the dropped a2 · W4ᵀ: same mask, same scale
                    if d_masked is None:
                        d_masked = torch.empty(rows, D, device=dev, dtype=act)
                    d_o = capi.dropout(dout[i], d_masked, outer)
                else:
                    d_o = dout[i].to(act) if dout.dtype != act else dout[i]
                a1, a2 = a1_all[i, jj], a2_all[i, jj]
                dw4 = torch.empty(D, x, **f32)
                capi.gemm_at_b(d_o, a2, dw4, ws_ab)
                w4t = w[pre + f"{self.out_idx}.weight"].float().t().contiguous().to(act)  # [x, D]: dA2 = dOut @ W4 as A @ (W4^T)^T
                capi.gemm(d_o, w4t, da2, None, capi.EPI_BIAS)
                if inner is not None:
                    capi.dropout(da2, da2, inner)
                dlw, dlb = torch.empty_like(lw), torch.empty_like(lb)
                capi.adapter_norm_gelu_bwd(a1, da2, da1, lw, lb, dlw, dlb, ws_ln, frames, P, x, joint)
                dw0 = torch.empty(x, D, **f32)
                capi.gemm_at_b(da1, src[i], dw0, ws_ab)
                grads[pre + "0.weight"], grads[pre + f"{self.out_idx}.weight"] = dw0, dw4
                grads[pre + f"{self.ln_idx}.weight"], grads[pre + f"{self.ln_idx}.bias"] = dlw, dlb
        return grads

    @torch.no_grad()
    def apply_packed(self, k_all, v_all, num_frames, temporal_pos, drop_rng=None):
        """In place on the packed exports [L, N*P, D] (raw encoder K/V, no positional embedding yet):
        afterwards they hold adapter(kv) + pos, i.e. exactly what the decoder attends to."""
        if not k_all.is_cuda:
            raise capi.DfdError("the adapter runs on HIP kernels only: pass device tensors")
        act = k_all.dtype
        w = self._weights(act)
        L, rows, D = k_all.shape
        P, x = self.patches, self.inner
        frames = rows // P
        a1 = torch.empty(rows, x, device=k_all.device, dtype=torch.float32 if self.mode == 2 else act)
        a2 = torch.empty(rows, x, device=k_all.device, dtype=act) if self.mode == 2 else a1
        joint = self.mode
        for i in range(L):
            for jj, (j, t) in enumerate((("k", k_all), ("v", v_all))):
                w0, lw, lb, w4 = w[(i, j)]
                inner, outer = self._drops(drop_rng, i, jj)
                capi.gemm(t[i], w0, a1, None, capi.EPI_BIAS)
                capi.adapter_norm_gelu(a1, a2, lw, lb, frames, P, x, joint)
                if inner is not None:
                    capi.dropout(a2, a2, inner)
                capi.gemm(a2, w4, t[i], None, capi.EPI_RESIDUAL_POS, pos=temporal_pos, tokens=P + 1, frames_per_clip=num_frames,
                          drop=outer)
        return k_all, v_all


    # ---- HIP-graph replay ------------------------------------------------------------------------------------------
    def _graph_forward(self, w, k_raw, v_raw, num_frames, temporal_pos, drop_rng, params):
        if self._graphs_failed:
            return None
        key = (k_raw.data_ptr(), v_raw.data_ptr(), tuple(k_raw.shape), str(k_raw.dtype), num_frames,
               None if temporal_pos is None else tuple(temporal_pos.shape),
               tuple(p.data_ptr() for p in params), drop_rng is not None, self.drop_inner, self.drop_outer, self.patches)
        graphs = _GRAPHS.setdefault(self, collections.OrderedDict())
        ent = graphs.get(key)
        if ent is not None:
            graphs.move_to_end(key)
        else:
            while len(graphs) >= self.max_graphs:
                torch.cuda.synchronize()
                graphs.popitem(last=False)
            # `temporal_pos` is a fresh f32 copy of the parameter on every call: the graph reads a static copy of it,
            # refreshed before each replay (as the dropout state)
            ent = dict(bwd={}, rng=None if drop_rng is None else drop_rng.clone(),
                       pos=None if temporal_pos is None else temporal_pos.clone())
            self._forward_train(w, k_raw, v_raw, num_frames, ent["pos"], ent["rng"])  # eager once: lazy initialisations
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    k_out, v_out, a1_all = self._forward_train(w, k_raw, v_raw, num_frames, ent["pos"], ent["rng"])
            except Exception as e:  # capture is an optimisation: a runtime that refuses it leaves the eager launches
                self._graphs_failed = f"{type(e).__name__}: {e}"
                logging.warning("adapter: HIP graph capture failed, staying on eager launches (%s)", self._graphs_failed)
                torch.cuda.synchronize()
                return None
            ent.update(fwd=g, k_out=k_out, v_out=v_out, a1_all=a1_all)
            graphs[key] = ent
        if drop_rng is not None:
            ent["rng"].copy_(drop_rng)
        if temporal_pos is not None:
            ent["pos"].copy_(temporal_pos)
        ent["fwd"].replay()
        return ent

    def _graph_backward(self, ent, w, k_raw, v_raw, dk, dv):
        sig = (dk.data_ptr(), dv.data_ptr(), str(dk.dtype))
        b = ent["bwd"].get(sig)
        if b is None:
            if len(ent["bwd"]) >= 2:  # gradient buffers keep moving (the decoder is not replaying graphs): copy into a static pair
                sig = "static"
                b = ent["bwd"].get(sig)
        if b is None:
            static = sig == "static"
            b = dict(dk=dk.clone() if static else dk, dv=dv.clone() if static else dv, static=static)
            self._backward_train(w, k_raw, v_raw, ent["a1_all"], b["dk"], b["dv"], ent["rng"])  # eager once
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    grads = self._backward_train(w, k_raw, v_raw, ent["a1_all"], b["dk"], b["dv"], ent["rng"])
            except Exception as e:
                self._graphs_failed = f"{type(e).__name__}: {e}"
                logging.warning("adapter: HIP graph capture of the backward failed, staying on eager launches (%s)", self._graphs_failed)
                torch.cuda.synchronize()
                return self._backward_train(w, k_raw, v_raw, ent["a1_all"], dk, dv, ent["rng"])
            b.update(graph=g, grads=grads)
            ent["bwd"][sig] = b
        if b["static"]:
            b["dk"].copy_(dk)
            b["dv"].copy_(dv)
        b["graph"].replay()
        names = list(b["grads"])
        src = [b["grads"][k] for k in names]
        dst = [torch.empty_like(t) for t in src]  # autograd may keep or accumulate into what it is handed
        torch._foreach_copy_(dst, src)
        return dict(zip(names, dst))


class _AdapterFn(torch.autograd.Function):
    """autograd node around the adapter's HIP forward / backward.  Outputs (k, v) = adapter(raw) + pos."""

    @staticmethod
    def forward(ctx, adapter, k_raw, v_raw, num_frames, temporal_pos, names, drop_rng, *params):
        w = {n: p.detach() for n, p in zip(names, params)}
        ent = adapter._graph_forward(w, k_raw, v_raw, num_frames, temporal_pos, drop_rng, params) if adapter.use_graphs else None
        if ent is not None:
            k_out, v_out, a1_all = ent["k_out"], ent["v_out"], ent["a1_all"]
        else:
            k_out, v_out, a1_all = adapter._forward_train(w, k_raw, v_raw, num_frames, temporal_pos, drop_rng)
        ctx.graph_entry = ent
        ctx.adapter, ctx.w, ctx.names, ctx.drop_rng = adapter, w, names, drop_rng
        ctx.after_backward, adapter._after_backward = adapter._after_backward, None
        ctx.saved = (k_raw, v_raw, a1_all)
        ctx.req = [p.requires_grad for p in params]
        return k_out, v_out

    @staticmethod
    def backward(ctx, dk, dv):
        k_raw, v_raw, a1_all = ctx.saved
        if ctx.graph_entry is not None:
            grads = ctx.adapter._graph_backward(ctx.graph_entry, ctx.w, k_raw, v_raw, dk.contiguous(), dv.contiguous())
        else:
            grads = ctx.adapter._backward_train(ctx.w, k_raw, v_raw, a1_all, dk.contiguous(), dv.contiguous(), ctx.drop_rng)
        out = [grads.get(nm) if rq else None for nm, rq in zip(ctx.names, ctx.req)]
        if ctx.after_backward is not None:
            ctx.after_backward()
        return (None, None, None, None, None, None, None, *out)
