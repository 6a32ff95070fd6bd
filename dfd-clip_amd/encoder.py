"""CLIP visual encoder that exports per-layer attention keys/values, on HIP kernels.

Host-side mirror of the reference's modified `VisionTransformer` (reference
`src/clip/model.py:254-294`; blocks `:202-226`; attention `:171-199`; stack `:229-251`): same
constructor, attributes, parameter names (so reference checkpoints load unchanged) and the
same `forward(x, with_out, with_q)` result — a list with one dict per block holding
`k`, `v` `[N, tokens, heads, 64]` (bias included, un-scaled, CLS row kept).  `ln_post` and
`proj` exist as parameters and, as in the reference, are never applied.

All arithmetic runs in libdfdclip_hip.so.  Three precisions:
  * "fp32": f32 operands on the exact-f32 matrix cores — the 1e-3 parity path;
  * "bf16": bf16 GEMM/attention operands, f32 accumulation, f32 residual stream, f32
    LayerNorm/softmax statistics — the throughput path;
  * "fp8" (BASELINE.json configs[4]): the bf16 path with the three large projections of every block — q|k|v,
    c_fc, c_proj, 92 % of the encoder's FLOPs — on OCP e4m3 operands (`dfd_gemm_fp8`, block-scaled matrix
    cores at twice the bf16 rate).  Weights: one scale per output row (amax / 448), quantised once.
    Activations: LayerNorm writes its output, and c_fc its QuickGELU output, directly as e4m3 with ONE static
    scale per tensor and layer, taken from a calibration batch (`calibrate_fp8`; the first fp8 forward
    calibrates on its own input otherwise): amax / 448, values beyond it saturate.  Accumulation, bias,
    residual stream, attention, out_proj and the K/V export stay as in the bf16 path.

HBM layout for a batch of N frames (M = N*tokens rows, D = width), all row-major:
  x    [M, D]   f32   residual stream (updated in place by the residual epilogues)
  h    [M, D]   act   LayerNorm output = GEMM A operand
  qkv  [M, 3D]  act   q | k | v column blocks, head = 64 contiguous columns
  mix  [M, D]   act   attention output
  u    [M, 4D]  act   QuickGELU(c_fc) output
`extract_kv` additionally writes the decoder's K/V operands straight from the QKV GEMM
epilogue: `[N*P, D]` per selected layer, CLS row dropped, temporal positional embedding added
(reference `src/models.py:505-509`, `:326-334`), and skips the work of the last selected
layer that cannot reach any exported tensor.
"""
import contextlib

import torch
from torch import nn

from . import capi


class RuntimeStateMixin:
    """Per-process runtime state (device caches, streams, events, captured graphs) is not part of the model:
    `copy.deepcopy` (the trainer's teacher mode, reference `src/trainer.py:68`) gives the copy fresh, empty
    state instead of duplicating gigabytes of cached buffers or sharing streams and events."""
    _RUNTIME_STATE = {}

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in self._RUNTIME_STATE:
                new.__dict__[k] = copy.deepcopy(self._RUNTIME_STATE[k])
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new


class _Holder(nn.Module):
    """Parameter container; never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder")


class _AttnParams(_Holder):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.normal_(self.in_proj_weight, std=d ** -0.5)
        nn.init.normal_(self.in_proj_bias, std=0.02)


class _Mlp(_Holder):
    def __init__(self, d):
        super().__init__()
        self.c_fc = nn.Linear(d, 4 * d)
        self.c_proj = nn.Linear(4 * d, d)


class ResidualAttentionBlock(_Holder):
    def __init__(self, d_model, n_head):
        super().__init__()
        self.attn = _AttnParams(d_model)
        self.attn.n_head = n_head
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = _Mlp(d_model)
        self.ln_2 = nn.LayerNorm(d_model)


class Transformer(_Holder):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class VisionTransformer(RuntimeStateMixin, nn.Module):
    _RUNTIME_STATE = {"_prepared": None, "_ws": {}, "_side_streams": [], "_calib": None}

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, precision="bf16"):
        super().__init__()
        assert width % heads == 0 and width // heads == 64, "kernels are built for 64-wide heads"
        assert precision in ("fp32", "bf16", "fp8")
        self.input_resolution = input_resolution
        self.output_dim = output_dim
        self.width = width
        self.layers = layers
        self.heads = heads
        self.patch_size = patch_size
        self.precision = precision
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._prepared = None
        self._ws = {}
        self.frame_chunk = 0  # frames per pass; 0 = the batch split evenly over `streams`
        # independent frame chunks can run round-robin on several HIP streams so one chunk's kernels fill the
        # tails of another's (measured on MI355X, B16xT30: 2 streams +3 %, 3-4 streams lose); default 1
        # keeps per-kernel timings meaningful
        self.streams = 1
        self._side_streams = []
        # uint8 ingest (`Detector._transform`, reference models.py:756-768): CLIP's normalisation constants;
        # antialias follows current torchvision's tensor default (older releases resized without it)
        self.pixel_mean = (0.48145466, 0.4578275, 0.40821073)
        self.pixel_std = (0.26862954, 0.26130258, 0.27577711)
        self.antialias = True
        # bf16 path: residual branches are stored as bf16 deltas and added inside the next LayerNorm
        # (see `_residual`); the fp32 parity path keeps the read-modify-write epilogue
        self.deferred_residual = precision in ("bf16", "fp8")
        self._fp8 = None  # fp8 path: per-layer activation scales and column-scale vectors, DERIVED from the two below
        self._fp8_amax = None  # ... the calibrated activation maxima [layers, 3] (host, f32): survive .to() / load_state_dict /
        self._fp8_margin = 1.0  # broadcasts (which only invalidate what is derived from the weights); `fp8_calibration()`
        self._calib = None
        # which GEMM outputs are stored non-temporally (capi.gemm stream_out): they are written once and read back only
        # after other traffic has flushed the caches anyway, and keeping them out of L2 leaves the operand panels there
        self.stream_out = {"qkv": True, "out": True, "fc": True, "proj": True}
        # compute units the persistent GEMMs leave free (capi.gemm spare_cus): set by Detector while it overlaps the
        # decoder's backward / optimizer of the previous step with this encoder pass
        self.spare_cus = 0
        self.spare_layers = 0  # ... and only in the first `spare_layers` blocks of a pass (0 = all of them)
        # ... and only by these GEMMs of a block.  c_proj (N = D: 4.3 rounds of 256-row tiles on every CU, 5 either way on
        # 224) gives its spare CUs away for free; q|k|v and c_fc pay a whole extra round of tiles for them
        self.spare_gemms = {"qkv": False, "out": False, "fc": False, "proj": True}
        # ... except in the first `spare_window_layers` blocks of a pass, where EVERY GEMM leaves them: at world size > 1
        # the gradient all-reduce of the previous step (an RCCL kernel on the caller's stream) lands in that window, and a
        # persistent GEMM that holds all 256 CUs would make its workgroups queue for a whole GEMM each
        self.spare_window_layers = 0
        # True: a GEMM outside the collective's window leaves the spare CUs only where that costs its shape no extra round of
        # tiles.  `Detector._encode` clears it for training passes: there the decoder's backward beside the encoder is worth
        # a fifth round of ViT-L/14's c_proj (A/B, B8xT30: train 201.5 binding vs 200.0 if-free; forward 209.6 vs 213.0)
        self.spare_if_free = True

    # ---- derived device-side operands ---------------------------------------------------
    @property
    def tokens(self):
        return (self.input_resolution // self.patch_size) ** 2 + 1

    @property
    def act_dtype(self):
        return torch.float32 if self.precision == "fp32" else torch.bfloat16

    # ---- fp8 activation scales ------------------------------------------------------------------
    def _ensure_fp8(self, frames):
        if self.precision == "fp8" and self._fp8_amax is None and getattr(self, "_calib", None) is None:
            import logging
            logging.warning("fp8 encoder: no calibration yet, taking the activation scales from this batch "
                            "(call encoder.calibrate_fp8(frames) with representative frames to fix them beforehand)")
            self.calibrate_fp8(frames[:min(frames.shape[0], 64)])

    @torch.no_grad()
    def calibrate_fp8(self, x, margin=1.0):
        """Static activation scales of the fp8 path from one batch of frames [N,3,R,R]: runs the bf16 arithmetic
        once, records the largest magnitude of each quantised tensor (ln_1 output, ln_2 output, QuickGELU(c_fc)
        output) per layer, and fixes scale = margin * amax / 448.  Later batches saturate beyond it."""
        assert self.precision == "fp8"
        p = self._prepare()
        frames = self._as_frames(x)
        n = frames.shape[0]
        M = n * self.tokens
        rec = []
        self._calib = rec
        try:
            ws = self._workspace(n, 1, slot=99)
            self._embed(frames, ws, p)
            for bp in p["blocks"]:
                self._block(ws, bp, ws["qkv"][0], M, n)
        finally:
            self._calib = None
            self._ws.pop((n, self.precision, 1, 99), None)
        amax = torch.stack(rec).view(len(p["blocks"]), 3).float().cpu()  # one host sync, at calibration time only
        self._fp8_amax, self._fp8_margin, self._fp8 = amax, float(margin), None
        self._fp8_layers()
        return (amax * margin / capi.FP8_MAX).clamp_min(1e-12)

    def fp8_calibration(self):
        """The calibrated activation maxima ([layers, 3] f32 on the host, or None) — store them next to a checkpoint and
        hand them to `load_fp8_calibration` (they are not part of `state_dict`: the schema is the reference's)."""
        return None if self._fp8_amax is None else self._fp8_amax.clone()

    def load_fp8_calibration(self, amax, margin=1.0):
        amax = torch.as_tensor(amax, dtype=torch.float32).cpu()
        assert amax.shape == (self.layers, 3)
        self._fp8_amax, self._fp8_margin, self._fp8 = amax, float(margin), None

    def _fp8_layers(self):
        """Per-layer activation scales x weight-row scales, rebuilt from the calibration whenever the prepared weights are."""
        if self._fp8 is None:
            assert self._fp8_amax is not None, "fp8 encoder: not calibrated"
            p = self._prepare()
            scales = (self._fp8_amax * self._fp8_margin / capi.FP8_MAX).clamp_min(1e-12)
            layers = []
            for l, bp in enumerate(p["blocks"]):
                s1, s2, su = (float(v) for v in scales[l])
                layers.append(dict(h1_inv=1.0 / s1, h2_inv=1.0 / s2, u_inv=1.0 / su, cs_qkv=(bp["s_qkv"] * s1).contiguous(),
                                   cs_fc=(bp["s_fc"] * s2).contiguous(), cs_proj=(bp["s_proj"] * su).contiguous()))
            self._fp8 = layers
        return self._fp8

    def invalidate(self):
        """Call after changing parameters in place (load_state_dict and .to() do it themselves)."""
        self._prepared = None
        self._fp8 = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._prepared = None
        self._fp8 = None
        self._ws = {}
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._prepared = None
        self._fp8 = None
        return out

    def _prepare(self):
        if self._prepared is not None:
            return self._prepared
        dev = self.class_embedding.device
        if dev.type != "cuda":
            raise capi.DfdError("the encoder runs on HIP kernels only: move the model to a GPU (.to('cuda'))")
        act = self.act_dtype
        kreal = 3 * self.patch_size ** 2
        kpad = (kreal + 63) // 64 * 64  # multiple of the tuned GEMM's K step (ViT-L/14: 588 -> 640); pad columns are zero
        wp = torch.zeros(self.width, kpad, device=dev, dtype=torch.float32)
        wp[:, :kreal] = self.conv1.weight.detach().reshape(self.width, kreal)
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        p = dict(kpad=kpad, w_patch=wp.to(act).contiguous(), cls=f32(self.class_embedding),
                 pos=f32(self.positional_embedding), ln_pre=(f32(self.ln_pre.weight), f32(self.ln_pre.bias)), blocks=[])
        def q8(w):
            """[N, K] weight -> (e4m3 bytes of w / scale_row, scale_row f32 [N]); scale_row = amax_row / 448."""
            w = w.detach().to(torch.float32)
            sc = (w.abs().amax(dim=1) / capi.FP8_MAX).clamp_min(1e-12)
            return (w / sc[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous(), sc.contiguous()

        for blk in self.transformer.resblocks:
            if self.precision == "fp8":
                f8 = dict(zip(("w_qkv8", "s_qkv"), q8(blk.attn.in_proj_weight)))
                f8.update(zip(("w_fc8", "s_fc"), q8(blk.mlp.c_fc.weight)))
                f8.update(zip(("w_proj8", "s_proj"), q8(blk.mlp.c_proj.weight)))
            else:
                f8 = {}
            p["blocks"].append(dict(f8, idx=len(p["blocks"]), 
                ln1=(f32(blk.ln_1.weight), f32(blk.ln_1.bias)), ln2=(f32(blk.ln_2.weight), f32(blk.ln_2.bias)),
                w_qkv=blk.attn.in_proj_weight.detach().to(act).contiguous(), b_qkv=f32(blk.attn.in_proj_bias),
                w_out=blk.attn.out_proj.weight.detach().to(act).contiguous(), b_out=f32(blk.attn.out_proj.bias),
                w_fc=blk.mlp.c_fc.weight.detach().to(act).contiguous(), b_fc=f32(blk.mlp.c_fc.bias),
                w_proj=blk.mlp.c_proj.weight.detach().to(act).contiguous(), b_proj=f32(blk.mlp.c_proj.bias)))
        self._prepared = p
        return p

    def _workspace(self, n, keep_layers, slot=0):
        """Activation buffers for n frames.  Rows are padded to a multiple of 256 so tiled kernels
        never read out of bounds; pad rows hold zeros/garbage that is never stored to real rows."""
        key = (n, self.precision, keep_layers, slot)
        ws = self._ws.get(key)
        if ws is None:
            dev = self.class_embedding.device
            act, D = self.act_dtype, self.width
            M = n * self.tokens
            Mp = (M + 255) // 256 * 256
            P = self.tokens - 1
            Pp = (n * P + 255) // 256 * 256
            kpad = self._prepare()["kpad"]
            ws = dict(
                x=torch.zeros(Mp, D, device=dev, dtype=torch.float32), h=torch.zeros(Mp, D, device=dev, dtype=act),
                mix=torch.zeros(Mp, D, device=dev, dtype=act), u=torch.zeros(Mp, 4 * D, device=dev, dtype=act),
                patches=torch.zeros(Pp, kpad, device=dev, dtype=act),
                delta=torch.zeros(Mp, D, device=dev, dtype=act) if self.deferred_residual else None,
                delta2=torch.zeros(Mp, D, device=dev, dtype=act) if self.deferred_residual else None, pending=0,
                qkv=[torch.zeros(Mp, 3 * D, device=dev, dtype=act) for _ in range(keep_layers)])
            if self.precision == "fp8":
                ws["h8"] = torch.zeros(Mp, D, device=dev, dtype=torch.uint8)
                ws["u8"] = torch.zeros(Mp, 4 * D, device=dev, dtype=torch.uint8)
            if len(self._ws) > 6:
                self._ws.clear()
            self._ws[key] = ws
        return ws

    def graph_guard(self, n):
        """The objects whose device buffers a captured pass over `n` frames has baked in as addresses: the prepared weights,
        the fp8 scale tables, the workspace.  A graph keeps this tuple (so none of them can be freed under it) and is
        valid only while the encoder would still use the very same objects (`is`) — `.to()`, `load_state_dict`, a new
        calibration or an evicted workspace make it stale."""
        return (self._prepare(), self._fp8_layers() if self.precision == "fp8" else None, self._workspace(n, 1, slot=0))

    # ---- kernels sequence ---------------------------------------------------------------
    def _as_frames(self, x):
        """f32 frames [N,3,R,R] already transformed, or uint8 frames [N,3,H,W] straight from the
        video reader (any H, W): those are resized / cropped / normalised by the ingest kernel."""
        if x.dtype == torch.uint8:
            assert x.dim() == 4 and x.shape[1] == 3, "uint8 frames must be [N, 3, H, W]"
            return x.contiguous()
        assert tuple(x.shape[1:]) == (3, self.input_resolution, self.input_resolution), \
            f"frames must be [N, 3, {self.input_resolution}, {self.input_resolution}]"
        return x.to(torch.float32).contiguous()

    def _embed(self, frames, ws, p):
        """conv1 as patchify + GEMM, CLS row, + positional embedding, ln_pre (model.py:277-292)."""
        n = frames.shape[0]
        P, D = self.tokens - 1, self.width
        if frames.dtype == torch.uint8:
            # raw video frames: Resize/CenterCrop/Normalize (`Detector._transform`) happen in the ingest kernel
            capi.preprocess_u8(frames, ws["patches"][:n * P], self.input_resolution, self.patch_size, self.pixel_mean,
                               self.pixel_std, antialias=self.antialias)
        else:
            capi.patchify(frames, ws["patches"], self.input_resolution, self.patch_size)
        capi.gemm(ws["patches"], p["w_patch"], ws["x"], None, capi.EPI_PATCH_EMBED, m=n * P, pos=p["pos"], cls=p["cls"],
                  tokens=self.tokens)
        ws["pending"] = 0
        # ln_pre is not run here: the first block's ln_1 does it in the same pass over the rows (`_ln`, dfd_layernorm2)
        ws["ln_pre"] = p["ln_pre"]

    def _ln(self, ws, gb, M, store=True, q=None, discard_x=False):
        """h = LayerNorm(x).  On the bf16 path the residual branches that have not been added yet
        (`ws["pending"]` of them: out_proj wrote `ws["delta"]`, c_proj `ws["delta2"]`) are folded in first
        inside the same pass over the rows (dfd_add_layernorm).  ln_2 (`store=False`) normalises
        x + delta without storing it; the next ln_1 adds both deltas and stores x once per block."""
        x = ws["x"]
        h, inv = (ws["h8"], q) if q is not None else (ws["h"], 0.0)  # q: e4m3 output with this inverse scale (fp8 path)
        pend = ws.get("pending", 0)
        pre = ws.pop("ln_pre", None)
        if pre is not None:  # first LayerNorm after the patch embedding: x <- ln_pre(x), h = ln_1(x), one pass
            assert pend == 0
            if self.width <= 2048:
                capi.layernorm2(x[:M], pre[0], pre[1], gb[0], gb[1], h[:M], out_inv_scale=inv)
            else:
                capi.layernorm(x[:M], pre[0], pre[1], x[:M])
                capi.layernorm(x[:M], gb[0], gb[1], h[:M], out_inv_scale=inv)
        elif pend == 0:
            capi.layernorm(x[:M], gb[0], gb[1], h[:M], out_inv_scale=inv)
        elif discard_x:
            # last block of an extraction pass (K and V thirds only): nothing reads the residual stream after this
            # LayerNorm, so the sum is normalised without being stored (290 MB of writes less at B16xT30)
            capi.add_layernorm(x[:M], ws["delta"][:M], gb[0], gb[1], h[:M], delta2=ws["delta2"][:M] if pend == 2 else None,
                               store_x=False, out_inv_scale=inv)
            ws["pending"] = 0
        elif not store:
            assert pend == 1
            capi.add_layernorm(x[:M], ws["delta"][:M], gb[0], gb[1], h[:M], store_x=False, out_inv_scale=inv)
        else:
            capi.add_layernorm(x[:M], ws["delta"][:M], gb[0], gb[1], h[:M], delta2=ws["delta2"][:M] if pend == 2 else None,
                               out_inv_scale=inv)
            ws["pending"] = 0

    def _residual(self, ws, a, w, b, M, spare_cus=0, spare_if_free=False):
        """x = x + Linear(a) (model.py:222-223).  fp32 path: read-modify-write of x in the GEMM
        epilogue.  bf16 path: the GEMM stores its output as a bf16 delta (plain store epilogue, a
        quarter of the epilogue bytes) and the add is deferred to the LayerNorm that follows."""
        if self.deferred_residual:
            pend = ws.get("pending", 0)
            assert pend < 2
            capi.gemm(a, w, ws["delta2" if pend else "delta"], b, capi.EPI_BIAS, m=M, stream_out=self.stream_out["proj" if pend else "out"],
                      spare_cus=spare_cus, spare_if_free=spare_if_free)
            ws["pending"] = pend + 1
        else:
            capi.gemm(a, w, ws["x"], b, capi.EPI_BIAS_RESIDUAL, m=M)

    def _flush(self, ws, M):
        """Materialise x when something other than a LayerNorm reads it next."""
        pend = ws.get("pending", 0)
        if pend >= 1:
            ws["x"][:M] += ws["delta"][:M]
        if pend == 2:
            ws["x"][:M] += ws["delta2"][:M]
        ws["pending"] = 0

    def _block(self, ws, bp, qkv, M, n, kv_only=False, export=None):
        """One residual attention block (model.py:220-226).  `export` = (k_out, v_out, tpos, T)
        makes the QKV epilogue also write the decoder operands; `kv_only` stops after the
        projection (nothing after it can reach an exported tensor)."""
        calib = getattr(self, "_calib", None)
        # fp8: chunks below the e4m3 kernel's smallest shape (1024 rows: 6 frames of ViT-B/16, 4 of ViT-L/14 — `ema_frame`
        # batches, a short last clip) run the block on the bf16 operands that calibration keeps anyway
        f8 = self._fp8_layers()[bp["idx"]] if self.precision == "fp8" and calib is None and M >= capi.FP8_MIN_ROWS else None
        so = self.stream_out
        sp = self.spare_cus if (self.spare_layers <= 0 or bp["idx"] < self.spare_layers) else 0
        sp_all = self.spare_cus if bp["idx"] < self.spare_window_layers else 0
        sg = self.spare_gemms
        sp_qkv, sp_out, sp_fc, sp_proj = (max(sp_all, sp if sg[k] else 0) for k in ("qkv", "out", "fc", "proj"))
        # outside the collective's window the spare CUs are a request: a GEMM leaves them only where that costs its shape no
        # extra round of tiles (ViT-B/16's c_proj: 5 rounds on 224 CUs as on 256; ViT-L/14's would need 5 instead of 4)
        free = sp_all == 0 and self.spare_if_free
        D = self.width
        # q | k | v projection (+ K/V export); the last tapped layer computes only the K and V thirds
        first = 1 if kv_only else 0
        rows = slice(D, None) if first else slice(None)
        kw = dict(m=M, tokens=self.tokens, qkv_first=first, stream_out=so["qkv"], spare_cus=sp_qkv, spare_if_free=free)
        if export is not None:
            kw.update(pos=export[2], k_export=export[0], v_export=export[1], frames_per_clip=export[3])
        if f8 is not None:
            self._ln(ws, bp["ln1"], M, q=f8["h1_inv"], discard_x=kv_only)
            capi.gemm_fp8(ws["h8"], bp["w_qkv8"][rows], qkv[:, rows], f8["cs_qkv"][rows], bp["b_qkv"][rows], capi.EPI_QKV_EXPORT, **kw)
        else:
            self._ln(ws, bp["ln1"], M, discard_x=kv_only and calib is None)
            if calib is not None:
                calib.append(ws["h"][:M].abs().max())
            capi.gemm(ws["h"], bp["w_qkv"][rows], qkv[:, rows], bp["b_qkv"][rows], capi.EPI_QKV_EXPORT, **kw)
        if kv_only:
            return
        capi.attention_fwd(qkv, ws["mix"], n, self.tokens, self.heads)
        self._residual(ws, ws["mix"], bp["w_out"], bp["b_out"], M, spare_cus=sp_out, spare_if_free=free)
        if f8 is not None:
            self._ln(ws, bp["ln2"], M, store=False, q=f8["h2_inv"])
            capi.gemm_fp8(ws["h8"], bp["w_fc8"], ws["u8"], f8["cs_fc"], bp["b_fc"], capi.EPI_BIAS_QUICKGELU, m=M,
                          out_inv_scale=f8["u_inv"], stream_out=so["fc"], spare_cus=sp_fc, spare_if_free=free)
            pend = ws.get("pending", 0)  # c_proj: the second deferred residual of the block (see `_residual`)
            capi.gemm_fp8(ws["u8"], bp["w_proj8"], ws["delta2" if pend else "delta"], f8["cs_proj"], bp["b_proj"], capi.EPI_BIAS, m=M,
                          stream_out=so["proj"], spare_cus=sp_proj, spare_if_free=free)
            ws["pending"] = pend + 1
        else:
            self._ln(ws, bp["ln2"], M, store=False)
            if calib is not None:
                calib.append(ws["h"][:M].abs().max())
            capi.gemm(ws["h"], bp["w_fc"], ws["u"], bp["b_fc"], capi.EPI_BIAS_QUICKGELU, m=M, stream_out=so["fc"], spare_cus=sp_fc, spare_if_free=free)
            if calib is not None:
                calib.append(ws["u"][:M].abs().max())
            self._residual(ws, ws["u"], bp["w_proj"], bp["b_proj"], M, spare_cus=sp_proj, spare_if_free=free)

    @torch.no_grad()
    def forward(self, x, with_out=False, with_q=False):
        """Reference API: frames [N,3,R,R] -> list of per-block dicts (k, v[, q][, out])."""
        if not x.is_cuda:
            raise capi.DfdError("the encoder runs on HIP kernels only: pass device tensors")
        p = self._prepare()
        frames = self._as_frames(x)
        n = frames.shape[0]
        self._ensure_fp8(frames)
        D, H, tok = self.width, self.heads, self.tokens
        M = n * tok
        ws = self._workspace(n, self.layers)
        self._embed(frames, ws, p)
        result = []
        for l, bp in enumerate(p["blocks"]):
            qkv = ws["qkv"][l]
            self._block(ws, bp, qkv, M, n)
            t = qkv[:M].view(n, tok, 3, H, 64)
            d = {"k": t[:, :, 1], "v": t[:, :, 2]}
            if with_q:
                d["q"] = t[:, :, 0]
            if with_out:
                self._flush(ws, M)
                d["out"] = ws["x"][:M].view(n, tok, D).clone()
            result.append(d)
        return result

    @torch.no_grad()
    def extract_kv(self, x, layer_indices, num_frames, temporal_pos=None, out=None, pos_ready=None, in_place=None):
        """Fused extraction for the decoder: frames [N,3,R,R] (N = B*T) ->
        (k, v): two tensors [L, N*P, D] in the activation dtype, one slab per selected layer,
        rows ordered (clip, frame, patch) — i.e. `[B, T*P, heads, 64]` per layer — with the
        temporal positional embedding `temporal_pos` [T, D] (f32) added when given.
        `pos_ready`: an event that must have fired before `temporal_pos` is read (the pipelined encoder
        stream, `Detector._encode`): waited for right before the first tapped layer's projection, so the
        layers below it do not wait.
        `in_place`: a buffer [L, N, tokens, 3*D] (activation dtype).  The tapped layers then write their q|k|v
        activation THERE instead of into the shared workspace and nothing is exported: the return value is a pair of
        strided views [L, N, P, D] (k = buffer[:, :, 1:, D:2D], v = [..., 2D:]) WITHOUT the positional embedding —
        the decoder's attention kernels read them in place and add it on the fly (`Decoder.run((k, v, pos), m)`).
        Saves the export's writes (2/3 of a projection's output per tapped layer)."""
        if not x.is_cuda:
            raise capi.DfdError("the encoder runs on HIP kernels only: pass device tensors")
        p = self._prepare()
        frames = self._as_frames(x)
        n = frames.shape[0]
        assert n % num_frames == 0
        self._ensure_fp8(frames)
        D, tok = self.width, self.tokens
        P = tok - 1
        L = len(layer_indices)
        if in_place is not None:
            assert in_place.shape == (L, n, tok, 3 * D) and in_place.dtype == self.act_dtype and in_place.is_contiguous()
            k_out = v_out = None
        elif out is None:
            k_out = torch.empty(L, n * P, D, device=frames.device, dtype=self.act_dtype)
            v_out = torch.empty_like(k_out)
        else:
            k_out, v_out = out
        last = max(layer_indices)
        slot = {l: i for i, l in enumerate(layer_indices)}
        clips = n // num_frames
        chunk = self.frame_chunk if self.frame_chunk > 0 else -(-clips // max(1, self.streams)) * num_frames
        chunk = max(num_frames, chunk // num_frames * num_frames)  # whole clips keep (frame % T) aligned
        starts = list(range(0, n, chunk))
        # chunks are independent: with `streams` > 1 they are issued round-robin on side streams, so one
        # chunk's kernels fill the idle CUs of another chunk's kernel tails (e.g. out_proj = 4.3 tile waves)
        n_str = min(self.streams, len(starts))
        if n_str > 1:
            cur = torch.cuda.current_stream()
            if len(self._side_streams) < n_str:
                self._side_streams = [torch.cuda.Stream() for _ in range(n_str)]
            for st in self._side_streams[:n_str]:
                st.wait_stream(cur)
        for ci, f0 in enumerate(starts):
            nf = min(chunk, n - f0)
            ctx = torch.cuda.stream(self._side_streams[ci % n_str]) if n_str > 1 else contextlib.nullcontext()
            with ctx:
                ws = self._workspace(nf, 1, slot=(ci % n_str) if n_str > 1 else 0)
                qkv = ws["qkv"][0]
                M = nf * tok
                self._embed(frames[f0:f0 + nf], ws, p)
                waited = pos_ready is None or temporal_pos is None or in_place is not None
                for l in range(last + 1):
                    exp = None
                    qkv_l = qkv
                    if l in slot and in_place is not None:
                        qkv_l = in_place[slot[l]].view(n * tok, 3 * D)[f0 * tok:(f0 + nf) * tok]
                    elif l in slot:
                        i = slot[l]
                        if not waited:
                            torch.cuda.current_stream().wait_event(pos_ready)
                            waited = True
                        exp = (k_out[i, f0 * P:(f0 + nf) * P], v_out[i, f0 * P:(f0 + nf) * P], temporal_pos, num_frames)
                    self._block(ws, p["blocks"][l], qkv_l, M, nf, kv_only=(l == last), export=exp)
        if n_str > 1:
            for st in self._side_streams[:n_str]:
                cur.wait_stream(st)
        if in_place is not None:
            return in_place[:, :, 1:, D:2 * D], in_place[:, :, 1:, 2 * D:]
        return k_out, v_out
