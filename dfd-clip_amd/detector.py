"""`Detector`: the drop-in model boundary (reference `src/models.py:394-780`).

Same constructor, attributes, `predict` / `forward` / `configure_optimizers` contracts and
state_dict key schema as the reference class, so reference-style trainer / evaluator /
inference loops (`src/trainer.py:147-151`, `src/evaluator.py:80-83`, `inference.py:115-118`)
run unchanged on top of it.  The frozen encoder and the decoder execute in
libdfdclip_hip.so; PyTorch supplies device memory, streams and the tiny per-sample loss.

    Detector(config, num_frames, accelerator, precision="bf16" | "fp32" | "fp8")

`precision` is the only addition to the reference signature: "fp32" is the parity path
(logits within 1e-3 of the reference's fp32 CPU result), "bf16" the throughput path, "fp8" the bf16 path
with the encoder's large projections on e4m3 matrix-core operands (BASELINE configs[4]; `calibrate_fp8`).
"""
import contextlib
import logging
import os
import random
from itertools import combinations
from math import comb

import numpy as np

import torch
from torch import nn

from . import capi
from .config import default_detector_config
from .adapter import CompInvAdapter
from .decoder import Decoder
from .encoder import RuntimeStateMixin, VisionTransformer
from .weights import ARCHS, resolve_layer_indices

_ENC_STREAMS = {}  # device -> the process's high-priority encoder stream (`Detector._encode`)

CLIP_CACHE = os.path.expanduser("~/.cache/clip")  # where the reference's downloader leaves checkpoints (clip/clip.py:94)


class TaskLoss:
    """Unreduced loss of one task head: `loss(logits [B, out_dim], labels) -> [B]`.  The trainer takes the mean itself
    (reference `src/trainer.py:154-155`); the config names a loss by the reference's factory name, optionally with
    `args` (reference `src/models.py:447-452`)."""

    def __call__(self, logits, labels):  # pragma: no cover
        raise NotImplementedError


class PerSampleCrossEntropy(TaskLoss):
    """`auc_roc` (reference `src/models.py:34-45`): cross entropy per sample, optional class weights and label smoothing."""

    def __init__(self, weight=None, label_smoothing=0.0, **_unused):
        self.weight = list(weight) if weight else None
        self.label_smoothing = float(label_smoothing)

    def __call__(self, logits, labels):
        w = None if self.weight is None else torch.tensor(self.weight, device=logits.device)
        return torch.nn.functional.cross_entropy(logits, labels, weight=w, label_smoothing=self.label_smoothing, reduction="none")


class PerElementKLDiv(TaskLoss):
    """`kl_div` (reference `src/models.py:28-31`): KL(labels || softmax(logits)), element-wise."""

    def __init__(self, **_unused):
        pass

    def __call__(self, logits, labels):
        return torch.nn.functional.kl_div(logits.log_softmax(dim=1), labels, reduction="none")


class ExpectationMSE(TaskLoss):
    """`mse` (reference `src/models.py:20-25`): squared error of the expected bin index over the first 140 logits, / 1000."""
    BINS = 140

    def __init__(self, **_unused):
        pass

    def __call__(self, logits, labels):
        bins = torch.arange(self.BINS, dtype=torch.float32, device=logits.device)
        expected = logits[:, :self.BINS].softmax(dim=-1) @ bins
        return (expected - labels) ** 2 / 1000


TASK_LOSSES = {"auc_roc": PerSampleCrossEntropy, "kl_div": PerElementKLDiv, "mse": ExpectationMSE}


def make_task_loss(spec):
    """`spec`: a factory name, or a node {name, args} (the two forms the reference's configs use)."""
    if isinstance(spec, str):
        return TASK_LOSSES[spec]()
    return TASK_LOSSES[spec.name](**(dict(spec.args) if "args" in spec else {}))


def disable_gradients(module):
    for p in module.parameters():
        p.requires_grad = False
    return module


def _infer_arch_from_state_dict(sd):
    """What `build_model` reads off a checkpoint (reference `src/clip/model.py:453-470`)."""
    width = sd["conv1.weight"].shape[0]
    patch = sd["conv1.weight"].shape[-1]
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    grid = round((sd["positional_embedding"].shape[0] - 1) ** 0.5)
    return grid * patch, patch, width, layers, width // 64, sd["proj"].shape[1]


def _is_torchscript_archive(path):
    """The published CLIP checkpoints (ViT-B-16.pt, ...) are TorchScript archives: a zip whose root folder holds
    `constants.pkl` next to `data.pkl` (and `code/`).  A plain `torch.save` state_dict has `data.pkl` only."""
    import zipfile
    if not zipfile.is_zipfile(path):
        return False
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
    return any(n.endswith("/constants.pkl") or "/code/" in n for n in names)


def _halved_by_convert_weights(key):
    """Parameters the reference's `convert_weights` stores in fp16 before `load_state_dict` copies the checkpoint
    in (reference `src/clip/model.py:429-450`): Conv/Linear weights and biases and `proj`.  Its custom attention
    class is not `nn.MultiheadAttention`, so `in_proj_weight` / `in_proj_bias` stay fp32, as do LayerNorms and
    the class / positional embeddings."""
    return key == "conv1.weight" or key == "proj" or ".out_proj." in key or ".mlp.c_fc." in key or ".mlp.c_proj." in key


def load_clip_visual(name, precision):
    """Counterpart of `clip.load(name)[0].visual.float()` (reference `src/models.py:440`,
    `src/clip/clip.py:94-142`, `src/clip/model.py:453-496`).  The reference downloads the checkpoint by name;
    there is no network here, so: a path to a state_dict checkpoint (or `~/.cache/clip/<name>.pt`, where the
    reference's downloader leaves it) is loaded with `weights_only=True`; otherwise the architecture is built
    with its seeded random initialisation and a warning is logged.

    The architecture is read off the tensor shapes as `build_model` does, and the weights take the same path
    through fp16 as in the reference: `convert_weights` halves the Conv/Linear parameters and `proj` before the
    checkpoint is copied in, `.float()` widens them again, so an fp32 checkpoint loses those mantissa bits here
    too (an fp16 checkpoint, the published form, is unchanged by it)."""
    path = name if os.path.isfile(name) else os.path.join(CLIP_CACHE, name.replace("/", "-") + ".pt")
    if os.path.isfile(path):
        if _is_torchscript_archive(path):
            raise RuntimeError(
                f"{path} is a TorchScript archive (the form OpenAI publishes); this loader reads plain state_dict "
                "checkpoints only (torch.load(weights_only=True) executes nothing from the file).  Convert it once, "
                "offline, where you trust the file:  torch.save(torch.jit.load(path, map_location='cpu').state_dict(), "
                "new_path)  and point `architecture` at new_path.")
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
            raise RuntimeError(f"{path}: expected a state_dict (name -> tensor), got {type(sd).__name__}")
        sd = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")} or sd
        if "conv1.weight" not in sd or "proj" not in sd:
            raise RuntimeError(f"{path}: no CLIP ViT visual tower in this checkpoint (ResNet towers are not built)")
        vit = VisionTransformer(*_infer_arch_from_state_dict(sd), precision=precision)
        vit.load_state_dict({k: (v.half().float() if _halved_by_convert_weights(k) else v.float()) for k, v in sd.items()})
        return vit
    if name not in ARCHS:
        raise RuntimeError(f"Model {name} not found; available architectures = {list(ARCHS)}")
    logging.warning("no checkpoint for %s: encoder uses its random initialisation", name)
    return VisionTransformer(*ARCHS[name], precision=precision)


class ClipTransform:
    """Resize(bicubic, shorter side) -> CenterCrop -> float in [0,1] -> Normalize (reference
    `src/models.py:756-768`), on tensors [T, 3, H, W] uint8 or float.

    uint8 frames already on the GPU go through the ingest kernel (`dfd_preprocess_u8`); anything
    else takes the same arithmetic through torch ops (dataloader workers on the host).  Geometry
    and rounding follow torchvision's tensor path: longer side int(n_px*long/short), a resized
    uint8 image is rounded back to uint8, crop origin round((size - n_px)/2)."""
    MEAN = (0.48145466, 0.4578275, 0.40821073)
    STD = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, n_px, antialias=True):
        self.n_px = n_px
        self.antialias = antialias

    def geometry(self, h, w):
        s, l = (h, w) if h <= w else (w, h)
        new_l = int(self.n_px * l / s)
        nh, nw = (self.n_px, new_l) if h <= w else (new_l, self.n_px)
        return nh, nw, int(round((nh - self.n_px) / 2.0)), int(round((nw - self.n_px) / 2.0))

    def __call__(self, frames):
        x = frames
        if x.is_cuda and x.dtype == torch.uint8:
            lead = x.shape[:-3]
            out = torch.empty(x.numel() // (3 * x.shape[-2] * x.shape[-1]), 3, self.n_px, self.n_px, device=x.device)
            tile = max(d for d in range(1, 33) if self.n_px % d == 0)  # work tile of the kernel, any divisor
            capi.preprocess_u8(x.reshape(-1, *x.shape[-3:]).contiguous(), out, self.n_px, tile, self.MEAN, self.STD,
                               antialias=self.antialias, patch_rows=False)
            return out.view(*lead, 3, self.n_px, self.n_px)
        was_u8 = x.dtype == torch.uint8
        h, w = x.shape[-2:]
        nh, nw, top, left = self.geometry(h, w)
        if (nh, nw) != (h, w):
            x = torch.nn.functional.interpolate(x.float(), size=(nh, nw), mode="bicubic", antialias=self.antialias,
                                                align_corners=False)
            if was_u8:
                x = x.round().clamp(0, 255)
        x = x[..., top:top + self.n_px, left:left + self.n_px]
        x = x.float() / 255.0 if was_u8 else x
        mean = torch.tensor(self.MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(self.STD, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std


class Detector(RuntimeStateMixin, nn.Module):
    _RUNTIME_STATE = {"_kv_static": None, "_kv_cache": {}, "_enc_stream": None, "_pipe_events": [[], []], "_pipe_step": 0, "_pos_snap": None,
                      "_drop_master": None, "_enc_graphs": {}, "_enc_graph_seen": {}, "_enc_graphs_failed": None, "_all_params": None}

    def _apply(self, fn, *a, **k):
        self._all_params = None  # (.to() / .half() may re-create parameter objects)
        return super()._apply(fn, *a, **k)

    def zero_grad(self, set_to_none=True):
        """`nn.Module.zero_grad` without the walk over the module tree: the parameter objects are created once in
        `__init__` (the list is rebuilt by `_apply` / `load_state_dict` / `invalidate_caches`), and collecting the ~230 of
        them through `named_parameters` every call cost 0.6 ms of host time — the reference's trainer calls
        `model.zero_grad()` twice per step (`src/trainer.py:110`, `:177`).  Every parameter is still looked at."""
        if self._all_params is None:
            self._all_params = list(self.parameters())
        for p in self._all_params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.requires_grad_(False)
                    p.grad.zero_()

    def invalidate_caches(self):
        """Drop every device-side copy derived from the parameters (bf16 encoder weights, transposed decoder
        weights, adapter operands, captured graphs).  `load_state_dict` / `.to()` / optimizer steps are tracked
        automatically; call this after writing parameters in place by other means (`dist.broadcast_parameters`)."""
        self.encoder.invalidate()
        self.decoder.invalidate_caches()
        self._enc_graphs, self._enc_graph_seen = {}, {}
        self._all_params = None
        if self.adapter is not None:
            self.adapter.invalidate_caches()

    @staticmethod
    def get_default_config():
        return default_detector_config()

    def __init__(self, config, num_frames, accelerator=None, precision="bf16"):
        super().__init__()
        assert config.decode_mode in ["stride", "index"]
        capi.load_library()  # fail at construction, not at first forward, when the kernels are missing
        self.config = config
        self.precision = precision
        self.num_frames = num_frames
        if config.foundation != "clip":
            raise NotImplementedError("only the CLIP foundation is built (DINOv2 is out of scope)")
        ctx = accelerator.main_process_first() if accelerator is not None else contextlib.nullcontext()
        with ctx:
            self.encoder = disable_gradients(load_clip_visual(config.architecture, precision))
        self.decode_mode = config.decode_mode
        self.out_dim = config.out_dim
        self.weight_decay = config.weight_decay
        self.optimizer = config.optimizer
        self.train_mode = config.train_mode
        self.op_mode = config.op_mode
        self.losses = [make_task_loss(spec) for spec in config.losses]
        self.layer_indices = resolve_layer_indices(config, len(self.encoder.transformer.resblocks))
        self.decoder = Decoder(self, config, num_frames)
        self.adapter = self._build_adapter(config, num_frames)
        self.transform = ClipTransform(self.encoder.input_resolution)
        # opt-in: replay the decoder's training-step kernels as HIP graphs (fixed batch shape; see decoder.py)
        self.static_graphs = False
        self._kv_static = None
        self._kv_cache = {}  # K/V buffer sets of earlier batch signatures (see `_encode`)
        self.kv_cache_sets = 4  # signatures whose sets stay allocated: train / eval x full batch / last batch of an epoch
        # opt-in: run the frozen encoder on its own stream so that step N+1's encoder pass overlaps step N's
        # decoder backward / all-reduce / optimizer (see `predict`); `inputs_ready` = the caller guarantees that
        # the clips handed to forward() are already complete in device memory
        self.pipeline_encoder = False
        self.kv_in_place = True  # without an adapter the decoder reads K/V out of the q|k|v activations (see `_encode`)
        self.inputs_ready = False
        self.pipeline_spare_cus = None  # None = one compute unit per shader engine (CUs / 8), see `_encode`
        self.pipeline_spare_layers = 0  # encoder blocks at the start of a pass that leave those CUs free (0 = all); which
                                        # of a block's GEMMs do is `encoder.spare_gemms` (c_proj only: see there)
        # Multi-GPU (world size > 1): in the first `pipeline_collective_layers` blocks of a pipelined training pass EVERY
        # encoder GEMM leaves the spare CUs free, not only c_proj — that is where the previous step's gradient all-reduce
        # (one RCCL kernel of at most NCCL_MAX_NCHANNELS workgroups) runs beside the encoder; set by the launcher / bench
        self.pipeline_collective_layers = 0
        # pipelined inference: c_proj leaves the spare CUs to the previous batch's decoder kernels too (5 rounds of tiles on
        # 224 CUs as on 256, so it is free for the encoder; forward 912 -> 917.6 clips/s, alternating runs on one box)
        self.pipeline_spare_in_eval = True
        self._enc_stream = None
        self._enc_graphs, self._enc_graph_seen, self._enc_graphs_failed = {}, {}, None
        self.encoder_graph_pause = False  # True: launch the encoder's kernels one by one (per-kernel HIP-event timing, `bench.py`)
        self._all_params = None
        self._pipe_events = [[], []]
        self._pipe_step = 0
        self._pos_snap = None
        # train-mode dropout (reference models.py:163, :294, :304, :804-912; every configs/deepfake/*.yaml sets 0.5):
        # counter-based masks from a device-resident {seed, step}; see `seed_dropout`
        self.dropout_p = float(config.dropout) if "dropout" in config else 0.0
        self._drop_seed = None
        self._drop_master = None
        # trainable extras (reference models.py:488-496)
        if "temporal" in self.train_mode and self.train_mode.temporal == "ranking":
            self.ranking_transform_param = nn.Parameter((self.encoder.width ** -0.5) * torch.randn(self.encoder.width, 1),
                                                        requires_grad=True)
        for key in ("compression", "nerf_raw"):
            if key in self.train_mode:
                # unreachable upstream: the reference's compression branch unpacks five dims from K/V that its decoder
                # has already flattened to four (models.py:604 -> ValueError) and nerf_raw reads `_b`, defined only
                # there (models.py:673 -> NameError); there is no behaviour to match
                raise NotImplementedError(f"train_mode.{key}: this branch raises in the reference itself (models.py:604, :673)")
        if "patch_mask" in self.train_mode:
            if self.train_mode.patch_mask.type not in ("batch", "sample"):
                # "guide" reads a pickled saliency map (models.py:494-496): unpickling foreign files is not done here
                raise NotImplementedError(f"patch_mask.type={self.train_mode.patch_mask.type} is not built")
            if self.adapter is not None and self.adapter.struct.endswith("nln"):
                raise NotImplementedError("patch_mask with the nln adapter: its LayerNorm is sized for all patches "
                                          "(the reference fails on this combination too)")

    @torch.no_grad()
    def calibrate_fp8(self, x, margin=1.0):
        """fp8 path: fix the encoder's static activation scales from representative clips x [B,T,3,R,R] (or frames
        [N,3,R,R]); see `VisionTransformer.calibrate_fp8`.  Without it the first forward calibrates on its own batch."""
        return self.encoder.calibrate_fp8(x.flatten(0, 1) if x.dim() == 5 else x, margin=margin)

    def seed_dropout(self, seed):
        """Fix the dropout stream: the same seed (and the same number of training forwards since) gives the
        same masks.  Unseeded, the stream starts from `torch.initial_seed()` (so `torch.manual_seed` before the
        first training step makes a run repeatable); the process rank is folded in so ranks draw different masks."""
        self._drop_seed = int(seed)
        self._drop_master = None

    def _next_drop_rng(self, device):
        """This forward's dropout state (device int64 {seed, step}) — or None outside train() / with p == 0 — and
        advance the step counter.  The decoder and the adapter of one forward share it (distinct sites)."""
        if not self.training or self.dropout_p <= 0:
            return None
        if self._drop_master is None or self._drop_master.device != device:
            from . import dist as ddist
            seed = (self._drop_seed if self._drop_seed is not None else torch.initial_seed()) + 0x9E3779B97F4A7C15 * ddist.rank()
            self._drop_master = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        snap = self._drop_master.clone()
        self._drop_master[1] += 1
        return snap

    def _encode(self, x, t, pos, keep, allow_in_place=True):
        """Encoder pass over the clips -> ((k, v) export, pipeline context or None).

        `keep` / `pipeline_encoder`: the export lands in persistent buffers (what a previous step returned is
        overwritten by a later one; only in these opt-in modes).  Pipelined: the frozen encoder does not depend
        on the trainable parameters, so the encoder pass of batch N+1 may run while batch N's decoder (and
        adapter) backward, gradient all-reduce and optimizer step are still executing.  It gets its own
        high-priority HIP stream and alternates between two export sets.  It waits for (a) the last readers of
        the set it is about to overwrite (forward / backward of two batches ago), (b) the caller's stream,
        unless the caller vouches that the inputs are ready (`inputs_ready`: static or prefetched batches) —
        without that the wait includes the previous step's tail and nothing overlaps."""
        b = x.shape[0]
        pipelined = bool(self.pipeline_encoder)
        out = None
        # K/V in place (`kv_in_place`, the default without an adapter): the tapped layers keep their q|k|v activation
        # in a buffer of their own and the decoder's attention kernels read keys and values straight out of it (CLS
        # row skipped, positional embedding added on the fly) — the projection writes no export: -74 us per tapped
        # layer at B16xT30 and 1.7 GB of HBM writes less per step.  The buffers are persistent (what a previous call
        # returned is overwritten by a later one, two sets alternate when pipelined).
        in_place = bool(self.kv_in_place) and allow_in_place
        # persistent buffers: the opt-in modes, and in-place inference (nothing outlives the call there); an in-place
        # TRAINING forward outside those modes gets a buffer of its own, because its autograd node reads it later
        if keep or pipelined or (in_place and not torch.is_grad_enabled()):
            # training and inference keep separate sets: an evaluation pass between two epochs must not evict (and so
            # re-address) the buffers the training graphs were captured on
            key = (b, t, tuple(x.shape[-2:]), x.dtype, pipelined, pos is None, in_place, bool(torch.is_grad_enabled() and self.training))
            if self._kv_static is None or self._kv_static[0] != key:
                P_ = (self.encoder.input_resolution // self.encoder.patch_size) ** 2
                if in_place:
                    shape = (len(self.layer_indices), b * t, P_ + 1, 3 * self.encoder.width)
                    new_set = lambda: torch.empty(shape, device=x.device, dtype=self.encoder.act_dtype)
                else:
                    shape = (len(self.layer_indices), b * t * P_, self.encoder.width)
                    new_set = lambda: (torch.empty(shape, device=x.device, dtype=self.encoder.act_dtype),
                                       torch.empty(shape, device=x.device, dtype=self.encoder.act_dtype))
                if self._kv_static is not None and (pipelined or self._kv_static[0][4]):
                    # the batch shape or the mode changed (the last batch of an epoch, train <-> eval) around the pipelined
                    # path: the readers of the old sets are tracked by events that are dropped below, and a set evicted
                    # here may be handed back by the allocator and written by the encoder stream — let the device
                    # finish first (a few times per epoch, so a full drain is fine)
                    torch.cuda.synchronize()
                # the sets of the most recent signatures are kept (train / eval x full / last batch of an epoch): going back
                # to one finds its buffers at the same addresses, so the decoder's captured graphs for it stay valid epoch
                # after epoch; what is evicted takes its graphs with it
                if self._kv_static is not None:
                    self._kv_cache[self._kv_static[0]] = self._kv_static[1]
                sets = self._kv_cache.pop(key, None)
                while len(self._kv_cache) > self.kv_cache_sets - 1:
                    gone = self._kv_cache.pop(next(iter(self._kv_cache)))
                    ptrs = set()
                    for s_ in gone:
                        for t_ in (s_ if isinstance(s_, tuple) else (s_,)):
                            ptrs.add(t_.data_ptr())
                            if t_.dim() == 4:  # in-place sets: the decoder sees the K and V thirds as strided views
                                D_ = t_.shape[-1] // 3
                                ptrs.add(t_[:, :, 1:, D_:2 * D_].data_ptr())
                                ptrs.add(t_[:, :, 1:, 2 * D_:].data_ptr())
                    self.decoder.drop_graphs_for(ptrs)
                self._kv_static = None
                self._kv_static = (key, sets if sets is not None else ([new_set(), new_set()] if pipelined else [new_set()]))
                self._pipe_events = [[], []]
                self._pipe_step = 0
            slot = self._pipe_step % len(self._kv_static[1])
            self._pipe_step += 1
            out = self._kv_static[1][slot]
        if in_place:
            if out is None:
                P_ = (self.encoder.input_resolution // self.encoder.patch_size) ** 2
                out = torch.empty(len(self.layer_indices), b * t, P_ + 1, 3 * self.encoder.width, device=x.device,
                                  dtype=self.encoder.act_dtype)
            kw = dict(in_place=out)
            enc_pos = None   # the decoder reads the live parameter on its own stream: no snapshot, no race
        else:
            kw = dict(out=out)
            enc_pos = pos
        dec_pos = pos
        finish = (lambda kv: (kv[0], kv[1], dec_pos)) if in_place else (lambda kv: kv)
        if not pipelined:
            return finish(self.encoder.extract_kv(x.flatten(0, 1), self.layer_indices, t, enc_pos, **kw)), None
        cur = torch.cuda.current_stream()
        if self._enc_stream is None:
            # high priority: the encoder's GEMM workgroups are dispatched first, the decoder's small
            # latency-bound kernels take what is left over (tile-round tails).  ONE such stream per device for all
            # detectors of the process: with a stream of its own per model, the third model built in a process (bench.py's
            # secondary legs) ran its pipelined pass 19 % slower than alone, 30.7 against 25.7 ms, as if its two streams
            # no longer overlapped (HIP maps streams onto a few hardware queues; measured with tools/lab/sec_fp8_probe.py)
            self._enc_stream = _ENC_STREAMS.setdefault((x.device.type, x.device.index), None) or torch.cuda.Stream(device=x.device, priority=-1)
            _ENC_STREAMS[(x.device.type, x.device.index)] = self._enc_stream
        E = self._enc_stream
        for ev in self._pipe_events[slot]:
            E.wait_event(ev)
        if not self.inputs_ready:
            E.wait_stream(cur)
        pos_ready = None
        pos = enc_pos
        if pos is not None:
            # `pos` is a view of a TRAINABLE parameter and stream E does not wait for the caller's stream
            # (inputs_ready): read live, step N+1's export could see a positional embedding that step N's
            # optimizer is still writing.  So the value is snapshotted on the caller's stream — after everything
            # already queued there, i.e. after the previous optimizer step, which is the value the reference
            # reads — and E waits for that copy only where it first needs it, at the first tapped layer's
            # projection: the layers below it still overlap the previous step's backward and optimizer.
            if self._pos_snap is None or self._pos_snap[0].shape != pos.shape or self._pos_snap[0].device != pos.device:
                self._pos_snap = [torch.empty_like(pos), torch.empty_like(pos)]
            snap = self._pos_snap[slot]
            snap.copy_(pos)
            pos_ready = torch.cuda.Event()
            pos_ready.record(cur)
            pos = snap
        # While a training step's backward / optimizer is what runs beside this pass, the persistent c_proj GEMM of every
        # block leaves one compute unit per shader engine free (MI355X: 32 of 256): the ~450 small dependent kernels of
        # the decoder then get a 0.4 ms window per block without waiting for a whole GEMM to end (round 2: every GEMM of
        # the first four blocks did; c_proj alone, in every block, is 1.3 % faster and leaves c_fc its CUs).  Measured (B16xT30 train step): 0-30 spare CUs 22.4-22.8 ms, 32-34 spare
        # 20.9-21.0 ms, 40+ 22.0 ms (the encoder loses more than the overlap returns) — the workgroup dispatcher deals
        # workgroups to shader engines in turn, so a small kernel stalls as soon as ONE engine has no free CU.
        spare = self.pipeline_spare_cus
        if spare is None:
            spare = torch.cuda.get_device_properties(x.device).multi_processor_count // 8
        self.encoder.spare_cus = spare if (torch.is_grad_enabled() and self.training) or self.pipeline_spare_in_eval else 0
        self.encoder.spare_layers = self.pipeline_spare_layers
        self.encoder.spare_window_layers = self.pipeline_collective_layers if torch.is_grad_enabled() and self.training else 0
        self.encoder.spare_if_free = not (torch.is_grad_enabled() and self.training)
        try:
            with torch.cuda.stream(E):
                frames = x.flatten(0, 1)
                if not (in_place and self.static_graphs and self._replay_encoder(frames, t, out)):
                    self.encoder.extract_kv(frames, self.layer_indices, t, pos, pos_ready=pos_ready, **kw)
                D_ = self.encoder.width
                kv = finish((out[:, :, 1:, D_:2 * D_], out[:, :, 1:, 2 * D_:]) if in_place else out)
        finally:
            self.encoder.spare_cus = 0
        x.record_stream(E)
        cur.wait_stream(E)
        events = [torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()]  # forward, decoder bwd, adapter bwd
        self._pipe_events[slot] = events
        return kv, (cur, *events)

    def _replay_encoder(self, frames, t, out):
        """The frozen encoder's pass as ONE HIP graph launch (static_graphs + pipelined + K/V in place): ~110 kernel
        launches per pass cost 1.2-1.4 ms of host time, and a training step whose host falls behind (a loaded machine:
        enqueue 6 -> 9 ms) was measured to lose 13 % of its throughput to the bubbles.  A graph is captured for an
        (input address, K/V set, launch policy) signature the second time it is seen in a row of calls — a loader that hands
        out a new buffer every step never repeats one and stays on eager launches; static or recycled batches (what
        `inputs_ready` is for) replay.  Nothing in the pass depends on host state: weights frozen, no dropout, no waits
        (the positional embedding is added by the decoder in this mode).  Returns False when the caller has to launch."""
        enc = self.encoder
        if self._enc_graphs_failed or self.encoder_graph_pause or enc.streams != 1:
            return False
        if enc.frame_chunk:
            return False
        key = (frames.data_ptr(), tuple(frames.shape), frames.dtype, out.data_ptr(), t, tuple(self.layer_indices),
               enc.precision, enc.spare_cus, enc.spare_layers, enc.spare_window_layers, enc.spare_if_free, enc.deferred_residual,
               tuple(sorted(enc.stream_out.items())), tuple(sorted(enc.spare_gemms.items())))
        guard = enc.graph_guard(frames.shape[0])
        ent = self._enc_graphs.get(key)
        if ent is not None and not all(a is b for a, b in zip(ent[1], guard)):
            torch.cuda.synchronize()  # (a replay may still be executing on the buffers the entry keeps alive)
            del self._enc_graphs[key]
            ent = None
        g = None if ent is None else ent[0]
        if g is None:
            seen = self._enc_graph_seen.get(key, 0) + 1
            self._enc_graph_seen = {key: seen} if len(self._enc_graph_seen) > 16 else {**self._enc_graph_seen, key: seen}
            if seen < 2:
                return False  # (this eager pass is also the warm-up a capture needs: lazy initialisations, fp8 calibration)
            while len(self._enc_graphs) >= 4:
                torch.cuda.synchronize()  # a replay of the entry may still be executing
                self._enc_graphs.pop(next(iter(self._enc_graphs)))
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    enc.extract_kv(frames, self.layer_indices, t, None, in_place=out)
            except Exception as e:  # capture is an optimisation: a runtime that refuses it leaves the eager launches
                self._enc_graphs_failed = f"{type(e).__name__}: {e}"
                logging.warning("encoder: HIP graph capture failed, staying on eager launches (%s)", self._enc_graphs_failed)
                torch.cuda.synchronize()
                return False
            self._enc_graphs[key] = (g, guard)
        else:
            self._enc_graphs[key] = self._enc_graphs.pop(key)  # most recently used last
        g.replay()
        return True

    def predict(self, x, m, with_video_features=False, with_adapt_features=False, train=False):
        """x [B,T,3,R,R], m [B,T] bool -> (task_logits list of [B,out_dim] with L2 norm 5, features)."""
        b, t, c, h, w = x.shape
        if t != self.num_frames and self.decoder.positional_embedding is not None:
            raise RuntimeError(f"The size of tensor a ({t}) must match the size of tensor b ({self.num_frames}) "
                               "at non-singleton dimension 1 (temporal positional embedding)")
        # the encoder is frozen and runs without autograd (reference models.py:440, :501); the decoder is
        # differentiable w.r.t. its own parameters
        pos = self.decoder.temporal_pos()
        self.decoder.use_graphs = False
        if self.adapter is not None:
            self.adapter.use_graphs = False
        pipe = None
        drop_rng = self._next_drop_rng(x.device)
        masked = train and "patch_mask" in self.train_mode
        if self.adapter is None and not masked:
            self.decoder.use_graphs = bool(self.static_graphs and train)
            kv, pipe = self._encode(x, t, pos, keep=bool(self.static_graphs and train))
        elif masked:
            # keep a random subset of patch positions per layer (models.py:511-544): "batch" draws once for all
            # layers, "sample" per layer.  The raw export is row-gathered, then adapter / positional add follow.
            kr, vr = self.encoder.extract_kv(x.flatten(0, 1), self.layer_indices, t, None)
            P = kr.shape[1] // (b * t)
            num_select = int(P * self.train_mode.patch_mask.ratio)
            ks, vs, idx = [], [], None
            for i in range(len(self.layer_indices)):
                if idx is None or self.train_mode.patch_mask.type == "sample":
                    idx = torch.as_tensor(np.random.choice(range(P), num_select, replace=False), device=kr.device)
                ks.append(kr[i].view(b * t, P, -1).index_select(1, idx).reshape(b * t * num_select, -1))
                vs.append(vr[i].view(b * t, P, -1).index_select(1, idx).reshape(b * t * num_select, -1))
            kr, vr = torch.stack(ks), torch.stack(vs)
            if self.adapter is not None:
                self.adapter.patches = num_select
                try:
                    kv = self.adapter.run(kr, vr, t, pos, drop_rng)
                finally:
                    self.adapter.patches = (self.encoder.input_resolution // self.encoder.patch_size) ** 2
            else:
                if pos is not None:  # the decoder's positional add on the gathered rows (models.py:326-329)
                    pp = pos.to(kr.dtype).view(1, 1, t, 1, -1)
                    kr = (kr.view(len(ks), b, t, num_select, -1) + pp).view_as(kr)
                    vr = (vr.view(len(vs), b, t, num_select, -1) + pp).view_as(vr)
                kv = (kr.contiguous(), vr.contiguous())
        else:
            # raw K/V export, then adapter(kv) + pos (models.py:546-549, :326-329); differentiable w.r.t. the
            # adapter's parameters when they are trainable
            graphs = bool(self.static_graphs and train and torch.is_grad_enabled())
            self.adapter.use_graphs = self.decoder.use_graphs = graphs
            kv, pipe = self._encode(x, t, None, keep=graphs, allow_in_place=False)
            if pipe is not None:
                self.adapter._after_backward = pipe[3].record  # its backward is the last reader of the raw export
            kv = self.adapter.run(kv[0], kv[1], t, pos, drop_rng)
        if pipe is not None:
            self.decoder._after_backward = pipe[2].record  # recorded on the backward's stream when it ends
        _, video_features, task_logits = self.decoder.run(kv, m, drop_rng)
        if pipe is not None:
            pipe[1].record(pipe[0])
        features = {}
        if with_video_features:
            features["video"] = video_features
        if with_adapt_features:
            if self.adapter:
                # as in the reference, what comes back has been flattened and pos-embedded by the decoder
                # (models.py:329-334 mutate the dicts in place)
                hh = self.encoder.heads
                features["adapt"] = [{"k": kv[0][i].view(b, -1, hh, 64), "v": kv[1][i].view(b, -1, hh, 64)}
                                     for i in range(len(self.layer_indices))]
            else:
                raise Exception("cannot return adaptive features without an adapter")
        return task_logits, features

    def _build_adapter(self, config, num_frames):
        """`adapter.type` none / normal (fresh) / pretrain (parameters from a checkpoint of a whole Detector, optionally
        frozen): reference `src/models.py:454-478`.  Checkpoints are read with `weights_only=True`."""
        kind = config.adapter.type
        if kind == "none":
            return None
        if kind not in ("normal", "pretrain"):
            raise NotImplementedError(f"adapter.type = {kind!r}")
        adapter = CompInvAdapter(config, self, num_frames=num_frames)
        if kind == "pretrain":
            ckpt = torch.load(config.adapter.path, map_location="cpu", weights_only=True)
            adapter.load_state_dict({k.split(".", 1)[1]: v for k, v in ckpt.items() if "adapter" in k})
            if config.adapter.frozen:
                adapter = disable_gradients(adapter)
        logging.info("adapter %s: %s", config.adapter.struct.type, "fresh parameters" if kind == "normal" else f"parameters of {config.adapter.path}")
        return adapter

    def forward(self, x, y, m, comp=None, speed=None, train=False, single_task=None, *args, **kargs):
        b, t, c, h, w = x.shape
        if "ema_frame" in self.op_mode and self.op_mode.ema_frame:
            if x.dtype == torch.uint8:
                x = self.transform(x)
                h, w = x.shape[-2:]
            # exponential moving average of the frames -> one frame per clip (models.py:572-578)
            r = self.op_mode.ema_frame
            _x = torch.zeros((b, 1, c, h, w), device=x.device)
            for i in range(t):
                _x = _x * r + x[:, i].unsqueeze(1) * (1 - r)
            x, m = _x, m[:, 0].unsqueeze(1)
        task_logits, features = self.predict(x, m, with_video_features=True, train=train)
        video_features = features["video"]
        task_losses = [
            loss_fn(logits, labels) if single_task == None or i == single_task else 0
            for i, loss_fn, logits, labels in zip(range(len(self.losses)), self.losses, task_logits, y)
        ]
        if not train:
            return task_losses, task_logits
        return task_losses, task_logits, self._other_losses(task_losses, video_features, features, comp, speed, b, x.device)

    def _other_losses(self, task_losses, video_features, features, comp, speed, b, device):
        """The reference's auxiliary training losses that can be reached there — the playback-speed terms on the [B, D]
        video features (reference `src/models.py:680-736`).  Small tensors, plain torch arithmetic, differentiable
        through the HIP autograd nodes."""
        if "temporal" not in self.train_mode:
            return {}
        by_speed = torch.argsort(speed, descending=True).tolist()  # clip indices, fastest first
        kind = self.train_mode.temporal
        if kind == "ranking":
            return {"speed/rank": 0.05 * self._speed_ranking(video_features, by_speed, device)}
        if kind == "triplet":
            return {"speed/triplet": 0.01 * self._speed_triplets(video_features, speed, by_speed, device)}
        raise NotImplementedError(f"train_mode.temporal = {kind!r}")

    def _speed_ranking(self, feats, by_speed, device):
        """Every faster clip must out-score every slower one by the margin-ranking default (`:684-704`): mean over the
        B (B - 1) / 2 ordered pairs."""
        score = (feats @ self.ranking_transform_param).squeeze()
        n = len(by_speed)
        terms = [torch.nn.functional.margin_ranking_loss(score[by_speed[i]].repeat(n - 1 - i), score[by_speed[i + 1:], ...],
                                                         torch.ones(n - 1 - i, device=device), reduction="none")
                 for i in range(n - 1)]
        return torch.cat(terms).mean()

    def _speed_triplets(self, feats, speed, by_speed, device):
        """Up to ten random clip triples, ordered by speed (fast, middle, slow); the middle clip must sit closer to each end
        than the other end does, by the speed gap (`:706-733`).  Mean over both directions of every triple."""
        n = len(by_speed)
        clips = list(range(n))
        random.shuffle(clips)
        triples = combinations(clips, 3)
        count = min(comb(n, 3), 10)
        total = torch.tensor(0.0, device=device)
        for _ in range(count):
            fast, mid, slow = sorted(next(triples), key=by_speed.index)
            for anchor, other in ((fast, slow), (slow, fast)):
                total = total + torch.nn.functional.triplet_margin_loss(anchor=feats[anchor], positive=feats[mid], negative=feats[other],
                                                                        margin=torch.abs(speed[other] - speed[mid]))
        return total / (2 * count)

    def configure_optimizers(self, lr):
        """Optimizer over the trainable parameters only — decoder (+ adapter); the encoder is frozen (reference
        `src/models.py:740-754`: SGD with momentum 0.95, or AdamW; both with the config's weight decay).  A
        `torch.optim.Optimizer`, so `OneCycleLR` and the trainer's `param_groups[0]["lr"]` read work unchanged."""
        trainable = [p for p in self.parameters() if p.requires_grad]
        if self.optimizer == "sgd" and trainable and all(p.is_cuda and p.dtype == torch.float32 for p in trainable):
            # one HIP launch per step, which also keeps the decoder's transposed weight copies current (optim.py)
            from .optim import FusedSGD
            return FusedSGD(trainable, lr=lr, momentum=0.95, weight_decay=self.weight_decay, mirrors=self.decoder)
        make = {"sgd": lambda: torch.optim.SGD(trainable, lr=lr, momentum=0.95, weight_decay=self.weight_decay),
                "adamw": lambda: torch.optim.AdamW(trainable, lr=lr, weight_decay=self.weight_decay)}.get(self.optimizer)
        return make() if make is not None else None  # (an unknown name yields None in the reference too)
