"""`Detector`: the drop-in model boundary (reference `src/models.py:394-780`).

Same constructor, attributes, `predict` / `forward` / `configure_optimizers` contracts and
state_dict key schema as the reference class, so reference-style trainer / evaluator /
inference loops (`src/trainer.py:147-151`, `src/evaluator.py:80-83`, `inference.py:115-118`)
run unchanged on top of it.  The frozen encoder and the decoder execute in
libdfdclip_hip.so; PyTorch supplies device memory, streams and the tiny per-sample loss.

    Detector(config, num_frames, accelerator, precision="bf16" | "fp32")

`precision` is the only addition to the reference signature: "fp32" is the parity path
(logits within 1e-3 of the reference's fp32 CPU result), "bf16" the throughput path.
"""
import contextlib
import logging
import os

import torch
from torch import nn

from . import capi
from .config import default_detector_config
from .adapter import CompInvAdapter
from .decoder import Decoder
from .encoder import VisionTransformer
from .weights import ARCHS, resolve_layer_indices

CLIP_CACHE = os.path.expanduser("~/.cache/clip")  # where the reference's downloader leaves checkpoints (clip/clip.py:94)


def auc_roc(weight=None, label_smoothing=0.0, *args, **kargs):
    """Per-sample cross entropy (reference `src/models.py:34-45`)."""
    def driver(logits, y, _weight=weight, _label_smoothing=label_smoothing):
        if _weight:
            _weight = torch.tensor(_weight, device=logits.device)
        return torch.nn.functional.cross_entropy(logits, y, weight=_weight, label_smoothing=_label_smoothing,
                                                 reduction="none")
    return driver


def kl_div(*args, **kargs):
    """Reference `src/models.py:28-31`."""
    def driver(logits, y):
        return torch.nn.functional.kl_div(torch.nn.functional.log_softmax(logits, dim=1), y, reduction="none")
    return driver


def mse(logits, y):
    """Reference `src/models.py:20-25` (140-bin expectation regression)."""
    bins = torch.arange(140, dtype=torch.float32, device=logits.device)
    return torch.pow(logits[:, :140].softmax(dim=-1) @ bins - y, 2) / 1000


_LOSSES = {"auc_roc": auc_roc, "kl_div": kl_div, "mse": lambda *a, **k: mse}


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


def load_clip_visual(name, precision):
    """Counterpart of `clip.load(name)[0].visual.float()` (reference `src/models.py:440`,
    `src/clip/clip.py:94-142`).  The reference downloads the checkpoint by name; there is no
    network here, so: a path to a state_dict checkpoint (or `~/.cache/clip/<name>.pt`) is loaded
    with `weights_only=True`; otherwise the architecture is built with its seeded random
    initialisation and a warning is logged."""
    path = name if os.path.isfile(name) else os.path.join(CLIP_CACHE, name.replace("/", "-") + ".pt")
    if os.path.isfile(path):
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")} or sd
        vit = VisionTransformer(*_infer_arch_from_state_dict(sd), precision=precision)
        vit.load_state_dict({k: v.float() for k, v in sd.items()})
        return vit
    if name not in ARCHS:
        raise RuntimeError(f"Model {name} not found; available architectures = {list(ARCHS)}")
    logging.warning("no checkpoint for %s: encoder uses its random initialisation", name)
    return VisionTransformer(*ARCHS[name], precision=precision)


class ClipTransform:
    """Resize(bicubic, shorter side) -> CenterCrop -> float in [0,1] -> Normalize (reference
    `src/models.py:756-768`), on tensors [T, 3, H, W] uint8 or float."""
    MEAN = (0.48145466, 0.4578275, 0.40821073)
    STD = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, n_px):
        self.n_px = n_px

    def __call__(self, frames):
        x = frames
        if x.dtype == torch.uint8:
            x = x.float() / 255.0
        h, w = x.shape[-2:]
        s = self.n_px / min(h, w)
        nh, nw = max(self.n_px, round(h * s)), max(self.n_px, round(w * s))
        x = torch.nn.functional.interpolate(x, size=(nh, nw), mode="bicubic", antialias=True, align_corners=False)
        top, left = (nh - self.n_px) // 2, (nw - self.n_px) // 2
        x = x[..., top:top + self.n_px, left:left + self.n_px]
        mean = torch.tensor(self.MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(self.STD, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std


class Detector(nn.Module):
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
        self.losses = []
        for loss in config.losses:
            if type(loss) == str:
                self.losses.append(_LOSSES[loss]())
            else:
                self.losses.append(_LOSSES[loss.name](**(dict(loss.args) if "args" in loss else {})))
        self.layer_indices = resolve_layer_indices(config, len(self.encoder.transformer.resblocks))
        self.decoder = Decoder(self, config, num_frames)
        if config.adapter.type == "none":
            self.adapter = None
        elif config.adapter.type == "normal":
            self.adapter = CompInvAdapter(config, self, num_frames=num_frames)
            logging.info("Adapter operates without pretrained weights!!!")
        elif config.adapter.type == "pretrain":
            self.adapter = CompInvAdapter(config, self, num_frames=num_frames)
            data = torch.load(config.adapter.path, map_location="cpu", weights_only=True)
            data = {".".join(k.split(".")[1:]): v for k, v in data.items() if "adapter" in k}
            self.adapter.load_state_dict(data)
            if config.adapter.frozen:
                self.adapter = disable_gradients(self.adapter)
            logging.info(f"Adapter operates with pretrained weights:{config.adapter.path}")
        else:
            raise NotImplementedError()
        self.transform = ClipTransform(self.encoder.input_resolution)
        for key in ("patch_mask", "compression", "nerf_raw", "temporal"):
            if key in self.train_mode:
                raise NotImplementedError(f"train_mode.{key} is not built yet (SURVEY.md §8f rank 4)")
        if "ema_frame" in self.op_mode and self.op_mode.ema_frame:
            raise NotImplementedError("op_mode.ema_frame is not built yet (SURVEY.md §8f rank 4)")

    def predict(self, x, m, with_video_features=False, with_adapt_features=False, train=False):
        """x [B,T,3,R,R], m [B,T] bool -> (task_logits list of [B,out_dim] with L2 norm 5, features)."""
        b, t, c, h, w = x.shape
        if t != self.num_frames and self.decoder.positional_embedding is not None:
            raise RuntimeError(f"The size of tensor a ({t}) must match the size of tensor b ({self.num_frames}) "
                               "at non-singleton dimension 1 (temporal positional embedding)")
        # the encoder is frozen and runs without autograd (reference models.py:440, :501); the decoder is
        # differentiable w.r.t. its own parameters
        pos = self.decoder.temporal_pos()
        if self.adapter is None:
            kv = self.encoder.extract_kv(x.flatten(0, 1), self.layer_indices, t, pos)
        else:
            # raw K/V export, then adapter(kv) + pos (models.py:546-549, :326-329); differentiable w.r.t. the
            # adapter's parameters when they are trainable
            kv = self.encoder.extract_kv(x.flatten(0, 1), self.layer_indices, t, None)
            kv = self.adapter.run(kv[0], kv[1], t, pos)
        _, video_features, task_logits = self.decoder.run(kv, m)
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

    def forward(self, x, y, m, comp=None, speed=None, train=False, single_task=None, *args, **kargs):
        task_logits, features = self.predict(x, m, with_video_features=True, train=train)
        task_losses = [
            loss_fn(logits, labels) if single_task == None or i == single_task else 0
            for i, loss_fn, logits, labels in zip(range(len(self.losses)), self.losses, task_logits, y)
        ]
        if not train:
            return task_losses, task_logits
        return task_losses, task_logits, {}

    def configure_optimizers(self, lr):
        params = [p for p in self.parameters() if p.requires_grad]
        if self.optimizer == "sgd":
            return torch.optim.SGD(params=params, lr=lr, weight_decay=self.weight_decay, momentum=0.95)
        elif self.optimizer == "adamw":
            return torch.optim.AdamW(params=params, lr=lr, weight_decay=self.weight_decay)
