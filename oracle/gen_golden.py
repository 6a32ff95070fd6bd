"""ORACLE — TEST INFRASTRUCTURE ONLY; runs ONLY in the build container (needs /root/reference).

Generates the golden fixtures under `tests/golden/` by importing the reference's own
`src/clip/model.py` and `src/models.py` and running its `Detector` on seeded inputs and
seeded weights.  Fixtures hold data only (inputs are re-derivable from seeds; outputs are
stored); no reference source travels.  The three import stubs follow SURVEY.md Appendix A:
`yacs.config.CfgNode`, an inert `torchvision.transforms`, and a package shim whose
`clip.load` builds the `VisionTransformer` locally instead of downloading a checkpoint.

Usage:  python oracle/gen_golden.py [case ...]
"""
import contextlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from dfd_clip_amd.config import ConfigNode, default_detector_config  # noqa: E402
from dfd_clip_amd.weights import ARCHS, random_state_dict, synthetic_clips  # noqa: E402


def load_reference():
    class CN(dict):
        def __init__(self, init=None, new_allowed=False):
            super().__init__()
            for k, v in (init or {}).items():
                self[k] = CN(v) if isinstance(v, dict) and not isinstance(v, CN) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    yacs = types.ModuleType("yacs")
    yc = types.ModuleType("yacs.config")
    yc.CfgNode = CN
    yacs.config = yc
    sys.modules.update({"yacs": yacs, "yacs.config": yc})
    tv = types.ModuleType("torchvision")
    T = types.ModuleType("torchvision.transforms")

    class _Inert:
        def __init__(self, *a, **k):
            pass

    for n in ["Compose", "Resize", "CenterCrop", "ConvertImageDtype", "Normalize"]:
        setattr(T, n, _Inert)
    T.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    tv.transforms = T
    sys.modules.update({"torchvision": tv, "torchvision.transforms": T})
    pkg = types.ModuleType("refsrc")
    pkg.__path__ = [REF + "/src"]
    sys.modules["refsrc"] = pkg

    def _load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    cm = _load("refsrc.clip_model", REF + "/src/clip/model.py")
    sys.modules["refsrc.clip_model"] = cm

    def load(name, *a, **k):
        r, p, w, l, h, o = ARCHS[name]
        vit = cm.VisionTransformer(r, p, w, l, h, o)
        for blk in vit.transformer.resblocks:  # torch.empty in the reference
            torch.nn.init.normal_(blk.attn.in_proj_weight, std=w ** -0.5)
            torch.nn.init.normal_(blk.attn.in_proj_bias, std=0.02)
        return types.SimpleNamespace(visual=vit), None

    clipmod = types.ModuleType("refsrc.clip")
    clipmod.load = load
    sys.modules["refsrc.clip"] = clipmod
    pkg.clip = clipmod
    mm = _load("refsrc.models", REF + "/src/models.py")

    class Acc:
        @contextlib.contextmanager
        def main_process_first(self):
            yield

    def to_cn(node):
        return CN({k: (to_cn(v) if isinstance(v, dict) else v) for k, v in node.items()})

    return mm, Acc, to_cn


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


# name -> (arch, B, T, config overrides, what to store)
CASES = {
    "tiny": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1]), "full"),
    "tiny_stride": ("tiny", 2, 4, dict(), "light"),
    "tiny_adapter_nln": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                            adapter__frozen=0, adapter__struct={"type": "768-x-768-nln", "x": 32}), "light"),
    "tiny_adapter_ln": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                           adapter__frozen=0, adapter__struct={"type": "768-x-768-ln", "x": 32}), "light"),
    "tiny_adapter_gl": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                           adapter__frozen=0, adapter__struct={"type": "768-x-768", "x": 32}), "light"),
    "tiny_adapter_legacy": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], adapter__type="normal",
                                               adapter__frozen=0, adapter__struct={"type": "legacy-768-x-768", "x": 32}), "light"),
    "tiny_global": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__global_prediction=1), "light"),
    "tiny_attnmode": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__attn_mode="frame+temporal"), "light"),
    "tiny_nopos": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__temporal_position=0), "light"),
    "tiny_augq": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__aug_query=1), "light"),
    "small": ("small", 2, 3, dict(decode_mode="index", decode_indices=[1, 2]), "medium"),
    "small14": ("small14", 2, 3, dict(decode_mode="index", decode_indices=[0, 1]), "medium"),
    "vitb16_cfg1": ("ViT-B/16", 2, 8, dict(decode_mode="index", decode_indices=[6, 7, 8, 9, 10, 11]), "slices"),
    # BASELINE configs[3]'s architecture (width 1024, 24 layers, 16 heads, 257 tokens, patch K = 588), every other layer tapped
    "vitl14": ("ViT-L/14", 2, 2, dict(decode_mode="stride", decode_stride=2), "slices"),
    # GELU-first adapters at the real width (768 -> 256 -> 768 on ViT-B/16 keys / values, 1 clip x 2 frames, layers 10 and
    # 11 tapped): the LayerNorm behind the GELU normalises 256 values per row here, not the tiny model's 32
    "vitb16_adapter_gl": ("ViT-B/16", 1, 2, dict(decode_mode="index", decode_indices=[10, 11], adapter__type="normal",
                                                 adapter__frozen=0, adapter__struct={"type": "768-x-768", "x": 256}), "light"),
    "vitb16_adapter_legacy": ("ViT-B/16", 1, 2, dict(decode_mode="index", decode_indices=[10, 11], adapter__type="normal",
                                                     adapter__frozen=0, adapter__struct={"type": "legacy-768-x-768", "x": 256}), "light"),
    # training-mode extras (reference models.py:511-544, :572-578, :598-736)
    "tiny_ema": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], op_mode__ema_frame=0.3,
                                    op_mode__temporal_position=0), "light"),
    "tiny_rank": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1], train_mode__temporal="ranking"), "light"),
    "tiny_pmask": ("tiny", 2, 4, dict(decode_mode="index", decode_indices=[0, 1],
                                      train_mode__patch_mask={"type": "batch", "ratio": 0.5}), "light"),
}
EXTRA_INPUTS = dict(comp=["raw", "c23"], speed=[1.0, 2.5], np_seed=11)


def run_case(name, mm, Acc, to_cn):
    arch, B, T, over, store = CASES[name]
    res, patch, width, layers, heads, _ = ARCHS[arch]
    cfg = make_config(arch, **over)
    sd = random_state_dict(cfg, T, seed=0)
    # attn_mode "frame": a fully padded frame is a softmax over all -inf = NaN in the reference
    # (src/models.py:104-112), so that case uses an all-valid mask.
    x, m, y = synthetic_clips(B, T, res, seed=1234, masked_tail=("attn_mode" not in str(over)))
    torch.manual_seed(1)
    det = mm.Detector(to_cn(cfg), T, Acc())
    missing = det.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    det.eval()
    out = {}
    with torch.no_grad():
        losses, logits = det(x, [y], m, single_task=0)
        plog, feats = det.predict(x, m, with_video_features=True)
        enc = det.encoder(x.flatten(0, 1), with_out=True, with_q=True)
    out["logits"] = logits[0].numpy()
    out["losses"] = losses[0].numpy()
    out["video_feature"] = feats["video"].numpy()
    # the reference's own bf16 mixed-precision run (Accelerate `mixed_precision: bf16` = torch.autocast around the
    # forward): the pin for the bf16 HIP path, which cannot be held to the fp32 outputs' 1e-3
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        losses_a, logits_a = det(x, [y], m, single_task=0)
        _, feats_a = det.predict(x, m, with_video_features=True)
        enc_a = det.encoder(x.flatten(0, 1))
    out["logits_bf16"] = logits_a[0].float().numpy()
    out["losses_bf16"] = losses_a[0].float().numpy()
    out["video_feature_bf16"] = feats_a["video"].float().numpy()
    if "ema_frame" not in str(over):  # forward() averages the frames first (models.py:572-578), predict() does not
        assert torch.equal(plog[0], logits[0])
    lidx = det.layer_indices
    out["layer_indices"] = np.asarray(lidx)
    if store == "full":
        for l, d in enumerate(enc):
            for key in ("q", "k", "v", "out"):
                out[f"enc{l}_{key}"] = d[key].numpy()
    elif store == "medium":
        rows = list(range(8)) + list(range(96, 104)) + list(range(-8, 0))
        for l in lidx:
            for key in ("k", "v"):
                for fr in (0, B * T - 1):
                    out[f"enc{l}_{key}_f{fr}"] = enc[l][key][fr, rows].numpy()
                    out[f"enc{l}_{key}_f{fr}_bf16"] = enc_a[l][key][fr, rows].float().numpy()
        out["slice_rows"] = np.asarray(rows)
        out[f"enc{layers - 1}_out_f0"] = enc[layers - 1]["out"][0, rows].numpy()
    elif store == "slices":
        rows = [0, 1, 2, 3, -4, -3, -2, -1]
        for l in (lidx[0], lidx[-1]):
            for key in ("k", "v"):
                for fr in (0, B * T - 1):
                    out[f"enc{l}_{key}_f{fr}"] = enc[l][key][fr, rows].numpy()
                    out[f"enc{l}_{key}_f{fr}_bf16"] = enc_a[l][key][fr, rows].float().numpy()
        out["slice_rows"] = np.asarray(rows)
        out[f"enc{layers - 2}_out_f0"] = enc[layers - 2]["out"][0, rows].numpy()
    # training contract: forward(train=True) -> backward(mean loss) -> two SGD steps on one batch
    det.train()
    opt = det.configure_optimizers(0.01)
    step_losses = []
    speed = torch.tensor(EXTRA_INPUTS["speed"])
    for step in range(2):
        opt.zero_grad()
        np.random.seed(EXTRA_INPUTS["np_seed"] + step)
        tl, tz, other = det(x, [y], m, EXTRA_INPUTS["comp"], speed, train=True, single_task=0)
        if step == 0:
            out["train_task_loss"] = tl[0].detach().numpy().copy()
            for k_, v_ in other.items():
                out["other." + k_] = np.asarray(v_.detach().item())
        loss = tl[0].mean() + sum(other.values())
        loss.backward()
        if step == 0:
            for pn, p in det.named_parameters():
                if p.requires_grad and p.grad is not None:
                    g = p.grad.detach()
                    if store == "full" or g.numel() <= 4096:
                        out["grad0." + pn] = g.numpy().copy()
                    else:
                        out["grad0." + pn + ".norm"] = np.asarray(g.norm().item())
                        out["grad0." + pn + ".head"] = g.flatten()[:64].numpy().copy()
            for pn, p in det.named_parameters():
                assert (p.grad is None) == pn.startswith("encoder."), pn
        step_losses.append(loss.item())
        opt.step()
    out["step_losses"] = np.asarray(step_losses)
    for pn, p in det.named_parameters():
        if p.requires_grad:
            t = p.detach()
            if t.numel() <= 4096:
                out["after2." + pn] = t.numpy().copy()
            else:
                out["after2." + pn + ".head"] = t.flatten()[:64].numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: logits={out['logits'].tolist()} losses={out['losses'].tolist()} "
          f"-> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def run_loader_fixture(arch="tiny"):
    """Checkpoint-loader pin: the reference's own `build_model` (src/clip/model.py:453-496: architecture
    from tensor shapes, `convert_weights` to fp16, strict load) followed by `.visual.float()` (src/models.py:440) on the
    seeded synthetic checkpoint of tests/cases.py — the resulting visual-tower tensors and the inferred architecture."""
    from tests.cases import synthetic_clip_checkpoint
    cm = sys.modules["refsrc.clip_model"]
    out = {}
    for tag, dtype in (("fp32ckpt", None), ("fp16ckpt", torch.float16)):
        sd = synthetic_clip_checkpoint(arch, seed=3, dtype=dtype)
        model = cm.build_model(dict(sd))
        vis = model.visual.float()
        v = vis
        out[f"{tag}.arch"] = np.asarray([v.input_resolution, v.conv1.kernel_size[0], v.conv1.out_channels, len(v.transformer.resblocks),
                                         v.transformer.resblocks[0].attn.n_head, v.output_dim])
        for k, t in vis.state_dict().items():
            a = t.detach().numpy().copy()
            if dtype is torch.float16:  # every value came out of an fp16 tensor: stored in half the bytes, exactly
                assert np.array_equal(a.astype(np.float16).astype(np.float32), a), k
                a = a.astype(np.float16)
            out[f"{tag}.{k}"] = a
    path = os.path.join(ROOT, "tests", "golden", f"clip_loader_{arch}.npz")
    np.savez_compressed(path, **out)
    print(f"loader fixture ({arch}): arch {out['fp32ckpt.arch'].tolist()}, {len(out)} entries -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    torch.set_num_threads(8)
    mm, Acc, to_cn = load_reference()
    for c in (sys.argv[1:] or list(CASES) + ["loader"]):
        if c == "loader":
            run_loader_fixture("tiny")
        else:
            run_case(c, mm, Acc, to_cn)
