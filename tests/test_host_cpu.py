"""CPU checks of host-side logic: config node, weight schema / seed recipe, Detector
construction and state_dict schema, layer-index resolution."""
import os

import numpy as np
import pytest
import torch

from dfd_clip_amd.build import build
from dfd_clip_amd.config import ConfigNode, default_detector_config
from dfd_clip_amd.weights import ARCHS, decoder_schema, encoder_schema, random_state_dict, resolve_layer_indices, synthetic_clips
from tests.cases import build_case, make_config


def test_config_node_behaviours():
    c = default_detector_config()
    assert c.decode_mode == "stride" and c.decode_stride == 2 and c.adapter.type == "none"
    assert "temporal_position" in c.op_mode and "attn_mode" not in c.op_mode
    c.op_mode.attn_mode = "frame"
    assert c["op_mode"]["attn_mode"] == "frame"
    c2 = c.clone()
    c2.op_mode.attn_mode = "x"
    assert c.op_mode.attn_mode == "frame"
    with pytest.raises(AttributeError):
        c.nope
    assert isinstance(ConfigNode({"a": {"b": 1}}).a, ConfigNode)


def test_layer_indices():
    c = default_detector_config()
    assert resolve_layer_indices(c, 12) == [0, 2, 4, 6, 8, 10]
    c.decode_mode, c.decode_indices = "index", [6, 7, 8, 9, 10, 11]
    assert resolve_layer_indices(c, 12) == [6, 7, 8, 9, 10, 11]


def test_schema_sizes_match_survey():
    enc = encoder_schema("ViT-B/16")
    assert sum(int(np.prod(s)) for s in enc.values()) == 86_192_640
    dec = decoder_schema("ViT-B/16", 8, 6, [2])
    assert sum(int(np.prod(s)) for s in dec.values()) == 38_995_200


def test_seed_recipe_is_deterministic():
    cfg = make_config("tiny", decode_mode="index", decode_indices=[0, 1])
    a, b = random_state_dict(cfg, 4, seed=0), random_state_dict(cfg, 4, seed=0)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    c = random_state_dict(cfg, 4, seed=1)
    assert not torch.equal(a["encoder.proj"], c["encoder.proj"])
    x, m, y = synthetic_clips(2, 8, 32)
    assert x.shape == (2, 8, 3, 32, 32) and m[0].all() and m[1].tolist() == [True] * 6 + [False] * 2 and y.tolist() == [0, 1]


def test_detector_state_dict_schema_and_roundtrip():
    build()
    from dfd_clip_amd.detector import Detector
    case = build_case("tiny_augq")
    det = Detector(case["cfg"], case["T"], None, precision="fp32")
    assert list(det.state_dict().keys()) != []
    assert set(det.state_dict().keys()) == set(case["sd"].keys())
    for k, v in det.state_dict().items():
        assert tuple(v.shape) == tuple(case["sd"][k].shape), k
    det.load_state_dict(case["sd"])
    assert all(not p.requires_grad for p in det.encoder.parameters())
    trainable = [n for n, p in det.named_parameters() if p.requires_grad]
    assert trainable and all(n.startswith("decoder.") for n in trainable)
    opt = det.configure_optimizers(0.01)
    assert opt.defaults["momentum"] == 0.95 and opt.defaults["weight_decay"] == 0.01
    assert det.out_dim == [2] and det.layer_indices == [0, 1] and callable(det.transform)
    with pytest.raises(Exception):
        det.predict(case["x"], case["m"])  # CPU tensors: no fallback path


def test_full_size_schema_matches_reference_key_count():
    from dfd_clip_amd.detector import Detector
    cfg = make_config("ViT-B/16", decode_mode="index", decode_indices=[6, 7, 8, 9, 10, 11])
    det = Detector(cfg, 8, None)
    keys = det.state_dict().keys()
    assert len(keys) == 152 + 79  # encoder 152 + decoder 79
    assert sum(p.numel() for p in det.parameters() if p.requires_grad) == 38_995_200


def test_unbuilt_variants_raise():
    from dfd_clip_amd.detector import Detector
    cfg = make_config("tiny", adapter__type="normal", adapter__struct={"type": "768-bn", "x": 32})
    with pytest.raises(NotImplementedError):
        Detector(cfg, 4, None)
    cfg = make_config("tiny", foundation="dinov2")
    with pytest.raises(NotImplementedError):
        Detector(cfg, 4, None)


def test_adapter_state_dict_schema():
    from dfd_clip_amd.detector import Detector
    case = build_case("tiny_adapter_nln")
    det = Detector(case["cfg"], case["T"], None, precision="fp32")
    assert set(det.state_dict().keys()) == set(case["sd"].keys())
    det.load_state_dict(case["sd"])
    assert det.adapter.l0_k[1].weight.shape == (4, 32)


def test_clip_transform_geometry_and_host_path():
    """`ClipTransform` (reference `src/models.py:756-768`): torchvision's geometry rules (longer
    side truncated, crop origin rounded half to even) and the uint8 re-quantisation after a
    resize; the C helper `dfd_preprocess_geometry` must agree (host-only call, no GPU)."""
    import ctypes

    import torch
    from dfd_clip_amd import capi
    from dfd_clip_amd.detector import ClipTransform
    t = ClipTransform(224)
    assert t.geometry(224, 224) == (224, 224, 0, 0)
    assert t.geometry(224, 225) == (224, 225, 0, 0)       # round(0.5) = 0
    assert t.geometry(224, 227) == (224, 227, 0, 2)       # round(1.5) = 2
    assert t.geometry(480, 853) == (224, 398, 0, 87)      # int(224*853/480) = 398
    assert t.geometry(1920, 1080) == (398, 224, 87, 0)
    lib = capi.load_library()
    for h, w in [(224, 224), (224, 225), (224, 227), (480, 853), (1920, 1080), (97, 131), (640, 360)]:
        v = [ctypes.c_int() for _ in range(4)]
        assert lib.dfd_preprocess_geometry(h, w, 224, *[ctypes.byref(i) for i in v]) == 0
        assert tuple(i.value for i in v) == t.geometry(h, w)
    # no resize: exactly (u8/255 - mean)/std
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (2, 3, 224, 224), generator=g, dtype=torch.uint8)
    want = (u8.float() / 255.0 - torch.tensor(t.MEAN).view(1, 3, 1, 1)) / torch.tensor(t.STD).view(1, 3, 1, 1)
    assert torch.equal(t(u8), want)
    # resize: output pixels sit on the uint8 grid
    out = ClipTransform(32)(torch.randint(0, 256, (1, 3, 50, 70), generator=g, dtype=torch.uint8))
    assert out.shape == (1, 3, 32, 32)
    lv = (out * torch.tensor(t.STD).view(1, 3, 1, 1) + torch.tensor(t.MEAN).view(1, 3, 1, 1)) * 255.0
    assert (lv - lv.round()).abs().max() < 1e-3


def test_ema_teacher_update_and_schedule():
    """`EmaTeacher` (reference `src/trainer.py:66-69`, `:179-190`): deep copy, p_t <- (1-r) p_t + r p after
    every step, teaching switches on once teach_at < steps.  Host logic only (no forward)."""
    import torch
    from dfd_clip_amd.detector import Detector
    from dfd_clip_amd.harness import EmaTeacher
    cfg = make_config("tiny", decode_mode="index", decode_indices=[0, 1])
    model = Detector(cfg, 4, None, precision="fp32")
    teacher = EmaTeacher(model, ema_ratio=0.25, teach_at=2)
    assert teacher.module is not model
    before = {n: p.detach().clone() for n, p in teacher.module.named_parameters()}
    with torch.no_grad():
        for p in model.parameters():
            if p.requires_grad:
                p.add_(1.0)
    states = []
    for _ in range(3):
        teacher.update(model)
        states.append(teacher.teaching)
    assert states == [False, False, True]       # teach_at=2 < steps=3
    for (n, pt), (_, pm) in zip(teacher.module.named_parameters(), model.named_parameters()):
        if pm.requires_grad:  # three EMA steps towards p+1: 1 - 0.75^3
            torch.testing.assert_close(pt, before[n] + (1 - 0.75 ** 3), atol=1e-6, rtol=0)
        else:
            torch.testing.assert_close(pt, before[n], atol=1e-6, rtol=0)
        assert not pt.requires_grad or pm.requires_grad


def _clip_checkpoint(arch, dtype, seed=3):
    """A synthetic CLIP checkpoint in the published key layout: `visual.`-prefixed ViT tower plus a few of the
    text-tower keys `build_model` reads (reference clip/model.py:453-480)."""
    from dfd_clip_amd.weights import _fill
    rng = np.random.default_rng(seed)
    sd = {"visual." + k: _fill(rng, k, shp).to(dtype) for k, shp in encoder_schema(arch).items()}
    sd["text_projection"] = torch.zeros(8, 8, dtype=dtype)
    sd["positional_embedding"] = torch.zeros(4, 8, dtype=dtype)
    sd["logit_scale"] = torch.tensor(1.0)
    return sd


@pytest.mark.parametrize("arch", ["tiny", "small14"])
def test_load_clip_visual_from_fp16_checkpoint(tmp_path, arch):
    """`clip.load(path)` counterpart (reference clip/clip.py:121-122, :131-139; model.py:453-470): architecture
    inferred from tensor shapes, `visual.` tower extracted, fp16 values widened to fp32 unchanged."""
    from dfd_clip_amd.detector import load_clip_visual
    build()
    sd = _clip_checkpoint(arch, torch.float16)
    path = str(tmp_path / "ckpt.pt")
    torch.save(sd, path)
    vit = load_clip_visual(path, "fp32")
    res, patch, width, layers, heads, out_dim = ARCHS[arch]
    assert (vit.input_resolution, vit.patch_size, vit.width, vit.layers, vit.heads, vit.output_dim) == (res, patch, width, layers, heads, out_dim)
    got = vit.state_dict()
    assert set(got) == {k[len("visual."):] for k in sd if k.startswith("visual.")}
    for k, v in got.items():
        assert v.dtype == torch.float32 and torch.equal(v, sd["visual." + k].float()), k


def test_load_clip_visual_fp32_checkpoint_takes_the_fp16_round_trip(tmp_path):
    """`convert_weights` (reference clip/model.py:429-450, :494) halves Conv/Linear parameters and `proj` before
    the checkpoint is copied in; the reference's own attention class keeps `in_proj_*` in fp32, like LayerNorms
    and embeddings.  An fp32 checkpoint therefore comes out rounded exactly there."""
    from dfd_clip_amd.detector import load_clip_visual
    build()
    sd = _clip_checkpoint("tiny", torch.float32)
    path = str(tmp_path / "ckpt32.pt")
    torch.save({k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}, path)  # bare tower also loads
    got = load_clip_visual(path, "bf16").state_dict()
    rounded = 0
    for k, v in got.items():
        src = sd["visual." + k]
        halved = k in ("conv1.weight", "proj") or any(t in k for t in (".out_proj.", ".mlp.c_fc.", ".mlp.c_proj."))
        want = src.half().float() if halved else src
        assert torch.equal(v, want), k
        rounded += int(halved and not torch.equal(want, src))
    assert rounded >= 4 * 2 + 2  # the round trip really changed those tensors
    assert torch.equal(got["transformer.resblocks.0.attn.in_proj_weight"], sd["visual.transformer.resblocks.0.attn.in_proj_weight"])


@pytest.mark.parametrize("tag,dtype", [("fp32ckpt", None), ("fp16ckpt", torch.float16)])
def test_load_clip_visual_equals_the_references_build_model(tmp_path, golden_dir, tag, dtype):
    """Pinned by the reference itself: tests/golden/clip_loader_tiny.npz holds what the reference's `build_model(sd)`
    (src/clip/model.py:453-496, incl. `convert_weights`) followed by `.visual.float()` (src/models.py:440) made of the
    seeded synthetic checkpoint of tests/cases.py (oracle/gen_golden.py loader).  `load_clip_visual` on the same
    checkpoint: same architecture, every tensor bit for bit."""
    from dfd_clip_amd.detector import load_clip_visual
    from tests.cases import synthetic_clip_checkpoint
    build()
    g = np.load(os.path.join(golden_dir, "clip_loader_tiny.npz"), allow_pickle=False)
    sd = synthetic_clip_checkpoint("tiny", seed=3, dtype=dtype)
    path = str(tmp_path / "clip.pt")
    torch.save(sd, path)
    vit = load_clip_visual(path, "fp32")
    assert [vit.input_resolution, vit.patch_size, vit.width, vit.layers, vit.heads, vit.output_dim] == g[tag + ".arch"].tolist()
    got = vit.state_dict()
    want = {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + ".") and not k.endswith(".arch")}
    assert set(got) == set(want)
    for k, v in got.items():
        assert v.dtype == torch.float32 and np.array_equal(v.numpy(), want[k].astype(np.float32)), k


def test_load_clip_visual_rejects_torchscript_archive_with_instructions(tmp_path):
    """The published ViT-B-16.pt is a TorchScript archive (reference clip/clip.py:127-130 tries torch.jit.load
    first).  This loader executes nothing from a file, so it must say what the file is and how to convert it."""
    from dfd_clip_amd.detector import load_clip_visual
    build()

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(2))

        def forward(self, x):
            return x + self.w

    path = str(tmp_path / "ViT-X.pt")
    torch.jit.script(M()).save(path)
    with pytest.raises(RuntimeError, match="TorchScript archive.*state_dict"):
        load_clip_visual(path, "bf16")
    bad = str(tmp_path / "resnet.pt")
    torch.save({"visual.layer1.0.conv1.weight": torch.zeros(1)}, bad)
    with pytest.raises(RuntimeError, match="no CLIP ViT visual tower"):
        load_clip_visual(bad, "bf16")
    with pytest.raises(RuntimeError, match="not found"):
        load_clip_visual("ViT-Z/99", "bf16")


def test_isa_lint_no_register_scalar_offset_on_wide_stores():
    """tools/isa_lint.py over every kernel source: no vector store of more than 8 bytes per lane carries a register in its
    scalar-offset field (the form the compiler's store-data hazard handling skips; on gfx950 it corrupted output tiles
    sporadically in round 3).  Compiles the device code to assembly with hipcc: no GPU involved."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    assert lint.main([]) == 0


def test_detector_zero_grad_matches_module_semantics():
    """`Detector.zero_grad` walks a cached parameter list instead of the module tree: same effect as `nn.Module.zero_grad`
    for both `set_to_none` forms, a gradient on a frozen parameter is cleared too, a parameter unfrozen later is seen, and
    a deep copy (the trainer's teacher) has a list of its own."""
    import copy
    from dfd_clip_amd.detector import Detector
    case = build_case("tiny")
    det = Detector(case["cfg"], case["T"], None, precision="fp32")
    params = list(det.parameters())
    trainable = [p for p in params if p.requires_grad]
    frozen = [p for p in params if not p.requires_grad]
    assert trainable and frozen
    for p in params:
        p.grad = torch.ones_like(p)
    det.zero_grad()
    assert all(p.grad is None for p in params)
    for p in trainable:
        p.grad = torch.ones_like(p)
    det.zero_grad(set_to_none=False)
    assert all(p.grad is not None and not p.grad.any() for p in trainable)
    frozen[0].requires_grad_(True)  # unfrozen after the list was built
    frozen[0].grad = torch.ones_like(frozen[0])
    det.zero_grad()
    assert frozen[0].grad is None
    twin = copy.deepcopy(det)
    for p in twin.parameters():
        p.grad = torch.ones_like(p)
    for p in det.parameters():
        p.grad = torch.ones_like(p)
    twin.zero_grad()
    assert all(p.grad is None for p in twin.parameters()) and all(p.grad is not None for p in det.parameters())
