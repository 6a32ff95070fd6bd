"""GPU parity of the uint8 ingest kernel (`dfd_preprocess_u8`): the device-side
`Detector._transform` (reference `src/models.py:756-768`) fused with patch extraction.

The reference delegates this arithmetic to torchvision, which is not installed here, so the
check is against the ATen ops torchvision's tensor path calls, run on the CPU in fp32:
`F.interpolate(bicubic, antialias=…)` -> round/clamp to the uint8 grid -> centre crop -> /255
-> Normalize.  Parity with torchvision itself is unpinned (its default for `antialias` changed
across releases; both settings are covered).

Tolerance: the normalised value of a pixel is a function of its uint8 level, so outputs agree
to fp32 rounding (1e-6) unless the resized value falls within float rounding of a .5 tie, where
an fma/non-fma difference moves it one uint8 level (0.0146/std).  At most 0.05 % of pixels
may differ, and none by more than one level."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


@pytest.fixture(scope="module")
def capi():
    from dfd_clip_amd import capi as c
    c.load_library()
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return c


def smooth_u8(n, h, w, seed):
    """Image-like uint8 frames: low-frequency content plus noise, full 0..255 range."""
    rng = np.random.default_rng(seed)
    low = torch.from_numpy(rng.uniform(0, 255, (n, 3, max(2, h // 16), max(2, w // 16))).astype(np.float32))
    img = F.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)
    img = img + torch.from_numpy(rng.normal(0, 12, (n, 3, h, w)).astype(np.float32))
    return img.round().clamp(0, 255).to(torch.uint8)


def reference_transform(frames, res, antialias):
    n, _, h, w = frames.shape
    s, l = (h, w) if h <= w else (w, h)
    new_l = int(res * l / s)
    nh, nw = (res, new_l) if h <= w else (new_l, res)
    x = frames.float()
    if (nh, nw) != (h, w):
        x = F.interpolate(x, size=(nh, nw), mode="bicubic", antialias=antialias, align_corners=False)
        x = x.round().clamp(0, 255)
    top, left = int(round((nh - res) / 2.0)), int(round((nw - res) / 2.0))
    x = x[..., top:top + res, left:left + res] / 255.0
    return (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)


def check(got, want, exact=False):
    got = got.float().cpu()
    err = (got - want).abs()
    level = (1.0 / 255.0) / min(STD)
    if exact:
        assert err.max().item() <= 2e-6, f"max err {err.max().item():.3e}"
        return
    off = (err > 2e-6)
    assert err.max().item() <= level * 1.001 + 2e-6, f"a pixel moved by more than one uint8 level: {err.max().item():.4f}"
    assert off.float().mean().item() <= 5e-4, f"{off.sum().item()} of {off.numel()} pixels off by one level"


@pytest.mark.parametrize("h,w", [(224, 224), (224, 300), (256, 320), (150, 150), (301, 224), (448, 448), (360, 640), (97, 131)])
@pytest.mark.parametrize("antialias", [False, True])
def test_frames_layout(capi, h, w, antialias):
    frames = smooth_u8(3, h, w, seed=h * 1000 + w)
    want = reference_transform(frames, 224, antialias)
    out = torch.empty(3, 3, 224, 224, device="cuda")
    capi.preprocess_u8(frames.cuda(), out, 224, 16, MEAN, STD, antialias=antialias, patch_rows=False)
    check(out, want, exact=(h, w) == (224, 224))


def test_geometry_matches_host(capi):
    from dfd_clip_amd.detector import ClipTransform
    t = ClipTransform(224)
    for h, w in [(224, 224), (225, 224), (224, 225), (227, 300), (1080, 1920), (480, 853), (97, 131), (640, 360)]:
        assert capi.preprocess_geometry(h, w, 224) == t.geometry(h, w), (h, w)


@pytest.mark.parametrize("res,patch,kpad", [(224, 16, 768), (224, 14, 640), (224, 32, 3072), (32, 16, 768)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_rows_equal_patchify_of_frames(capi, res, patch, kpad, dtype):
    """layout 1 writes what `dfd_patchify` writes for the layout-0 frames, pad columns zeroed."""
    frames = smooth_u8(2, res + 37, res + 12, seed=patch).cuda()
    fr = torch.empty(2, 3, res, res, device="cuda")
    capi.preprocess_u8(frames, fr, res, patch, MEAN, STD, antialias=True, patch_rows=False)
    P = (res // patch) ** 2
    want = torch.empty(2 * P, kpad, device="cuda", dtype=dtype)
    capi.patchify(fr, want, res, patch)
    got = torch.full((2 * P, kpad), 7.0, device="cuda", dtype=dtype)
    capi.preprocess_u8(frames, got, res, patch, MEAN, STD, antialias=True, patch_rows=True)
    assert torch.equal(got, want)


def test_clip_transform_device_equals_host(capi):
    from dfd_clip_amd.detector import ClipTransform
    frames = smooth_u8(4, 200, 260, seed=5)
    for aa in (False, True):
        t = ClipTransform(224, antialias=aa)
        check(t(frames.cuda()), t(frames))
        check(t(frames.cuda().view(2, 2, 3, 200, 260)).flatten(0, 1), t(frames))


def test_rejects_bad_arguments(capi):
    frames = torch.zeros(1, 3, 64, 64, dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 3, 224, 224, device="cuda")
    with pytest.raises(capi.DfdError):
        capi.preprocess_u8(frames, out, 224, 15, MEAN, STD, patch_rows=False)      # res % patch
    with pytest.raises(capi.DfdError):
        capi.preprocess_u8(frames, out, 224, 16, MEAN, (0.2, 0.0, 0.2), patch_rows=False)  # zero std
    big = torch.zeros(1, 3, 2240, 2240, dtype=torch.uint8, device="cuda")       # 10x antialiased downscale: LDS window too large
    with pytest.raises(capi.DfdError):
        capi.preprocess_u8(big, out, 224, 32, MEAN, STD, antialias=True, patch_rows=False)


def test_detector_accepts_uint8_clips(capi):
    """`predict` on raw uint8 clips == `predict` on `model.transform(clips)`."""
    from dfd_clip_amd.config import default_detector_config
    from dfd_clip_amd.detector import Detector
    from dfd_clip_amd.weights import random_state_dict
    cfg = default_detector_config()
    cfg.architecture = "small"
    cfg.decode_mode = "index"
    cfg.decode_indices = [1, 2]
    cfg.out_dim = [2]
    cfg.losses = ["auc_roc"]
    det = Detector(cfg, num_frames=3, precision="fp32").cuda().eval()
    det.load_state_dict(random_state_dict(cfg, 3, seed=3))
    clips = smooth_u8(6, 240, 300, seed=9).view(2, 3, 3, 240, 300).cuda()
    m = torch.ones(2, 3, dtype=torch.bool, device="cuda")
    with torch.no_grad():
        a = det.predict(clips, m)[0][0]
        b = det.predict(det.transform(clips), m)[0][0]
    assert torch.allclose(a, b, atol=1e-5), (a, b)
