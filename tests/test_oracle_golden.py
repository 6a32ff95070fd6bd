"""Pins the CPU oracle (`oracle/ref_cpu.py`) against outputs of the reference itself.

`tests/golden/*.npz` were written by `oracle/gen_golden.py`, which imported the reference's
`Detector` in the build container and ran it on these seeded inputs and weights.  The
reference has no tests or vectors of its own for this path, so these are the
pins.  Tolerance: fp32 vs fp32 on the same torch CPU ops, 2e-5 absolute on O(1) values.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from tests.cases import CASES, build_case, load_golden, oracle_kwargs

TOL = 2e-5
FAST = [c for c in CASES if c not in ("vitb16_cfg1", "vitl14")]


@pytest.mark.parametrize("name", FAST)
def test_oracle_logits_match_reference(name):
    case = build_case(name)
    g = load_golden(name)
    op = case["cfg"].op_mode
    ema = op.ema_frame if "ema_frame" in op else 0
    losses, logits = ref_cpu.detector_forward_eval(case["sd"], case["x"], [case["y"]], case["m"], single_task=0, ema_frame=ema,
                                                   **oracle_kwargs(case))
    np.testing.assert_allclose(logits[0].numpy(), g["logits"], atol=TOL, rtol=0)
    np.testing.assert_allclose(losses[0].numpy(), g["losses"], atol=TOL, rtol=1e-5)
    if not ema:  # the golden's video_feature comes from predict(), which does not average the frames
        _, feat = ref_cpu.detector_predict(case["sd"], case["x"], case["m"], **oracle_kwargs(case))
        np.testing.assert_allclose(feat.numpy(), g["video_feature"], atol=TOL, rtol=0)
    assert list(g["layer_indices"]) == case["layer_indices"]


def test_oracle_encoder_per_layer_tensors():
    """Every layout rule of the encoder: q|k|v row order, head = contiguous 64-channel slice,
    CLS row kept, bias included, `out` = residual stream after the block."""
    case = build_case("tiny")
    g = load_golden("tiny")
    kvs = ref_cpu.encoder_forward(case["sd"], case["x"].flatten(0, 1), case["heads"], case["patch"],
                                  with_out=True, with_q=True)
    assert len(kvs) == case["layers"]
    for l, d in enumerate(kvs):
        for key in ("q", "k", "v", "out"):
            np.testing.assert_allclose(d[key].numpy(), g[f"enc{l}_{key}"], atol=TOL, rtol=0, err_msg=f"{l}/{key}")


def test_oracle_encoder_197_tokens_slices():
    case = build_case("small")
    g = load_golden("small")
    rows = list(g["slice_rows"])
    kvs = ref_cpu.encoder_forward(case["sd"], case["x"].flatten(0, 1), case["heads"], case["patch"], with_out=True)
    n = case["B"] * case["T"]
    for l in case["layer_indices"]:
        for key in ("k", "v"):
            for fr in (0, n - 1):
                np.testing.assert_allclose(kvs[l][key][fr, rows].numpy(), g[f"enc{l}_{key}_f{fr}"], atol=TOL, rtol=0)
    L = case["layers"] - 1
    np.testing.assert_allclose(kvs[L]["out"][0, rows].numpy(), g[f"enc{L}_out_f0"], atol=5e-5, rtol=0)


def test_oracle_gradients_and_sgd_steps():
    """Training contract (reference `src/trainer.py:147-177`): forward(train) -> mean loss ->
    backward -> SGD(momentum 0.95, wd 0.01) twice on one batch.  The oracle's forward is
    differentiable torch, so autograd of it is compared with autograd of the reference."""
    case = build_case("tiny")
    g = load_golden("tiny")
    w = {k: v.clone() for k, v in case["sd"].items()}
    params = {k: v.requires_grad_(True) for k, v in w.items() if not k.startswith("encoder.")}
    opt = torch.optim.SGD(list(params.values()), lr=0.01, weight_decay=0.01, momentum=0.95)
    step_losses = []
    for step in range(2):
        opt.zero_grad()
        losses, _ = ref_cpu.detector_forward_eval(w, case["x"], [case["y"]], case["m"], single_task=0,
                                                  **oracle_kwargs(case))
        loss = losses[0].mean()
        loss.backward()
        if step == 0:
            for k, p in params.items():
                np.testing.assert_allclose(p.grad.numpy(), g["grad0." + k], atol=2e-6, rtol=1e-4, err_msg=k)
        step_losses.append(loss.item())
        opt.step()
    np.testing.assert_allclose(step_losses, g["step_losses"], atol=1e-5)
    for k, p in params.items():
        key = "after2." + k
        if key in g.files:
            np.testing.assert_allclose(p.detach().numpy(), g[key], atol=1e-5, rtol=0, err_msg=k)
        else:
            np.testing.assert_allclose(p.detach().flatten()[:64].numpy(), g[key + ".head"], atol=1e-5, rtol=0)


def test_oracle_vitb16_cfg1():
    """BASELINE.json configs[0]: ViT-B/16, 2 clips x 8 frames, decode_indices 6..11."""
    case = build_case("vitb16_cfg1")
    g = load_golden("vitb16_cfg1")
    kw = oracle_kwargs(case)
    with torch.no_grad():
        logits, feat, _ = ref_cpu.detector_predict(case["sd"], case["x"], case["m"], return_kvs=True, **kw)
        enc = ref_cpu.encoder_forward(case["sd"], case["x"].flatten(0, 1)[[0, 15]], case["heads"], case["patch"],
                                      with_out=True)
    np.testing.assert_allclose(logits[0].numpy(), g["logits"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(feat.numpy(), g["video_feature"], atol=1e-4, rtol=0)
    rows = list(g["slice_rows"])
    for l in (6, 11):
        for key in ("k", "v"):
            for i, fr in enumerate((0, 15)):
                np.testing.assert_allclose(enc[l][key][i, rows].numpy(), g[f"enc{l}_{key}_f{fr}"], atol=1e-4, rtol=0)


def test_oracle_vitl14():
    """BASELINE.json configs[3]'s architecture: ViT-L/14 (width 1024, 24 layers, 16 heads, 257 tokens), 2 clips x 2
    frames, every other layer tapped (reference `src/clip/model.py:453-470`, `src/models.py:459`)."""
    case = build_case("vitl14")
    g = load_golden("vitl14")
    kw = oracle_kwargs(case)
    assert case["layer_indices"] == list(range(0, 24, 2)) == list(g["layer_indices"])
    with torch.no_grad():
        logits, feat = ref_cpu.detector_predict(case["sd"], case["x"], case["m"], **kw)
        enc = ref_cpu.encoder_forward(case["sd"], case["x"].flatten(0, 1)[[0, 3]], case["heads"], case["patch"])
    np.testing.assert_allclose(logits[0].numpy(), g["logits"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(feat.numpy(), g["video_feature"], atol=1e-4, rtol=0)
    rows = list(g["slice_rows"])
    for l in (0, 22):
        for key in ("k", "v"):
            for i, fr in enumerate((0, 3)):
                np.testing.assert_allclose(enc[l][key][i, rows].numpy(), g[f"enc{l}_{key}_f{fr}"], atol=1e-4, rtol=0)


def test_every_fixture_is_current():
    """Every fixture was written by the current generator: it carries the training-contract keys and the
    reference's bf16-autocast outputs."""
    for name in CASES:
        g = load_golden(name)
        for key in ("logits", "losses", "video_feature", "train_task_loss", "step_losses", "logits_bf16", "video_feature_bf16"):
            assert key in g.files, (name, key)
        # the reference's own bf16 run stays within a few 1e-2 of its fp32 run on these norm-5 logits
        assert np.abs(g["logits_bf16"] - g["logits"]).max() < 0.15, name
