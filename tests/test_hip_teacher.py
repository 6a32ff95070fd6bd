"""GPU test of the trainer's EMA-teacher mode (reference `src/trainer.py:66-69`, `:124-165`,
`:179-190`) through `harness.train_step`: two tasks, a batch labelled for task 0 only; before
teaching only task 0 gets a loss, afterwards task 1 is trained against the teacher's soft
labels.  The soft-label loss and the EMA update are checked against plain torch math."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.cases import make_config

pytestmark = pytest.mark.gpu


def test_teacher_mode_train_steps():
    from dfd_clip_amd.detector import Detector
    from dfd_clip_amd.harness import EmaTeacher, train_step
    from dfd_clip_amd.weights import random_state_dict, synthetic_clips
    cfg = make_config("tiny", decode_mode="index", decode_indices=[0, 1])
    cfg.out_dim = [2, 3]
    cfg.losses = ["auc_roc", "auc_roc"]
    T = 4
    model = Detector(cfg, T, None, precision="fp32")
    model.load_state_dict(random_state_dict(cfg, T, seed=0))
    model = model.cuda()
    x, m, y = synthetic_clips(2, T, 32, seed=5)
    x, m, y = x.cuda(), m.cuda(), y.cuda()
    opt = model.configure_optimizers(0.05)
    teacher = EmaTeacher(model, ema_ratio=0.5, teach_at=0)
    proj1 = "decoder.proj1x3"
    p_before = dict(model.named_parameters())[proj1].detach().clone()
    t_before = dict(teacher.module.named_parameters())[proj1].detach().clone()

    # step 1: not teaching yet -> task 1's head receives no gradient, only weight decay moves it
    out = train_step(model, opt, [(x, y, m, None, None, 0)], teacher=teacher)
    assert out["logits"][0].shape == (2, 2) and teacher.teaching
    p_after = dict(model.named_parameters())[proj1].detach()
    np.testing.assert_allclose(p_after.cpu().numpy(), (p_before * (1 - 0.05 * cfg.weight_decay)).cpu().numpy(), atol=1e-7)
    t_after = dict(teacher.module.named_parameters())[proj1].detach()
    np.testing.assert_allclose(t_after.cpu().numpy(), (0.5 * t_before + 0.5 * p_after).cpu().numpy(), atol=1e-7)

    # step 2: teaching -> task 1 is trained on softmax(teacher logits); verify its loss value and that its head moves
    with torch.no_grad():
        _, t_logits = teacher.module(x, [None, None], m, single_task=-1)
        soft = t_logits[1].softmax(dim=-1)
        model.eval()
        _, s_logits = model(x, [None, None], m, single_task=-1)
        want_loss1 = F.cross_entropy(s_logits[1], soft, reduction="none")
    labels = teacher.labels(x, m, y, 0, 2)
    assert labels[0] is y and torch.allclose(labels[1], soft)
    model.train()
    losses, _, _ = model(x, labels, m, train=True, single_task=None)
    torch.testing.assert_close(losses[1].detach(), want_loss1, atol=1e-5, rtol=0)
    p_mid = dict(model.named_parameters())[proj1].detach().clone()
    train_step(model, opt, [(x, y, m, None, None, 0)], teacher=teacher)
    moved = (dict(model.named_parameters())[proj1].detach() - p_mid * (1 - 0.05 * cfg.weight_decay)).abs().max().item()
    assert moved > 1e-6, "task 1's projection must receive a gradient from the teacher's soft labels"
