"""Reference-shaped callers of the model boundary: train step, evaluation pass, per-video inference.

These restate the CONTRACTS of the reference's host loops in this repo's own
code, so that a reference user finds the same step semantics on top of `Detector`:

  train_step      `Trainer.run`'s loop body (`src/trainer.py:108-177`): zero_grad -> for each training
                  set: forward(train=True, single_task=idx) -> backward(task_loss[idx].mean() + Σ other)
                  -> [gradient all-reduce] -> optimizer.step -> lr_scheduler.step.
  make_one_cycle  the OneCycleLR the trainer builds (`src/trainer.py:51-60`): initial lr = max/25 and
                  `total_steps = max_steps * num_processes` because Accelerate steps a prepared
                  scheduler once per process per optimizer step; `step_scheduler` mirrors that.
  EmaTeacher      the trainer's "teacher" mode (`src/trainer.py:66-69`, `:124-137`, `:179-190`): a deep copy of
                  the model updated as an exponential moving average after every optimizer step; once
                  `teach_at` steps have passed it supplies soft labels for the tasks a batch has no labels for.
  evaluate        `Evaluator.run` (`src/evaluator.py:50-97`) + the metric callbacks
                  (`src/callbacks/metrics.py:86-155`): no_grad, eval(), model(x, y_list, m, single_task)
                  -> softmax -> gather over ranks -> accuracy / AUROC on p[:, 1].
  infer_videos    `inference.py:105-162`: per video, chunks of `batch_size` clips -> predict -> softmax ->
                  clip- or video-level (mean over clips) probabilities -> gather -> accuracy / AUROC with
                  the dummy [0, 1] pair the reference appends before computing.
"""
import copy

import torch
from torch.optim.lr_scheduler import OneCycleLR

from . import dist as ddist


def make_one_cycle(optimizer, learning_rate, max_steps, num_processes=None):
    n = num_processes or ddist.world_size()
    return OneCycleLR(optimizer=optimizer, max_lr=learning_rate, total_steps=max_steps * n)


def step_scheduler(scheduler, num_processes=None):
    for _ in range(num_processes or ddist.world_size()):
        scheduler.step()


class EmaTeacher:
    """EMA copy of the model that labels the tasks a batch does not cover (reference
    `src/trainer.py:66-69` deep copy, `:179-185` update p_t <- (1-r) p_t + r p, `:188-190` teaching
    starts once `teach_at < steps`, `:124-137` pseudo labels = softmax of the teacher's logits)."""

    def __init__(self, model, ema_ratio, teach_at):
        self.module = copy.deepcopy(model)
        self.ema_ratio = float(ema_ratio)
        self.teach_at = int(teach_at)
        self.steps = 0
        self.teaching = False

    @torch.no_grad()
    def update(self, model):
        r = self.ema_ratio
        for p1, p2 in zip(self.module.parameters(), model.parameters()):
            p1.data = (1 - r) * p1.data + r * p2.data
        self.steps += 1
        if not self.teaching and self.teach_at < self.steps:
            self.teaching = True

    @torch.no_grad()
    def labels(self, frames, mask, labels, task_index, total_tasks):
        """Per-task label list: the batch's own labels for its task, the teacher's class probabilities
        for every other task (the model's losses accept probability targets)."""
        _, teacher_logits = self.module(frames, [None] * total_tasks, mask, single_task=-1)
        return [labels if i == task_index else teacher_logits[i].softmax(dim=-1) for i in range(total_tasks)]


def train_step(model, optimizer, batches, scheduler=None, total_tasks=None, teacher=None):
    """One optimizer step over `batches`: list of (frames, labels, mask, comps, speed, task_index)
    — one entry per training set, as the reference draws one batch per set per step.
    With an `EmaTeacher` that has started teaching, every task contributes a loss (the other tasks
    against the teacher's soft labels); the teacher is updated after the optimizer step.
    Returns {"losses": [...per batch mean task loss...], "logits": [...]}."""
    total_tasks = total_tasks or len(model.out_dim)
    model.zero_grad()
    model.train()
    out = {"losses": [], "logits": []}
    teaching = teacher is not None and teacher.teaching
    for frames, labels, mask, comps, speed, task_index in batches:
        if teaching:
            y_list = teacher.labels(frames, mask, labels, task_index, total_tasks)
        else:
            y_list = [labels if i == task_index else None for i in range(total_tasks)]
        task_losses, task_logits, other = model(frames, y_list, mask, comps, speed, train=True,
                                                single_task=None if teaching else task_index)
        if teaching:
            loss = sum(l.mean() for l in task_losses) + sum(other[k].mean() for k in other)
        else:
            loss = task_losses[task_index].mean() + sum(other[k].mean() for k in other)
        loss.backward()
        out["losses"].append(task_losses[task_index].detach())
        out["logits"].append(task_logits[task_index].detach())
    ddist.allreduce_gradients([p for p in model.parameters() if p.requires_grad])
    optimizer.step()
    if scheduler is not None:
        step_scheduler(scheduler)
    model.zero_grad()
    if teacher is not None:
        teacher.update(model)
    return out


def binary_auroc(labels, scores):
    """Area under the ROC curve of `scores` for the positive class 1 (ties get the average rank);
    what sklearn.metrics.roc_auc_score — the backend of the reference's `roc_auc` metric — returns."""
    labels = torch.as_tensor(labels).flatten().to(torch.float64)
    scores = torch.as_tensor(scores).flatten().to(torch.float64)
    n_pos = labels.sum().item()
    n_neg = labels.numel() - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("AUROC needs both classes")
    order = torch.argsort(scores)
    s = scores[order]
    ranks = torch.empty_like(s)
    i, n = 0, s.numel()
    while i < n:  # average ranks over ties
        j = i
        while j + 1 < n and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = (i + j) / 2.0 + 1.0
        i = j + 1
    r = torch.empty_like(ranks)
    r[order] = ranks
    return ((r[labels == 1].sum().item() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


@torch.no_grad()
def evaluate(model, batches, total_tasks=None):
    """batches: iterable of (frames, labels, mask, task_index[, valid]) for this rank; `valid` = number
    of real samples when the last batch was padded to a common size across ranks.
    Returns dict(accuracy, roc_auc, loss, labels, probs) over ALL ranks' samples."""
    total_tasks = total_tasks or len(model.out_dim)
    model.eval()
    all_labels, all_probs, all_losses = [], [], []
    for batch in batches:
        frames, labels, mask, task_index = batch[:4]
        valid = batch[4] if len(batch) > 4 else None
        y_list = [labels if i == task_index else None for i in range(total_tasks)]
        task_losses, task_logits = model(frames, y_list, mask, single_task=task_index)
        probs = task_logits[task_index].detach().softmax(dim=-1)
        lab, pr, ls = ddist.gather_for_metrics((labels, probs, task_losses[task_index].detach()), valid=valid)
        all_labels.append(lab.cpu())
        all_probs.append(pr.cpu())
        all_losses.append(ls.cpu())
    labels, probs, losses = torch.cat(all_labels), torch.cat(all_probs), torch.cat(all_losses)
    res = dict(accuracy=(probs.argmax(dim=-1) == labels).float().mean().item(), loss=losses.mean().item(), labels=labels,
               probs=probs)
    try:
        res["roc_auc"] = binary_auroc(labels, probs[:, 1])
    except ValueError:
        res["roc_auc"] = float("nan")
    return res


@torch.no_grad()
def infer_videos(model, videos, batch_size=30, modality="video", task_index=0, device=None):
    """videos: iterable of (clips list of [T,3,R,R], label(s), masks list of [T]) for this rank; a video
    without clips is skipped, as the reference does.  Returns dict(accuracy, roc_auc, labels, probs)."""
    model.eval()
    device = device or next(model.parameters()).device
    labels_all, probs_all = [], []
    for clips, label, masks in videos:
        if len(clips) == 0:
            continue
        logits = []
        for i in range(0, len(clips), batch_size):  # last chunk ragged
            x = torch.stack(clips[i:i + batch_size]).to(device)
            m = torch.stack(masks[i:i + batch_size]).to(device)
            logits.append(model.predict(x, m)[0][task_index].detach().to("cpu"))
        p = torch.cat(logits).softmax(dim=-1)
        if modality == "clip":
            pred_prob, lab = p, torch.as_tensor(label).reshape(-1)
        elif modality == "video":
            pred_prob = p.mean(dim=0).unsqueeze(0)
            lab = torch.as_tensor(label).reshape(-1)[:1]
        else:
            raise NotImplementedError()
        n = ddist.world_size()
        if n > 1:  # ranks hold different numbers of rows per video in clip mode: pad to the maximum
            cnt = torch.tensor([pred_prob.shape[0]], dtype=torch.int64, device=device)
            mx = cnt.clone()
            torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
            pad = int(mx.item()) - pred_prob.shape[0]
            pp = torch.cat([pred_prob, pred_prob.new_zeros(pad, pred_prob.shape[1])]).to(device)
            ll = torch.cat([lab, lab.new_zeros(pad)]).to(device)
            ll, pp = ddist.gather_for_metrics((ll, pp), valid=pred_prob.shape[0])
            lab, pred_prob = ll.cpu(), pp.cpu()
        labels_all.append(lab)
        probs_all.append(pred_prob)
    labels, probs = torch.cat(labels_all), torch.cat(probs_all)
    # the reference appends one dummy sample of each class before computing (inference.py:159-160)
    lab2 = torch.cat([labels, torch.tensor([0, 1])])
    sc2 = torch.cat([probs[:, 1], torch.tensor([0.0, 1.0])])
    pred2 = torch.cat([probs.argmax(dim=-1), torch.tensor([0, 1])])
    return dict(accuracy=round((pred2 == lab2).float().mean().item(), 3), roc_auc=round(binary_auroc(lab2, sc2), 3),
                labels=labels, probs=probs)
