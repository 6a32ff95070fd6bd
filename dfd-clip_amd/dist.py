"""Multi-GPU plumbing: one process per GPU over `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

Clips are independent through the whole path, so ranks shard the batch and the forward has no
exchange step.  The reference's cross-rank traffic is Accelerate/DDP: the
gradient all-reduce of the trainable (decoder) parameters after each backward (C1,
`src/trainer.py:157-165`), the all-gather of per-clip outputs for metrics with the padded
tail dropped (C2/C3, `src/callbacks/metrics.py:98-99`, `inference.py:147-149`), the initial
parameter broadcast and barriers (C4).  Here they are three explicit calls.

`allreduce_gradients` sends ONE flat fp32 buffer (156 MB for ViT-B/16's decoder): xGMI is
point-to-point, 7 links per GPU, so a single large collective that RCCL can spread over every
link beats the per-bucket pattern DDP would issue; the encoder forward of the next micro-batch
does not depend on it, and it is ~1 % of a step.
"""
import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


_STAGE_GLOO = True
_PINNED = {}
_FLAT = {}


def _pinned_like(t):
    key = (t.numel(), t.dtype)
    buf = _PINNED.get(key)
    if buf is None:
        _PINNED.clear()
        buf = _PINNED[key] = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
    return buf


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s parameters and buffers (DDP does this at wrap time)."""
    if world_size() == 1:
        return
    tensors = list(module.parameters()) + list(module.buffers())
    if not tensors:
        return
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for group in by_dtype.values():
            flat = _flatten_dense_tensors([t.detach() for t in group])
            dist.broadcast(flat, src)
            for t, f in zip(group, _unflatten_dense_tensors(flat, group)):
                t.copy_(f)  # on the parameter itself: bumps its version counter, which the weight caches key on
    for sub in module.modules():  # device-side copies derived from the parameters (bf16 / transposed weights, graphs)
        if hasattr(sub, "invalidate_caches"):
            sub.invalidate_caches()


def allreduce_gradients(params):
    """Average `.grad` of the given parameters over ranks, in place, with one flat all-reduce.
    Parameters without a gradient on this rank contribute zeros (DDP's find_unused_parameters)."""
    n = world_size()
    params = [p for p in params if p.requires_grad]
    if n == 1 or not params:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    # one persistent flat buffer per (size, dtype, device) and two multi-tensor copies (in, out) instead of a fresh
    # 156 MB concatenation and one copy kernel per parameter every step
    key = (sum(g.numel() for g in grads), grads[0].dtype, grads[0].device)
    if any(g.dtype != key[1] or g.device != key[2] for g in grads):
        flat, views = _flatten_dense_tensors(grads), None  # mixed dtypes / devices: the general path
    else:
        ent = _FLAT.get(key)
        if ent is None or [v.shape for v in ent[1]] != [g.shape for g in grads]:
            _FLAT.clear()
            flat = torch.empty(key[0], dtype=key[1], device=key[2])
            ent = _FLAT[key] = (flat, _unflatten_dense_tensors(flat, grads))
        flat, views = ent
        torch._foreach_copy_(list(views), grads)
    if flat.is_cuda and dist.get_backend() == "gloo" and _STAGE_GLOO:
        # gloo has no device path: ProcessGroupGloo stages a CUDA tensor through a pinned host buffer it allocates
        # per call.  HYPOTHESIS for the multi-second steps of round 1's one-GPU rehearsal (never reproduced since:
        # DESIGN.md §6): entered while the device still has work queued, that allocation cannot reuse the previous
        # step's block (its copy event is still pending), so every step pins and unpins another 156 MB.  Not measured;
        # what IS known is that those steps were host-bound (host CPU time == wall time).  Staging through ONE
        # persistent pinned buffer removes the per-call allocation either way; RCCL ("nccl") reduces device buffers
        # directly and never comes here.
        host = _pinned_like(flat)
        host.copy_(flat, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host, non_blocking=True)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(n)
    torch._foreach_copy_(grads, list(views) if views is not None else list(_unflatten_dense_tensors(flat, grads)))


def gather_for_metrics(tensors, valid=None):
    """All-gather a tuple of per-sample tensors (same leading size on every rank) along dim 0 in
    rank order.  `valid`: number of real samples this rank holds (the rest is padding added to keep
    shapes equal — the duplicated tail the reference drops after gathering); defaults to all."""
    n = world_size()
    single = not isinstance(tensors, (tuple, list))
    ts = [tensors] if single else list(tensors)
    if n == 1:
        out = [t if valid is None else t[:valid] for t in ts]
        return out[0] if single else tuple(out)
    dev = ts[0].device
    cnt = torch.tensor([ts[0].shape[0] if valid is None else valid], device=dev, dtype=torch.int64)
    counts = [torch.zeros_like(cnt) for _ in range(n)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    out = []
    for t in ts:
        parts = [torch.empty_like(t) for _ in range(n)]
        dist.all_gather(parts, t.contiguous())
        out.append(torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0))
    return out[0] if single else tuple(out)


def barrier():
    if world_size() > 1:
        dist.barrier()
