"""`Detector.configure_optimizers`' SGD as ONE HIP launch (reference `src/models.py:740-754`: torch.optim.SGD, momentum
0.95, weight decay, over the trainable parameters; stepped once per batch by `src/trainer.py:157-177` under a OneCycleLR
that moves BOTH `lr` and `momentum` of the parameter group every step).

`FusedSGD` is a `torch.optim.Optimizer`: `param_groups[0]["lr"]` / `["momentum"]` are read at every step, `state[p]
["momentum_buffer"]` holds the velocity (a view into one flat buffer), `state_dict()` / `load_state_dict()` work as for
`torch.optim.SGD`, parameters without a gradient are skipped, a parameter's first step initialises its velocity with the
gradient — torch's semantics and torch's rounding order (csrc/optim.hip).  What differs is the mechanics: gradients are
packed into a persistent flat buffer by one multi-tensor copy, then `dfd_sgd_step` updates every parameter in one launch;
for the decoder's Linear weights that launch also rewrites the transposed f32 copy the decoder's row-streaming linears
read (`Decoder.weight_mirrors`), so no transpose kernel runs after an update.
"""
import torch

from . import capi


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.95, weight_decay=0.0, mirrors=None):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("lr, momentum and weight_decay must be non-negative")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False))
        # mirrors: None, or an object with `mirror_for(param) -> tensor [cols, rows] or None` and `mirrors_written(params)`
        # (the Decoder): the transposed copies this optimizer keeps in step with the weights
        self._mirrors = mirrors
        self._plans = {}

    def _plan(self, gi, group, active):
        """Flat buffers and device tables of group `gi` for the parameters that have a gradient this step."""
        key = tuple((p.data_ptr(), tuple(p.shape)) for p in active)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key and all(
                e["mirror"] is None or self._mirrors.current_mirror(e["p"]) is e["mirror"] for e in plan["entries"]):
            return plan  # same parameters, and the decoder still reads the transposed copies this plan writes
        dev = active[0].device
        for p in active:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise capi.DfdError("FusedSGD updates contiguous f32 parameters on the GPU")
        old = plan
        total = sum(p.numel() for p in active)
        gflat = torch.empty(total, device=dev, dtype=torch.float32)
        bflat = torch.zeros(total, device=dev, dtype=torch.float32)
        entries, off, blocks = [], 0, 0
        gviews = []
        for p in active:
            n = p.numel()
            gv, bv = gflat[off:off + n].view_as(p), bflat[off:off + n].view_as(p)
            st = self.state[p]
            seasoned = st.get("momentum_buffer") is not None
            if seasoned:
                bv.copy_(st["momentum_buffer"])  # carried over from the previous plan / a loaded state_dict
            st["momentum_buffer"] = bv
            mirror = self._mirrors.mirror_for(p) if (self._mirrors is not None and p.dim() == 2) else None
            rows, cols = (p.shape if mirror is not None else (0, 0))
            entries.append(dict(p=p, g=gv, buf=bv, mirror=mirror, numel=n, rows=rows, cols=cols, fresh=not seasoned))
            gviews.append(gv)
            off += n
        plan = dict(key=key, gflat=gflat, bflat=bflat, entries=entries, gviews=gviews, tables={}, old=None)
        self._plans[gi] = plan
        del old
        return plan

    @staticmethod
    def _table(entries, dev):
        rows, first = [], 0
        for e in entries:
            m = e["mirror"]
            rows.append([e["p"].data_ptr(), e["g"].data_ptr(), e["buf"].data_ptr(), 0 if m is None else m.data_ptr(), e["numel"],
                         int(e["rows"]) | (int(e["cols"]) << 32), first])
            first += capi.sgd_blocks(e["numel"], e["rows"], e["cols"], m is not None)
        return torch.tensor(rows, dtype=torch.int64).to(dev), first

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("nesterov") or group.get("dampening", 0) != 0:
                raise capi.DfdError("FusedSGD: plain momentum only (the reference's optimizer)")
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            plan = self._plan(gi, group, active)
            torch._foreach_copy_(plan["gviews"], [p.grad for p in active])
            # a parameter's first step copies the gradient into its velocity (torch.optim.SGD); all later ones blend
            fresh = [e for e in plan["entries"] if e["fresh"]]
            seasoned = [e for e in plan["entries"] if not e["fresh"]]
            for first, part in ((True, fresh), (False, seasoned)):
                if not part:
                    continue
                tkey = (first, tuple(id(e["p"]) for e in part))
                tab = plan["tables"].get(tkey)
                if tab is None:
                    tab = plan["tables"][tkey] = self._table(part, active[0].device)
                capi.sgd_step(tab[0], len(part), tab[1], group["lr"], group["momentum"], group["weight_decay"], first)
            # the kernel wrote through raw pointers: caches keyed on a parameter's version counter must see the update
            # (the call takes an ITERABLE of tensors; handed one tensor it would iterate over its rows)
            torch._C._increment_version([e["p"] for e in plan["entries"]])
            written = []
            for e in plan["entries"]:
                e["fresh"] = False
                if e["mirror"] is not None:
                    written.append((e["p"], e["mirror"]))
            if written:
                self._mirrors.mirrors_written(written)
        return loss
