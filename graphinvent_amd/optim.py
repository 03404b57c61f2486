"""
``FusedAdam`` — ``torch.optim.Adam`` (the optimizer the reference builds at Workflow.py:219-263) as ONE
HIP launch per step over a flat fp32 parameter bucket.

On construction the parameters of each group are re-pointed into one flat, 16-byte-segmented buffer
(``p.data`` becomes a view; ``Parameter`` identity, ``state_dict`` keys and ``load_state_dict`` are
unaffected).  The segment layout is the one the fused GGNN backward uses for its gradient bucket
(``gnn.mpnn.ggnn_backward_raw``), so when the gradients already live in that bucket the step reads
them in place; otherwise they are packed first.  Learning-rate schedulers work as usual
(``group["lr"]`` is read every step).  Semantics: Adam without amsgrad, ``weight_decay`` as L2 term,
bias correction as in torch (computed in double on the host).  Parameters with ``requires_grad=False`` are
not part of the bucket; parameters whose ``grad`` is None at a step are skipped like torch.optim.Adam skips
them (no moment decay, no weight decay: the step then runs as one launch per run of parameters that do have
gradients).  One deviation remains: the bias-correction step count is kept per GROUP, not per parameter, so
a parameter that skipped steps is corrected with the group's count (torch: with its own).
"""
from __future__ import annotations

from typing import List

import torch

from . import lib as L


def _pack_offsets(params) -> (List[int], int):
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) & ~3
    return offs, total


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        L.load()
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if not all(p.is_cuda and p.dtype == torch.float32 and p.device == dev for p in ps):
                raise RuntimeError("FusedAdam needs fp32 CUDA parameters on one device "
                                   "(call model.to('cuda') first)")
            offs, total = _pack_offsets(ps)
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            with torch.no_grad():
                for p, o in zip(ps, offs):
                    flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                    p.data = flat[o:o + p.numel()].view_as(p)
            self._flat.append(dict(params=ps, offs=offs, total=total, p=flat,
                                   m=torch.zeros_like(flat), v=torch.zeros_like(flat),
                                   g=None, step=0))
        self._built = True

    def add_param_group(self, param_group):
        if getattr(self, "_built", False):
            raise RuntimeError("FusedAdam packs its parameters into flat buckets at construction; "
                               "adding a parameter group afterwards is not supported")
        super().add_param_group(param_group)

    # ---- checkpointing: torch.optim.Adam's state layout ------------------------------------------
    # The moments live in the flat buckets; `state_dict()` exposes them per parameter exactly as
    # torch.optim.Adam would ({"step", "exp_avg", "exp_avg_sq"}), `load_state_dict()` copies them back,
    # so a resume keeps the moments and the bias-correction step count (Workflow.py:219-263 restarts).
    def state_dict(self):
        for st in self._flat:
            if st is None or st["step"] == 0:
                continue
            for p, o in zip(st["params"], st["offs"]):
                n = p.numel()
                self.state[p] = dict(step=torch.tensor(float(st["step"])),
                                     exp_avg=st["m"][o:o + n].view_as(p).clone(),
                                     exp_avg_sq=st["v"][o:o + n].view_as(p).clone())
        try:
            return super().state_dict()
        finally:
            self.state.clear()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for st in self._flat:
                if st is None:
                    continue
                steps = set()
                for p, o in zip(st["params"], st["offs"]):
                    ps = self.state.get(p)
                    if not ps:
                        continue
                    n = p.numel()
                    st["m"][o:o + n].copy_(ps["exp_avg"].reshape(-1))
                    st["v"][o:o + n].copy_(ps["exp_avg_sq"].reshape(-1))
                    steps.add(int(ps["step"]))
                if len(steps) > 1:
                    raise RuntimeError("FusedAdam: parameters of one group carry different step counts")
                if steps:
                    st["step"] = steps.pop()
        self.state.clear()

    def _grad_bucket(self, st) -> torch.Tensor:
        ps, offs = st["params"], st["offs"]
        g0 = ps[0].grad
        if g0 is not None:
            base = g0.data_ptr()
            store = g0.untyped_storage()
            if all(p.grad is not None and p.grad.is_contiguous() and
                   p.grad.data_ptr() == base + 4 * o for p, o in zip(ps, offs)) and \
                    base % 16 == 0 and store.nbytes() - (base - store.data_ptr()) >= 4 * st["total"]:
                # gradients already form the flat bucket (fused GGNN backward): zero-copy view
                return torch.as_strided(g0, (st["total"],), (1,))
        if st["g"] is None:
            st["g"] = torch.zeros(st["total"], dtype=torch.float32, device=st["p"].device)
        for p, o in zip(ps, offs):
            seg = st["g"][o:o + p.numel()]
            if p.grad is None:
                seg.zero_()
            else:
                seg.copy_(p.grad.reshape(-1))
        return st["g"]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for group, st in zip(self.param_groups, self._flat):
            if st is None:
                continue
            with torch.cuda.device(st["p"].device):       # launch on the bucket's device, whichever is current
                ps, offs = st["params"], st["offs"]
                have = [p.grad is not None for p in ps]
                if not any(have):
                    continue
                g = self._grad_bucket(st)
                st["step"] += 1
                b1, b2 = group["betas"]
                stream = torch.cuda.current_stream(st["p"].device).cuda_stream
                # one launch over the whole bucket, or one per run of consecutive parameters with gradients
                runs = [(0, st["total"])] if all(have) else []
                if not runs:
                    i = 0
                    while i < len(ps):
                        if have[i]:
                            j = i
                            while j + 1 < len(ps) and have[j + 1]:
                                j += 1
                            runs.append((offs[i], offs[j] + ((ps[j].numel() + 3) & ~3)))
                            i = j + 1
                        else:
                            i += 1
                for lo, hi in runs:
                    L.check(lib.gi_adam_step(st["p"].data_ptr() + 4 * lo, g.data_ptr() + 4 * lo,
                                             st["m"].data_ptr() + 4 * lo, st["v"].data_ptr() + 4 * lo, hi - lo,
                                             float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                             st["step"], stream), "gi_adam_step")
        L.WEIGHTS_EPOCH[0] += 1        # the kernel wrote through raw pointers: no version counter saw it
        return loss
