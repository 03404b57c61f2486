"""
Data-parallel training of the GGNN hot path: one process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on the MI355X node; ``gloo`` in the CPU tests).

The reference trains single-process, single-GPU (Workflow.py:766-798; no DDP / NCCL call site
anywhere).  Graphs are independent units, so the path shards by batch with exactly
ONE exchange step per optimizer step:

* every rank draws its own disjoint slice of each global minibatch from the same shuffled block
  (``ShardedBatchSampler``: identical permutation on all ranks from a shared seed, rank r takes
  global batches r, r+W, ...; ragged tails dropped so ranks stay in lock-step);
* forward / loss / backward are rank-local;
* gradients are averaged with one all-reduce over a single flat fp32 bucket.  The fused HIP
  backward already writes every parameter gradient into one contiguous buffer
  (``GGNN._grad_bucket``), so on the GPU path the collective runs in place on that buffer — no
  flatten / unflatten copies.  xGMI is point-to-point, a ring all-reduce is per-link bound: one
  24 MB bucket instead of 104 small tensors keeps it at a single latency term;
* on the HIP model the exchange OVERLAPS the backward: the readout's gradients (gather + APDReadout,
  ~86 % of the bucket and its contiguous tail) are final before the message passes are
  differentiated, so the backward is issued in two calls and the tail's all-reduce starts on a
  communication stream as soon as an event says it is complete; only the small head (message MLPs +
  GRU) is exchanged after the backward;
* identical Adam steps follow on every rank (``batchmean`` loss per rank + gradient mean =
  global-batch mean when the per-rank batch sizes are equal).
"""
from __future__ import annotations

import os
from typing import Callable, Iterator, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import lib as _lib
from .loss import apd_kl_loss, register_unit_gradient


class ShardedBatchSampler:
    """Index batches of one rank for one epoch over ``n_rows`` examples of a block
    (mirrors BlockDatasetLoader.py:32-63's shuffled minibatches; per-rank sharding is new)."""

    def __init__(self, n_rows: int, batch_size: int, rank: int = 0, world_size: int = 1,
                 seed: int = 0, shuffle: bool = True):
        if not (0 <= rank < world_size):
            raise ValueError("rank out of range")
        self.n_rows, self.batch_size = n_rows, batch_size
        self.rank, self.world_size, self.seed, self.shuffle = rank, world_size, seed, shuffle
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return (self.n_rows // self.batch_size) // self.world_size

    def __iter__(self) -> Iterator[np.ndarray]:
        order = np.arange(self.n_rows)
        if self.shuffle:
            order = np.random.default_rng([self.seed, self.epoch]).permutation(self.n_rows)
        for k in range(len(self)):
            g = k * self.world_size + self.rank                 # global batch index
            yield order[g * self.batch_size:(g + 1) * self.batch_size]


class DataParallel:
    """Wraps (model, optimizer[, scheduler]) into a data-parallel train step."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer,
                 scheduler=None, loss_fn: Callable = apd_kl_loss,
                 process_group: Optional[dist.ProcessGroup] = None, always_reduce: bool = False,
                 overlap: bool = True):
        self.model, self.optimizer, self.scheduler, self.loss_fn = model, optimizer, scheduler, loss_fn
        self.group = process_group
        self.always_reduce = always_reduce     # run the collective even at world size 1 (tests)
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        self.last_bucket_zero_copy = False
        self.last_overlapped = False
        self._pending = None           # (work handle, split offset) of the early tail all-reduce
        self._one = None               # cached root gradient (see step)
        self._comm_stream = None
        # overlap needs the HIP model (it exposes the flat bucket and the two-call backward)
        self.overlap = (overlap
                        and (self.world_size > 1 or always_reduce) and dist.is_initialized()
                        and hasattr(model, "_grad_bucket") and torch.cuda.is_available())

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every rank start from rank `src`'s weights."""
        if self.world_size == 1:
            return
        with torch.no_grad():
            for p in self.params:
                dist.broadcast(p.data, src=src, group=self.group)
        _lib.WEIGHTS_EPOCH[0] += 1

    # -- the one exchange step ---------------------------------------------------------------
    def _model_bucket(self) -> Optional[torch.Tensor]:
        bucket = getattr(self.model, "_grad_bucket", None)
        if bucket is None:
            return None
        base = bucket.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                return None
        return bucket

    def _early_allreduce(self, gflat: torch.Tensor, split: int, ready: "torch.cuda.Event") -> None:
        """Called from inside the HIP backward once gflat[split:] (readout gradients) is queued to
        be complete at `ready`: start its all-reduce on the communication stream."""
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=gflat.device)
        comm = self._comm_stream
        comm.wait_event(ready)
        # single shot: a second backward through the model before the exchange is finished must not
        # start another collective on (or accumulate into) the buffer this one reduces in place
        self.model._grad_ready_hook = None
        self.model._early_exchange_pending = True
        with torch.cuda.stream(comm):
            work = dist.all_reduce(gflat[split:], op=dist.ReduceOp.SUM, group=self.group,
                                   async_op=True)
        self._pending = (work, split, gflat.data_ptr(), gflat)

    def allreduce_gradients(self) -> None:
        if self.world_size == 1 and not self.always_reduce:
            return
        bucket = self._model_bucket()
        self.last_bucket_zero_copy = bucket is not None
        pending, self._pending = self._pending, None
        self.last_overlapped = False
        if hasattr(self.model, "_early_exchange_pending"):
            self.model._early_exchange_pending = False
        if pending is not None and (bucket is None or pending[2] != bucket.data_ptr()):
            # The early collective has SUM-reduced the tail of ITS buffer in place, but the
            # gradients the optimizer will read are not (all) views of that buffer any more
            # (cloned / accumulated / replaced after the backward): reducing them again would count
            # the tail world_size times.  There is no safe way to merge the two; fail loudly.
            pending[0].wait()
            raise RuntimeError("data-parallel overlap: an early all-reduce of the readout gradients "
                               "is pending, but param.grad are no longer views of the model's flat "
                               "gradient bucket; construct DataParallel(overlap=False) "
                               "for training loops that clone, accumulate or "
                               "replace gradients between backward and the optimizer step")
        if bucket is not None:                      # gradients already live in one flat buffer
            if pending is not None:
                work, split = pending[0], pending[1]   # tail is already being exchanged: head now
                if split > 0:
                    dist.all_reduce(bucket[:split], op=dist.ReduceOp.SUM, group=self.group)
                work.wait()                         # current stream waits for the tail's collective
                bucket.record_stream(self._comm_stream)
                self.last_overlapped = True
            else:
                dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
            bucket.mul_(1.0 / self.world_size)
            return
        grads = [p.grad for p in self.params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world_size)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def step(self, nodes: torch.Tensor, edges: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """forward -> zero_grad -> loss -> backward -> all-reduce -> optimizer (-> scheduler):
        the order of Workflow.py:785-796 with the exchange step inserted before the update."""
        output = self.model(nodes, edges)
        for p in self.params:                      # optimizer.zero_grad(set_to_none=True), minus its
            p.grad = None                          # per-call bookkeeping (0.09 ms of host time)
        loss = self.loss_fn(output, target)
        # the early exchange is armed only around this backward: gradients are fresh views of the
        # backward's own bucket here (zero_grad(set_to_none) above), which the overlap relies on
        if self.overlap:
            self.model._grad_ready_hook = self._early_allreduce   # see gnn/mpnn.py ggnn_backward_raw
        try:
            # the root gradient is a cached device scalar: autograd would otherwise launch a fill kernel
            # for ones_like(loss) every step
            one = self._one
            if one is None or one.device != loss.device or one.dtype != loss.dtype:
                one = self._one = register_unit_gradient(
                    torch.ones((), dtype=loss.dtype, device=loss.device))
            loss.backward(one)
        finally:
            if self.overlap:
                self.model._grad_ready_hook = None
        self.allreduce_gradients()
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss.detach()
