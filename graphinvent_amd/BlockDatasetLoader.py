"""
Drop-in for the reference's ``BlockDatasetLoader`` module (BlockDatasetLoader.py:11-147): same class
names and constructor arguments, so the unchanged ``Workflow.get_dataloader`` (Workflow.py:120-141:
``HDFDataset(hdf_path)`` + ``BlockDataLoader(dataset=..., batch_size=..., block_size=..., shuffle=True,
n_workers=..., pin_memory=True)``) and ``Workflow.train_epoch`` (Workflow.py:766-798) get the MI355X
input pipeline when ``graphinvent_amd/`` is ahead of ``graphinvent/`` on ``sys.path``:

* the block stays int8 in pinned host memory (the HDF's dtype), minibatches are vectorised row
  gathers, copied to the GPU one batch ahead on a side stream (``loader.ShardedBlockLoader``);
* the counting phase of ``graph_compact`` for the NEXT batch runs on that stream too
  (``ops.prefetch_compact``), so ``model(nodes, edges)`` in the unchanged loop finds its sizes on the
  host already: the forward's only host read-back and three kernels leave the step's critical path;
* the batches arrive on the device as int8; ``gnn.mpnn.GGNN.forward`` reads int8 directly, and the
  reference's ``Workflow.loss`` (target / target.sum -> KLDivLoss) works on them unchanged (integer
  true-division gives fp32).  ``batch = [b.to("cuda", non_blocking=True) ...]`` is then a no-op.

Differences from the reference loader, all deliberate: the file is opened read-only through libhdf5
(ctypes; the reference's ``h5py.File(path, "r+")`` needs write access and h5py); a ragged last
minibatch is dropped (``len()`` = rows // batch_size) so that data-parallel ranks stay in lock-step;
shuffling is over the whole file with a per-epoch seeded permutation instead of block-wise (the file is
in RAM anyway — ``block_size`` is accepted and ignored); ``n_workers`` is ignored (no Python worker
processes: a minibatch is three memcpy-speed gathers).  Rank / world size for sharding are taken
from ``torch.distributed`` when it is initialised.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

try:                                        # imported as graphinvent_amd.BlockDatasetLoader
    from .loader import ShardedBlockLoader, read_hdf_int8
except ImportError:                         # imported as top-level `BlockDatasetLoader` (drop-in layout)
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.append(_root)
    from graphinvent_amd.loader import ShardedBlockLoader, read_hdf_int8


class HDFDataset(torch.utils.data.Dataset):
    """``nodes`` / ``edges`` / ``APDs`` of a preprocessed GraphINVENT ``.h5`` file
    (BlockDatasetLoader.py:117-147).  Indexing returns fp32 tensors like the reference (:135-143);
    the loader below uses the int8 arrays directly."""

    def __init__(self, path: str) -> None:
        self.path = path
        self.nodes, self.edges, self.apds = read_hdf_int8(path)
        self.n_subgraphs = self.nodes.shape[0]

    @classmethod
    def from_arrays(cls, nodes: np.ndarray, edges: np.ndarray, apds: np.ndarray) -> "HDFDataset":
        """The same dataset from int8 arrays already in memory (tests, synthetic data)."""
        self = cls.__new__(cls)
        self.path = None
        self.nodes, self.edges, self.apds = (np.ascontiguousarray(a, dtype=np.int8)
                                             for a in (nodes, edges, apds))
        self.n_subgraphs = self.nodes.shape[0]
        return self

    def __getitem__(self, idx) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return tuple(torch.from_numpy(np.asarray(a[idx])).type(torch.float32)
                     for a in (self.nodes, self.edges, self.apds))

    def __len__(self) -> int:
        return self.n_subgraphs


class BlockDataLoader:
    """Same constructor as the reference's ``BlockDataLoader`` (BlockDatasetLoader.py:17-31); iterates
    ``(nodes, edges, apds)`` int8 minibatches resident on the GPU, one batch ahead of the consumer."""

    def __init__(self, dataset: HDFDataset, batch_size: int = 100, block_size: int = 10000,
                 shuffle: bool = True, n_workers: int = 0, pin_memory: bool = True,
                 device: str = "cuda", seed: int = 0, drop_zero_targets: bool = False) -> None:
        self.dataset, self.batch_size, self.block_size = dataset, batch_size, block_size
        self.shuffle, self.n_workers, self.pin_memory = shuffle, n_workers, pin_memory
        rank, world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        # drop_zero_targets=False keeps every row like the reference (all-zero target rows give a NaN
        # loss there too, DataProcesser.py:268-269)
        self._loader = ShardedBlockLoader(dataset.nodes, dataset.edges, dataset.apds, batch_size,
                                          rank=rank, world_size=world, seed=seed, shuffle=shuffle,
                                          device=device if torch.cuda.is_available() else None,
                                          drop_zero_targets=drop_zero_targets)
        self._epoch = 0

    def __iter__(self):
        self._loader.set_epoch(self._epoch)     # a new permutation every pass, identical on all ranks
        self._epoch += 1
        return iter(self._loader)

    def __len__(self) -> int:
        return len(self._loader)
