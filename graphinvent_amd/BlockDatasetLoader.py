"""
Drop-in for the reference's ``BlockDatasetLoader`` module (BlockDatasetLoader.py:11-147): same class
names and constructor arguments, so the unchanged ``Workflow.get_dataloader`` (Workflow.py:120-141:
``HDFDataset(hdf_path)`` + ``BlockDataLoader(dataset=..., batch_size=..., block_size=..., shuffle=True,
n_workers=..., pin_memory=True)``) and ``Workflow.train_epoch`` (Workflow.py:766-798) get the MI355X
input pipeline when ``graphinvent_amd/`` is ahead of ``graphinvent/`` on ``sys.path``:

* a block stays int8 in pinned host memory (the HDF's dtype), minibatches are vectorised row
  gathers, copied to the GPU one batch ahead on a side stream (``loader.BlockStreamLoader``);
* the counting phase of ``graph_compact`` for the NEXT batch runs on that stream too
  (``ops.prefetch_compact``), so ``model(nodes, edges)`` in the unchanged loop finds its sizes on the
  host already: the forward's only host read-back and three kernels leave the step's critical path;
* the batches arrive on the device as int8; ``gnn.mpnn.GGNN.forward`` reads int8 directly, and the
  reference's ``Workflow.loss`` (target / target.sum -> KLDivLoss) works on them unchanged (integer
  true-division gives fp32).  ``batch = [b.to("cuda", non_blocking=True) ...]`` is then a no-op.

Like the reference, the data is streamed block by block (``block_size`` rows) and never has to fit in memory
(``loader.BlockStreamLoader``: a block's rows are read with one libhdf5 hyperslab read into pinned memory while
the previous block is consumed — peak pinned host memory is two blocks per rank, whatever the file size; the
reference's MOSES training set is 75 GB, tutorials/2_using_a_new_dataset.md:14-18).  Blocks are visited in a
shuffled order and shuffled inside, as in the reference (BlockDatasetLoader.py:42-61); a block's ragged last
minibatch is kept (``drop_last=False``; the reference's own drop condition, :47-50, is an operator-precedence
accident that is never true for files of more than one block).

Differences from the reference loader, all deliberate: the file is opened read-only through libhdf5 (ctypes; the
reference's ``h5py.File(path, "r+")`` needs write access and h5py); ``n_workers`` is ignored (no Python worker
processes: a minibatch is three memcpy-speed gathers); under ``torch.distributed`` every rank reads only its own
contiguous slice of each block (1 / world_size of the file) and all ranks yield minibatches of identical sizes
in lock-step, so the trailing ``block_rows % world_size`` rows of a block are skipped in that epoch.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

try:                                        # imported as graphinvent_amd.BlockDatasetLoader
    from .loader import ArraySource, BlockStreamLoader, HDFSource, LazyRows
except ImportError:                         # imported as top-level `BlockDatasetLoader` (drop-in layout)
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.append(_root)
    from graphinvent_amd.loader import ArraySource, BlockStreamLoader, HDFSource, LazyRows


class HDFDataset(torch.utils.data.Dataset):
    """``nodes`` / ``edges`` / ``APDs`` of a preprocessed GraphINVENT ``.h5`` file
    (BlockDatasetLoader.py:117-147).  Like the reference's h5py datasets nothing is read at construction;
    indexing (an int or a slice) reads those rows and returns fp32 tensors like the reference (:135-143); the
    loader below streams int8 blocks from the same source."""

    def __init__(self, path: str) -> None:
        self.path = path
        self.source = HDFSource(path)
        self.n_subgraphs = self.source.n_rows

    @classmethod
    def from_arrays(cls, nodes: np.ndarray, edges: np.ndarray, apds: np.ndarray) -> "HDFDataset":
        """The same dataset from int8 arrays already in memory (tests, synthetic data)."""
        self = cls.__new__(cls)
        self.path = None
        self.source = ArraySource(*(np.ascontiguousarray(a, dtype=np.int8) for a in (nodes, edges, apds)))
        self.n_subgraphs = self.source.n_rows
        return self

    def _rows(self, lo: int, hi: int):
        outs = tuple(np.empty((hi - lo,) + tuple(shp), dtype=np.int8) for shp in self.source.row_shapes)
        self.source.read_rows(lo, hi, outs)
        return outs

    # the reference exposes the three h5py datasets as attributes (BlockDatasetLoader.py:128-130): lazy here too —
    # ``ds.nodes.shape`` / ``len(ds.nodes)`` read nothing, ``ds.nodes[lo:hi]`` reads those rows of that dataset only
    @property
    def nodes(self): return LazyRows(self.source, 0)

    @property
    def edges(self): return LazyRows(self.source, 1)

    @property
    def apds(self): return LazyRows(self.source, 2)

    def __getitem__(self, idx) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        if isinstance(idx, slice):
            lo, hi, step = idx.indices(self.n_subgraphs)
            if step != 1:
                raise IndexError("HDFDataset slices must be contiguous")
            rows = self._rows(lo, max(hi, lo))
            return tuple(torch.from_numpy(a).type(torch.float32) for a in rows)
        i = int(idx)
        if i < 0:
            i += self.n_subgraphs
        return tuple(torch.from_numpy(a[0]).type(torch.float32) for a in self._rows(i, i + 1))

    def __len__(self) -> int:
        return self.n_subgraphs


class BlockDataLoader:
    """Same constructor as the reference's ``BlockDataLoader`` (BlockDatasetLoader.py:17-31); iterates
    ``(nodes, edges, apds)`` int8 minibatches resident on the GPU, one batch ahead of the consumer, streaming
    the file block by block (``block_size`` rows at a time, two pinned blocks per rank)."""

    def __init__(self, dataset: HDFDataset, batch_size: int = 100, block_size: int = 10000,
                 shuffle: bool = True, n_workers: int = 0, pin_memory: bool = True,
                 device: str = "cuda", seed: int = 0, drop_zero_targets: bool = False,
                 drop_last: bool = False) -> None:
        self.dataset, self.batch_size, self.block_size = dataset, batch_size, block_size
        self.shuffle, self.n_workers, self.pin_memory = shuffle, n_workers, pin_memory
        rank, world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        # drop_zero_targets=False keeps every row like the reference (all-zero target rows give a NaN
        # loss there too, DataProcesser.py:268-269); drop_last=False keeps ragged last minibatches like it
        self._loader = BlockStreamLoader(dataset.source, batch_size, block_size=max(block_size, batch_size),
                                         rank=rank, world_size=world, seed=seed, shuffle=shuffle,
                                         device=device if torch.cuda.is_available() else None,
                                         drop_last=drop_last, drop_zero_targets=drop_zero_targets)
        if len(self._loader) == 0 and len(dataset) > 0:
            raise ValueError(f"{len(dataset)} rows give no minibatch of {batch_size} on {world} rank(s) with "
                             f"drop_last={drop_last}: validation_epoch would average an empty list (NaN)")
        self._epoch = 0

    def __iter__(self):
        self._loader.set_epoch(self._epoch)     # a new permutation every pass, identical on all ranks
        self._epoch += 1
        return iter(self._loader)

    def __len__(self) -> int:
        return len(self._loader)
