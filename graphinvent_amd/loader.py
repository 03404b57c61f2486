"""
Input pipeline for the GGNN hot path: what feeds ``model(nodes, edges)``.

The reference's ``BlockDatasetLoader`` (BlockDatasetLoader.py:11-147) loads a block of the
preprocessed HDF (int8 ``nodes``, ``edges``, ``APDs``; DataProcesser.py:157-161,273-289) into RAM,
converts it to **fp32 on the host** (:139-141), and yields shuffled minibatches built sample by
sample in Python worker processes (:110-111); ``Workflow.train_epoch`` then copies three fp32
tensors to the device (Workflow.py:781-782).  At the MI355X path's step time (3.7 ms for 1000
graphs) that is the first thing to cap throughput, so here:

* the block stays **int8** in (pinned) host memory — 1.2 KB/graph instead of 4.9 KB at GDB-13;
* a minibatch is three vectorised row gathers (``index_select``) into a pinned staging buffer and one
  asynchronous H2D copy each, issued on a side HIP stream **one batch ahead** of the consumer
  (double buffering), so the copy of batch k+1 overlaps the training step on batch k;
* the int8 tensors go **straight into the model and the loss** — ``graph_compact`` and the fused KL
  kernel read int8 (``GI_DTYPE_I8``); no fp32 copy of the inputs is ever materialised;
* every rank reads only its own slice of each block (``BlockStreamLoader``).

Rows whose APD target is all zero (dataset-size padding, DataProcesser.py:268-269; reference loss
= NaN on them) can be dropped up front with ``drop_zero_targets=True``.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Iterator, Optional, Tuple

import numpy as np
import torch



_HDF_NAMES = (b"nodes", b"edges", b"APDs")

# libhdf5 keeps global library state and most builds are not thread-safe: EVERY H5* call of this process goes
# through one lock per loaded library handle (two HDFSource objects — the training file read by the loader's
# background thread, the validation file read by the main thread — must serialise against each other too).
_HDF_LIBS = {}
_HDF_LIBS_GUARD = threading.Lock()


def _load_libhdf5(libhdf5: Optional[str] = None):
    """(library handle, H5T_NATIVE_INT8 id, the process-wide lock of that handle)."""
    with _HDF_LIBS_GUARD:
        if libhdf5 in _HDF_LIBS:
            return _HDF_LIBS[libhdf5]
        entry = _open_libhdf5(libhdf5)
        for known in _HDF_LIBS.values():                      # the same .so reached under two names: one lock
            if known[0]._handle == entry[0]._handle:
                entry = known
                break
        _HDF_LIBS[libhdf5] = entry
        return entry


def _open_libhdf5(libhdf5: Optional[str] = None):
    candidates = [libhdf5] if libhdf5 else ["libhdf5.so", "/opt/conda/lib/libhdf5.so",
                                            "libhdf5_serial.so"]
    for c in candidates:
        try:
            lib = ctypes.CDLL(c)
            break
        except OSError:
            continue
    else:
        raise RuntimeError("no libhdf5 shared library found (tried %s)" % candidates)
    i64 = ctypes.c_int64
    lib.H5open()
    for fn in ("H5Fopen", "H5Dopen2", "H5Dget_space", "H5Screate_simple"):
        getattr(lib, fn).restype = i64
    lib.H5Fopen.argtypes = [ctypes.c_char_p, ctypes.c_uint, i64]
    lib.H5Dopen2.argtypes = [i64, ctypes.c_char_p, i64]
    lib.H5Dget_space.argtypes = [i64]
    lib.H5Sget_simple_extent_ndims.argtypes = [i64]
    lib.H5Sget_simple_extent_dims.argtypes = [i64, ctypes.c_void_p, ctypes.c_void_p]
    lib.H5Screate_simple.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.H5Sselect_hyperslab.argtypes = [i64, ctypes.c_int] + [ctypes.c_void_p] * 4
    lib.H5Dread.argtypes = [i64] * 5 + [ctypes.c_void_p]
    for fn in ("H5Fclose", "H5Dclose", "H5Sclose"):
        getattr(lib, fn).argtypes = [i64]
    return lib, i64.in_dll(lib, "H5T_NATIVE_INT8_g").value, threading.Lock()


class HDFSource:
    """Row-range reads of the three int8 datasets of a GraphINVENT preprocessed ``.h5`` file
    (DataProcesser.py:157-161: contiguous, uncompressed int8) through libhdf5 (ctypes; h5py is not a
    dependency), opened read-only (the reference opens ``"r+"``, BlockDatasetLoader.py:125).  The file is never
    held in memory: ``read_rows(lo, hi, outs)`` selects the hyperslab of rows [lo, hi) of every dataset and
    reads it straight into the caller's (pinned) buffers.  One thread at a time may call into it (lock)."""

    def __init__(self, path: str, libhdf5: Optional[str] = None):
        self.path = path
        self._lib, self._int8, self._lock = _load_libhdf5(libhdf5)
        self._f, self._d, self.row_shapes = None, [], []
        with self._lock:
            self._open(path)

    def _open(self, path: str) -> None:
        lib = self._lib
        self._f = lib.H5Fopen(path.encode(), 0, 0)                  # H5F_ACC_RDONLY
        if self._f < 0:
            self._f = None
            raise OSError(f"cannot open {path}")
        n_rows = None
        for name in _HDF_NAMES:
            d = lib.H5Dopen2(self._f, name, 0)
            if d < 0:
                raise KeyError(name.decode())
            sp = lib.H5Dget_space(d)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (ctypes.c_uint64 * nd)()
            lib.H5Sget_simple_extent_dims(sp, dims, None)
            lib.H5Sclose(sp)
            dims = tuple(int(x) for x in dims)
            if n_rows is not None and dims[0] != n_rows:
                raise ValueError("nodes / edges / APDs disagree on the number of rows")
            n_rows = dims[0]
            self._d.append(d)
            self.row_shapes.append(dims[1:])
        self.n_rows = n_rows

    def read_rows(self, lo: int, hi: int, outs, which=(0, 1, 2)) -> None:
        """rows [lo, hi) of (nodes, edges, APDs) into the first hi - lo rows of the int8 arrays `outs`;
        `which` selects the datasets (indices into nodes / edges / APDs) `outs` corresponds to."""
        lib, n = self._lib, hi - lo
        if not 0 <= lo <= hi <= self.n_rows:
            raise IndexError((lo, hi, self.n_rows))
        if len(outs) != len(which):
            raise ValueError("one destination per selected dataset")
        if n == 0:
            return
        with self._lock:
            if self._f is None:
                raise ValueError(f"{self.path} is closed")
            for w, out in zip(which, outs):
                d, shp = self._d[w], self.row_shapes[w]
                arr = out.numpy() if torch.is_tensor(out) else out
                if arr.dtype != np.int8 or arr.shape[1:] != shp or arr.shape[0] < n or not arr.flags.c_contiguous:
                    raise ValueError("destination must be a C-contiguous int8 array of >= hi - lo rows")
                nd = 1 + len(shp)
                start = (ctypes.c_uint64 * nd)(lo, *([0] * len(shp)))
                count = (ctypes.c_uint64 * nd)(n, *shp)
                fsp = lib.H5Dget_space(d)
                msp = lib.H5Screate_simple(nd, count, None)
                ok = lib.H5Sselect_hyperslab(fsp, 0, start, None, count, None) >= 0 and \
                    lib.H5Dread(d, self._int8, msp, fsp, 0, arr.ctypes.data) >= 0
                lib.H5Sclose(msp); lib.H5Sclose(fsp)
                if not ok:
                    raise OSError(f"H5Dread failed for rows [{lo}, {hi}) of {self.path}")

    def close(self) -> None:
        with self._lock:
            if self._f is not None:
                for d in self._d:
                    self._lib.H5Dclose(d)
                self._lib.H5Fclose(self._f)
                self._f, self._d = None, []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArraySource:
    """The same interface over int8 arrays already in memory (tests, synthetic data, np.memmap)."""

    def __init__(self, nodes, edges, apds):
        self.arrays = tuple(np.asarray(a) for a in (nodes, edges, apds))
        if any(a.dtype != np.int8 for a in self.arrays):
            raise TypeError("the preprocessed HDF arrays are int8")
        if len({a.shape[0] for a in self.arrays}) != 1:
            raise ValueError("nodes / edges / APDs disagree on the number of rows")
        self.n_rows = self.arrays[0].shape[0]
        self.row_shapes = [a.shape[1:] for a in self.arrays]

    def read_rows(self, lo: int, hi: int, outs, which=(0, 1, 2)) -> None:
        if not 0 <= lo <= hi <= self.n_rows:
            raise IndexError((lo, hi, self.n_rows))
        for w, out in zip(which, outs):
            dst = out.numpy() if torch.is_tensor(out) else out
            dst[:hi - lo] = self.arrays[w][lo:hi]


class LazyRows:
    """One dataset of a source as a read-on-index view: what the reference's ``HDFDataset.nodes`` / ``.edges`` /
    ``.apds`` are (h5py datasets, BlockDatasetLoader.py:128-130) — ``shape``, ``len`` and ``view[i]`` /
    ``view[lo:hi]`` (int8 numpy, only those rows of only this dataset are read)."""

    def __init__(self, source, which: int):
        self._src, self._which = source, which
        self.shape = (source.n_rows,) + tuple(source.row_shapes[which])
        self.dtype = np.dtype(np.int8)
        self.ndim = len(self.shape)

    def __len__(self) -> int:
        return self.shape[0]

    def __getitem__(self, idx):
        n = self.shape[0]
        if isinstance(idx, slice):
            lo, hi, step = idx.indices(n)
            if step != 1:
                raise IndexError("row slices must be contiguous")
            hi = max(hi, lo)
            out = np.empty((hi - lo,) + self.shape[1:], dtype=np.int8)
            self._src.read_rows(lo, hi, (out,), which=(self._which,))
            return out
        i = int(idx)
        if i < 0:
            i += n
        out = np.empty((1,) + self.shape[1:], dtype=np.int8)
        self._src.read_rows(i, i + 1, (out,), which=(self._which,))
        return out[0]

    def __array__(self, dtype=None, copy=None):                # np.asarray(view): the caller asked for it all
        a = self[0:self.shape[0]]
        return a if dtype is None else a.astype(dtype)


def read_hdf_int8(path: str, libhdf5: Optional[str] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``nodes, edges, APDs`` of a GraphINVENT preprocessed ``.h5`` file, whole, as int8 arrays (small files,
    tests; training reads blocks through ``BlockStreamLoader``).  Raises if no libhdf5 can be loaded."""
    src = HDFSource(path, libhdf5)
    outs = tuple(np.empty((src.n_rows,) + shp, dtype=np.int8) for shp in src.row_shapes)
    src.read_rows(0, src.n_rows, outs)
    src.close()
    return outs


class BlockStreamLoader:
    """Block-wise, rank-sharded minibatches of a dataset that need not fit in memory — the streaming
    counterpart of ``BlockDataLoader`` / ``BlockDataset`` (BlockDatasetLoader.py:32-63, 77-99: load a block of
    ``block_size`` rows, shuffle inside it, cut it into minibatches, go to the next block).

    * ``source``: ``HDFSource`` (row-range reads through libhdf5 hyperslabs) or ``ArraySource``.
    * An epoch visits the blocks in a seeded random order (``shuffle``) that is identical on every rank.  Of each
      block a rank reads ONLY ITS OWN contiguous slice — rows ``[lo + s L, lo + (s + 1) L)`` with
      ``L = block_rows // world_size`` and ``s = (rank + epoch) % world_size``, so slices rotate over the epochs —
      in one sequential read, shuffles it, and cuts it into minibatches.  Every rank therefore yields the same
      number of minibatches of the same sizes per block (lock-step for the gradient all-reduce) without any
      communication, and reads 1 / world_size of the file.  With one rank a block is shuffled as a whole,
      exactly like the reference.
    * Double buffering: block k + 1's slice is read by a background thread (libhdf5 through ctypes releases the
      GIL) into the second pinned buffer while block k's minibatches are consumed; peak pinned host memory is
      2 slices + 2 staging minibatches, independent of the file size (``pinned_bytes``).
    * Staging of a minibatch: vectorised row gather (``np.take`` on whole rows: memcpy speed) into one of two
      pinned staging buffers, asynchronous H2D on a side stream one batch ahead of the consumer, graph_compact's
      counting phase for the batch on that stream (``ops.prefetch_compact``); int8 all the way into the model and
      the loss.
    * ``drop_last=False`` keeps a block's ragged last minibatch like the reference (same size on every rank);
      ``drop_zero_targets`` trims the TRAILING all-zero-target rows of the file (dataset-size padding,
      DataProcesser.py:268-269; NaN loss in the reference) — found once at construction, identically on all ranks.
    * With ``block_size >= rows`` and one rank the row order of an epoch is
      ``np.random.default_rng([seed, epoch]).permutation(rows)`` (what ``dp.ShardedBatchSampler`` draws).
    """

    def __init__(self, source, batch_size: int, block_size: int = 10000, rank: int = 0, world_size: int = 1,
                 seed: int = 0, shuffle: bool = True, device: Optional[str] = "cuda", drop_last: bool = False,
                 drop_zero_targets: bool = True, prefetch_compact: bool = True):
        if block_size < batch_size:
            raise ValueError("block size should be >= batch size (BlockDatasetLoader.py:83)")
        if not 0 <= rank < world_size:
            raise ValueError("rank out of range")
        self.src, self.batch_size, self.block_size = source, int(batch_size), int(block_size)
        self.rank, self.world, self.seed, self.shuffle, self.drop_last = rank, world_size, seed, shuffle, drop_last
        self.device = torch.device(device) if device is not None else None
        self.on_gpu = self.device is not None and self.device.type == "cuda"
        self.prefetch_compact = prefetch_compact and self.on_gpu
        self.epoch = 0
        self.n_rows = source.n_rows
        if drop_zero_targets and self.n_rows:
            self.n_rows = self._trim_trailing_zero_targets()
        self.n_blocks = (self.n_rows + self.block_size - 1) // self.block_size
        # per-rank slice length of a full block; the two slice buffers and the staging slots
        self._slice_cap = max(self.block_size // self.world, 1)
        pin = (lambda t: t.pin_memory()) if self.on_gpu else (lambda t: t)
        mk = lambda rows: tuple(pin(torch.empty((rows,) + tuple(shp), dtype=torch.int8)) for shp in source.row_shapes)
        self._slices = [mk(self._slice_cap), mk(self._slice_cap)]
        self._stage = [mk(self.batch_size), mk(self.batch_size)] if self.on_gpu else None
        self._stage_done = [None, None]
        self._stream = torch.cuda.Stream(self.device) if self.on_gpu else None
        self._reader = None                                   # background read of the next block's slice, if running
        self.pinned_bytes = sum(t.numel() for grp in self._slices + (self._stage or []) for t in grp)

    # ---- epoch plan (host arithmetic only; identical on every rank) -------------------------------------
    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _block_rows(self, b: int) -> int:
        return min(self.block_size, self.n_rows - b * self.block_size)

    def _slice_of(self, b: int):
        """(first row, rows) of this rank's slice of block b in the current epoch."""
        nb = self._block_rows(b)
        L = nb // self.world
        s = (self.rank + self.epoch) % self.world
        return b * self.block_size + s * L, L

    def _batches_in(self, L: int) -> int:
        return L // self.batch_size if self.drop_last else (L + self.batch_size - 1) // self.batch_size

    def __len__(self) -> int:
        return sum(self._batches_in(self._block_rows(b) // self.world) for b in range(self.n_blocks))

    def _block_order(self):
        order = np.arange(self.n_blocks)
        if self.shuffle and self.n_blocks > 1:
            order = np.random.default_rng([self.seed, self.epoch, 1 << 20]).permutation(self.n_blocks)
        return order

    def _row_order(self, b: int, L: int):
        if not self.shuffle:
            return np.arange(L)
        key = [self.seed, self.epoch] if self.n_blocks == 1 else [self.seed, self.epoch, int(b)]
        if self.world == 1:
            return np.random.default_rng(key).permutation(L)
        return np.random.default_rng(key + [self.rank]).permutation(L)

    def _trim_trailing_zero_targets(self) -> int:
        n = self.src.n_rows
        chunk = min(n, max(self.block_size, 1024))
        bufs = tuple(np.empty((chunk,) + tuple(shp), dtype=np.int8) for shp in self.src.row_shapes)
        hi = n
        while hi > 0:
            lo = max(hi - chunk, 0)
            self.src.read_rows(lo, hi, bufs)
            live = bufs[2][:hi - lo].reshape(hi - lo, -1).any(1)
            nz = np.nonzero(live)[0]
            if nz.size:
                return lo + int(nz[-1]) + 1
            hi = lo
        return 0

    # ---- staging ----------------------------------------------------------------------------------------
    def _emit(self, slice_np, idx: np.ndarray, slot: int):
        if not self.on_gpu:
            return tuple(torch.from_numpy(np.take(a, idx, axis=0)) for a in slice_np), None
        if self._stage_done[slot] is not None:
            self._stage_done[slot].synchronize()              # staging slot free again (its H2D finished)
        k = idx.shape[0]
        stage = self._stage[slot]
        for src, dst in zip(slice_np, stage):
            d = dst.numpy().reshape(dst.shape[0], -1)
            np.take(src.reshape(src.shape[0], -1), idx, axis=0, out=d[:k])
        with torch.cuda.stream(self._stream):
            dev = tuple(s[:k].to(self.device, non_blocking=True) for s in stage)
            if self.prefetch_compact:
                from . import ops
                ops.prefetch_compact(dev[0], dev[1], stream=self._stream)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._stage_done[slot] = ev
        return dev, ev

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        order = [int(b) for b in self._block_order()]
        order = [b for b in order if self._batches_in(self._block_rows(b) // self.world) > 0]
        if not order:
            return
        if self._reader is not None:                          # an abandoned iteration's reader still writes into a
            self._reader.join()                               # slice buffer this one is about to reuse
            self._reader = None
        err = []

        def read(b: int, which: int):
            try:
                lo, L = self._slice_of(b)
                self.src.read_rows(lo, lo + L, tuple(t.numpy() for t in self._slices[which]))
            except Exception as e:                            # surfaced in the consumer thread
                err.append(e)

        cur = torch.cuda.current_stream(self.device) if self.on_gpu else None
        which, slot = 0, 0
        read(order[0], which)
        pending = None                                        # (dev tensors, event) staged one batch ahead
        for k, b in enumerate(order):
            if err:
                raise err[0]
            reader = None
            if k + 1 < len(order):                            # next block's slice in the background
                reader = threading.Thread(target=read, args=(order[k + 1], which ^ 1), daemon=True)
                self._reader = reader
                reader.start()
            _, L = self._slice_of(b)
            slice_np = [t.numpy()[:L] for t in self._slices[which]]
            rows = self._row_order(b, L)
            nb = self._batches_in(L)
            for j in range(nb):
                idx = rows[j * self.batch_size:(j + 1) * self.batch_size]
                nxt = self._emit(slice_np, idx, slot)
                slot ^= 1
                if pending is not None:
                    yield self._hand_over(pending, cur)
                pending = nxt
            # the last batch of this block is still staged from this slice buffer: it has been gathered
            # (np.take is synchronous), so the buffer may be overwritten by the read after next
            if reader is not None:
                reader.join()
                self._reader = None
            which ^= 1
        if pending is not None:
            yield self._hand_over(pending, cur)
        if err:
            raise err[0]

    def _hand_over(self, pending, cur):
        dev, ev = pending
        if ev is not None:
            cur.wait_event(ev)                                # consumer stream waits for ITS batch only
            for t in dev:
                t.record_stream(cur)
        return dev
