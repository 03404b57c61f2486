"""CPU: the int8 sharded block loader (host-side logic; device copies are covered by -m gpu)."""
import os

import numpy as np
import pytest
import torch

from graphinvent_amd.loader import ArraySource, BlockStreamLoader, HDFSource, LazyRows, read_hdf_int8

REF_H5 = "/root/reference/data/pre-training/gdb13_1K-debug/train.h5"


def _fixture(golden_dir, split="train"):
    d = np.load(os.path.join(golden_dir, f"gdb13_1K-debug_{split}.npz"))
    return d["nodes"], d["edges"], d["APDs"]


def test_drops_padding_rows_and_shards_disjointly(golden_dir):
    n, e, a = _fixture(golden_dir)
    loaders = [BlockStreamLoader(ArraySource(n, e, a), 16, block_size=10000, rank=r, world_size=2, seed=3,
                                 device=None, drop_last=True) for r in range(2)]
    assert loaders[0].n_rows == 129                            # 21 all-zero padding rows dropped
    assert len(loaders[0]) == len(loaders[1]) == (129 // 2) // 16
    seen = []
    for ld in loaders:
        for nodes, edges, apds in ld:
            assert nodes.dtype == edges.dtype == apds.dtype == torch.int8
            assert nodes.shape == (16, 13, 8) and edges.shape == (16, 13, 13, 3) and apds.shape == (16, 625)
            assert int(apds.ne(0).any(1).sum()) == 16
            seen.extend(r.numpy().tobytes() for r in apds)
    assert len(seen) == 2 * 4 * 16
    keep = BlockStreamLoader(ArraySource(n, e, a), 16, device=None, drop_zero_targets=False)
    assert keep.n_rows == 150


def test_epochs_reshuffle_but_ranks_agree_on_the_permutation(golden_dir):
    n, e, a = _fixture(golden_dir, "valid")
    mk = lambda: BlockStreamLoader(ArraySource(n, e, a), 10, rank=0, world_size=2, seed=1, device=None)
    a0, b0 = mk(), mk()
    first = [x[2].clone() for x in a0]
    assert all(torch.equal(p, q[2]) for p, q in zip(first, b0))       # deterministic
    a0.set_epoch(1)
    assert not all(torch.equal(p, q[2]) for p, q in zip(first, a0))    # reshuffled


def test_rejects_non_int8():
    with pytest.raises(TypeError):
        ArraySource(np.zeros((4, 2, 2), np.float32), np.zeros((4, 2, 2, 1), np.int8), np.zeros((4, 3), np.int8))


@pytest.mark.skipif(not (os.path.exists(REF_H5) and os.path.exists("/opt/conda/lib/libhdf5.so")),
                    reason="reference HDF fixture / libhdf5 not on this box")
def test_hdf_reader_matches_committed_fixture(golden_dir):
    n, e, a = read_hdf_int8(REF_H5)
    fn, fe, fa = _fixture(golden_dir)
    assert np.array_equal(n, fn) and np.array_equal(e, fe) and np.array_equal(a, fa)


def test_dropin_blockdatasetloader_module_api(golden_dir):
    """`BlockDatasetLoader` drop-in: the reference's class names and constructor arguments
    (BlockDatasetLoader.py:17-31, 122-147; Workflow.get_dataloader, Workflow.py:131-137), importable as
    the top-level module when graphinvent_amd/ is ahead on sys.path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import BlockDatasetLoader as B; "
            "print(B.__file__); print(B.HDFDataset.__name__, B.BlockDataLoader.__name__)"
            % os.path.join(root, "graphinvent_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/")
    assert out.returncode == 0, out.stderr
    assert "graphinvent_amd/BlockDatasetLoader.py" in out.stdout and "HDFDataset BlockDataLoader" in out.stdout

    from graphinvent_amd.BlockDatasetLoader import BlockDataLoader, HDFDataset
    n, e, a = _fixture(golden_dir, "valid")
    ds = HDFDataset.from_arrays(n, e, a)
    assert len(ds) == n.shape[0]
    x = ds[3]
    assert all(t.dtype == torch.float32 for t in x) and x[0].shape == (13, 8)        # reference dtype
    assert all(t.shape[0] == 5 for t in ds[10:15])
    dl = BlockDataLoader(dataset=ds, batch_size=16, block_size=10000, shuffle=True, n_workers=0,
                         pin_memory=True)
    assert len(dl) == -(-n.shape[0] // 16)                    # the reference's ceil: ragged last batch kept
    first = [b[2].clone() for b in dl]
    assert len(first) == len(dl) and all(b.dtype == torch.int8 and b.shape[1] == 625 for b in first)
    assert sorted(b.shape[0] for b in first)[1:] == [16] * (len(dl) - 1) and sum(b.shape[0] for b in first) == 100
    # a file smaller than one minibatch with drop_last: the reference's validation_epoch would average an empty
    # list (NaN) — refused loudly instead (advisor finding, round 2)
    small = HDFDataset.from_arrays(n[:10], e[:10], a[:10])
    assert len(BlockDataLoader(dataset=small, batch_size=16)) == 1
    with pytest.raises(ValueError):
        BlockDataLoader(dataset=small, batch_size=16, drop_last=True)
    second = [b[2].clone() for b in dl]                                        # next epoch: reshuffled
    assert not all(torch.equal(p, q) for p, q in zip(first, second))


@pytest.mark.skipif(not (os.path.exists(REF_H5) and os.path.exists("/opt/conda/lib/libhdf5.so")),
                    reason="reference HDF fixture / libhdf5 not on this box")
def test_dropin_hdfdataset_reads_the_reference_file(golden_dir):
    from graphinvent_amd.BlockDatasetLoader import HDFDataset
    ds = HDFDataset(REF_H5)
    fn, fe, fa = _fixture(golden_dir)
    assert len(ds) == fn.shape[0] and np.array_equal(ds.apds, fa)
    assert torch.equal(ds[7][1], torch.from_numpy(fe[7]).float())
    assert torch.equal(ds[20:33][0], torch.from_numpy(fn[20:33]).float())


# ---- block-wise streaming (BlockDatasetLoader.py:32-63, 77-99) ------------------------------------------------
def test_block_stream_row_order_when_one_block_holds_the_file(golden_dir):
    n, e, a = _fixture(golden_dir)
    ld = BlockStreamLoader(ArraySource(n, e, a), 16, block_size=10000, device=None, drop_last=True, seed=3)
    assert ld.n_rows == 129 and len(ld) == 8                        # trailing padding rows trimmed
    for epoch in (0, 1):
        ld.set_epoch(epoch)
        order = np.random.default_rng([3, epoch]).permutation(129)
        got = list(ld)
        assert len(got) == 8
        for k, (bn, be, ba) in enumerate(got):                      # bit-identical rows in the seeded order
            idx = order[16 * k:16 * (k + 1)]
            assert np.array_equal(bn.numpy(), n[idx]) and np.array_equal(be.numpy(), e[idx]) \
                and np.array_equal(ba.numpy(), a[idx])


def test_lazy_dataset_views_read_only_what_is_asked():
    """HDFDataset.nodes / .edges / .apds (BlockDatasetLoader.py:128-130 are lazy h5py datasets): shape and len cost
    nothing, an index reads those rows of that dataset only (round-3 advisor finding)."""
    from graphinvent_amd.BlockDatasetLoader import HDFDataset
    rng = np.random.default_rng(0)
    n = rng.integers(0, 2, (50, 3, 2)).astype(np.int8); e = rng.integers(0, 2, (50, 3, 3, 1)).astype(np.int8)
    a = rng.integers(0, 5, (50, 7)).astype(np.int8)
    base = ArraySource(n, e, a)
    calls = []

    class Spy:
        n_rows, row_shapes = base.n_rows, base.row_shapes
        def read_rows(self, lo, hi, outs, which=(0, 1, 2)):
            calls.append((lo, hi, tuple(which)))
            base.read_rows(lo, hi, outs, which=which)
    ds = HDFDataset.from_arrays(n, e, a)
    ds.source = Spy()
    assert isinstance(ds.nodes, LazyRows) and ds.nodes.shape == (50, 3, 2) and len(ds.apds) == 50 and not calls
    assert np.array_equal(ds.edges[7:19], e[7:19]) and calls == [(7, 19, (1,))]
    assert np.array_equal(ds.apds[-1], a[49]) and calls[-1] == (49, 50, (2,))
    assert np.array_equal(np.asarray(ds.nodes), n) and calls[-1] == (0, 50, (0,))
    with pytest.raises(IndexError):
        ds.nodes[0:10:2]


def test_one_lock_per_libhdf5_handle_and_abandoned_reader_is_joined():
    """Two sources of the same library share ONE lock (libhdf5 keeps global state); a new iteration joins the
    background reader an abandoned one left running before it reuses the slice buffers."""
    import threading, time
    from graphinvent_amd import loader as L
    if os.path.exists("/opt/conda/lib/libhdf5.so") and os.path.exists(REF_H5):
        s1, s2 = HDFSource(REF_H5), HDFSource(REF_H5.replace("train.h5", "valid.h5"))
        assert s1._lock is s2._lock
        s1.close(); s2.close()
        with pytest.raises(ValueError):
            s1.read_rows(0, 1, tuple(np.empty((1,) + tuple(shp), np.int8) for shp in s1.row_shapes))
    gate = threading.Event()

    class Slow:
        n_rows, row_shapes = 64, [(2,), (2,), (3,)]
        active = 0
        def read_rows(self, lo, hi, outs, which=(0, 1, 2)):
            Slow.active += 1
            assert Slow.active == 1, "two reads into the slice buffers at once"
            if lo >= 16:
                gate.wait(0.5)
            for o in outs:
                o[:hi - lo] = 1
            Slow.active -= 1
    ld = BlockStreamLoader(Slow(), 4, block_size=16, device=None, shuffle=False, drop_zero_targets=False)
    it = iter(ld)
    next(it)                                                            # block 0 consumed, reader of block 1 running
    assert ld._reader is not None and ld._reader.is_alive()
    t0 = time.time()
    assert sum(1 for _ in ld) == 16                                     # new iteration: joins it first
    gate.set()


def test_block_stream_covers_every_row_once_blockwise_and_keeps_ragged_batches():
    n = np.zeros((100, 3, 2), np.int8); e = np.zeros((100, 3, 3, 1), np.int8); a = np.zeros((100, 7), np.int8)
    a[:, 0] = np.arange(100) % 100 + 1                                  # row id in the target (never all-zero)
    n[:, 0, 0] = a[:, 0]
    ld = BlockStreamLoader(ArraySource(n, e, a), 16, block_size=40, device=None, seed=5)
    assert ld.n_blocks == 3 and len(ld) == 3 + 3 + 2
    batches = list(ld)
    assert [b[2].shape[0] for b in batches].count(16) == 5 and sum(b[2].shape[0] for b in batches) == 100
    assert all(torch.equal(b[0][:, 0, 0], b[2][:, 0]) for b in batches)     # the three arrays stay aligned
    # block-wise: the first 3 minibatches are one block's rows (a contiguous range of 40 file rows), shuffled
    rows = [[int(v) - 1 for v in b[2][:, 0]] for b in batches]
    first_block = sorted(sum(rows[:3], []))
    assert first_block == list(range(first_block[0], first_block[0] + 40)) and first_block[0] % 40 == 0
    assert sum(rows[:3], []) != first_block                            # shuffled inside the block
    assert sorted(sum(rows, [])) == list(range(100))
    ld.set_epoch(1)
    again = [[int(v) - 1 for v in b[2][:, 0]] for b in ld]
    assert sorted(sum(again, [])) == list(range(100)) and again != rows


def test_block_stream_ranks_read_disjoint_slices_in_lock_step(golden_dir):
    n, e, a = _fixture(golden_dir)
    src = ArraySource(n, e, a)

    class Counting:                                                     # which file rows does a rank touch?
        def __init__(self): self.n_rows, self.row_shapes, self.seen = src.n_rows, src.row_shapes, []
        def read_rows(self, lo, hi, outs, which=(0, 1, 2)): self.seen.append((lo, hi)); src.read_rows(lo, hi, outs)
    srcs = [Counting(), Counting()]
    lds = [BlockStreamLoader(srcs[r], 16, block_size=50, rank=r, world_size=2, device=None, seed=1) for r in range(2)]
    for epoch in (0, 1):
        for ld, c in zip(lds, srcs):
            ld.set_epoch(epoch); c.seen.clear()
        b0, b1 = list(lds[0]), list(lds[1])
        assert [x[2].shape[0] for x in b0] == [x[2].shape[0] for x in b1] == [16, 9, 16, 9, 14] or \
            sorted(x[2].shape[0] for x in b0) == sorted(x[2].shape[0] for x in b1)
        assert len(b0) == len(b1) == len(lds[0]) == 5
        reads = [sorted(r for r in c.seen if r[1] - r[0] < 129) for c in srcs]      # (the trim scan reads more)
        rows0 = set(i for lo, hi in reads[0] for i in range(lo, hi))
        rows1 = set(i for lo, hi in reads[1] for i in range(lo, hi))
        assert not rows0 & rows1 and len(rows0) == len(rows1) == 25 + 25 + 14
        assert all(hi - lo <= 25 for lo, hi in reads[0])               # a slice per block, never the block
    assert torch.cat([x[2] for x in b0 + b1]).ne(0).any(1).all()       # no padding row reaches a rank


def test_block_stream_pinned_memory_is_two_slices_whatever_the_file_size():
    """10 M rows (12 GB as a GDB-13 file) through a virtual source: nothing but two block slices and two staging
    minibatches is ever allocated, and an epoch's plan is host arithmetic."""
    class Virtual:
        n_rows = 10_000_000
        row_shapes = [(13, 8), (13, 13, 3), (625,)]
        reads = 0
        def read_rows(self, lo, hi, outs, which=(0, 1, 2)):
            Virtual.reads += 1
            for o in outs:
                o[:hi - lo].reshape(hi - lo, -1)[:] = (np.arange(lo, hi) % 127 + 1).astype(np.int8)[:, None]
    ld = BlockStreamLoader(Virtual(), 1000, block_size=10000, device=None, seed=0, drop_zero_targets=False)
    row_bytes = 13 * 8 + 13 * 13 * 3 + 625
    assert ld.pinned_bytes == 2 * 10000 * row_bytes and len(ld) == 10_000
    it = iter(ld)
    got = [next(it) for _ in range(25)]                                 # crosses two block boundaries
    assert all(b[0].shape == (1000, 13, 8) for b in got) and Virtual.reads <= 4
    for blk in (got[:10], got[10:20]):                                  # each block = one contiguous 10 000-row range
        vals = np.concatenate([b[2][:, 0].numpy() for b in blk]).astype(np.int64)
        assert len(vals) == 10000
    ld8 = BlockStreamLoader(Virtual(), 1000, block_size=10000, rank=3, world_size=8, device=None,
                            drop_zero_targets=False)
    assert ld8.pinned_bytes == 2 * 1250 * row_bytes and len(ld8) == 2000


@pytest.mark.skipif(not (os.path.exists(REF_H5) and os.path.exists("/opt/conda/lib/libhdf5.so")),
                    reason="reference HDF fixture / libhdf5 not on this box")
def test_hdf_source_reads_row_ranges(golden_dir):
    fn, fe, fa = _fixture(golden_dir)
    src = HDFSource(REF_H5)
    assert src.n_rows == 150 and [tuple(s) for s in src.row_shapes] == [(13, 8), (13, 13, 3), (625,)]
    outs = tuple(np.zeros((40,) + tuple(s), dtype=np.int8) for s in src.row_shapes)
    src.read_rows(97, 131, outs)
    assert all(np.array_equal(o[:34], f[97:131]) for o, f in zip(outs, (fn, fe, fa)))
    with pytest.raises(IndexError):
        src.read_rows(140, 151, outs)
    ld = BlockStreamLoader(src, 16, block_size=64, device=None, seed=2)
    assert ld.n_rows == 129 and sum(b[2].shape[0] for b in ld) == 129


def test_hdf_roundtrip_through_libhdf5_without_the_reference_checkout(tmp_path):
    """tests/h5util.write_h5 creates the three int8 datasets the way DataProcesser.py:157-161 does; HDFSource and the
    drop-in HDFDataset read them back bit-exactly (runs wherever a libhdf5 is, e.g. on the GPU box)."""
    from graphinvent_amd.BlockDatasetLoader import HDFDataset
    from tests.h5util import have_libhdf5, write_h5
    if not have_libhdf5():
        pytest.skip("no libhdf5 on this box")
    rng = np.random.default_rng(1)
    n = rng.integers(0, 2, (77, 13, 8)).astype(np.int8); e = rng.integers(0, 2, (77, 13, 13, 3)).astype(np.int8)
    a = rng.integers(0, 9, (77, 625)).astype(np.int8)
    path = str(tmp_path / "x.h5")
    write_h5(path, n, e, a)
    rn, re_, ra = read_hdf_int8(path)
    assert np.array_equal(rn, n) and np.array_equal(re_, e) and np.array_equal(ra, a)
    ds = HDFDataset(path)
    assert np.array_equal(ds.edges[70:77], e[70:77]) and torch.equal(ds[3][0], torch.from_numpy(n[3]).float())
    got = np.concatenate([b[2].numpy() for b in BlockStreamLoader(ds.source, 10, block_size=30, device=None, seed=0)])
    assert sorted(map(bytes, got)) == sorted(map(bytes, a))


def test_eight_ranks_stay_in_lock_step_and_read_disjoint_eighths():
    """The world size of the scaling run: every rank yields the same minibatch sizes per block without talking to
    the others, the eight slices of a block are disjoint, and together they cover it up to the rows % 8 tail."""
    rows = 5003
    n = np.zeros((rows, 2, 2), np.int8); e = np.zeros((rows, 2, 2, 1), np.int8); a = np.ones((rows, 5), np.int8)
    ids = np.arange(rows)
    a[:, 0] = ids % 100 + 1; a[:, 1] = (ids // 100) % 100 + 1; a[:, 2] = ids // 10000 + 1
    lds = [BlockStreamLoader(ArraySource(n, e, a), 50, block_size=2000, rank=r, world_size=8, device=None, seed=7)
           for r in range(8)]
    for epoch in (0, 1, 2):
        per_rank = []
        for ld in lds:
            ld.set_epoch(epoch)
            bs = list(ld)
            got = np.concatenate([(b[2][:, 0].numpy().astype(np.int64) - 1) + 100 * (b[2][:, 1].numpy().astype(np.int64) - 1)
                                  + 10000 * (b[2][:, 2].numpy().astype(np.int64) - 1) for b in bs])
            per_rank.append(([b[2].shape[0] for b in bs], got))
        assert all(p[0] == per_rank[0][0] for p in per_rank) and len(per_rank[0][0]) == len(lds[0])
        allrows = np.concatenate([p[1] for p in per_rank])
        assert len(allrows) == len(set(allrows.tolist())) == 8 * (2000 // 8) * 2 + 8 * (1003 // 8)
