"""CPU: the int8 sharded block loader (host-side logic; device copies are covered by -m gpu)."""
import os

import numpy as np
import pytest
import torch

from graphinvent_amd.loader import ShardedBlockLoader, read_hdf_int8

REF_H5 = "/root/reference/data/pre-training/gdb13_1K-debug/train.h5"


def _fixture(golden_dir, split="train"):
    d = np.load(os.path.join(golden_dir, f"gdb13_1K-debug_{split}.npz"))
    return d["nodes"], d["edges"], d["APDs"]


def test_drops_padding_rows_and_shards_disjointly(golden_dir):
    n, e, a = _fixture(golden_dir)
    loaders = [ShardedBlockLoader(n, e, a, 16, rank=r, world_size=2, seed=3, device=None)
               for r in range(2)]
    assert loaders[0].block[0].shape[0] == 129                 # 21 all-zero padding rows dropped
    assert len(loaders[0]) == len(loaders[1]) == (129 // 16) // 2
    seen = []
    for ld in loaders:
        for nodes, edges, apds in ld:
            assert nodes.dtype == edges.dtype == apds.dtype == torch.int8
            assert nodes.shape == (16, 13, 8) and edges.shape == (16, 13, 13, 3) and apds.shape == (16, 625)
            assert int(apds.ne(0).any(1).sum()) == 16
            seen.append(apds.numpy().tobytes())
    assert len(seen) == 8
    keep = ShardedBlockLoader(n, e, a, 16, device=None, drop_zero_targets=False)
    assert keep.block[0].shape[0] == 150


def test_epochs_reshuffle_but_ranks_agree_on_the_permutation(golden_dir):
    n, e, a = _fixture(golden_dir, "valid")
    a0 = ShardedBlockLoader(n, e, a, 10, rank=0, world_size=2, seed=1, device=None)
    b0 = ShardedBlockLoader(n, e, a, 10, rank=0, world_size=2, seed=1, device=None)
    first = [x[2].clone() for x in a0]
    assert all(torch.equal(p, q[2]) for p, q in zip(first, b0))       # deterministic
    a0.set_epoch(1)
    assert not all(torch.equal(p, q[2]) for p, q in zip(first, a0))    # reshuffled


def test_rejects_non_int8():
    with pytest.raises(TypeError):
        ShardedBlockLoader(np.zeros((4, 2, 2), np.float32), np.zeros((4, 2, 2, 1), np.int8),
                           np.zeros((4, 3), np.int8), 2, device=None)


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
    assert len(dl) == n.shape[0] // 16
    first = [b[2].clone() for b in dl]
    assert len(first) == len(dl) and all(b.dtype == torch.int8 and b.shape == (16, 625) for b in first)
    second = [b[2].clone() for b in dl]                                        # next epoch: reshuffled
    assert not all(torch.equal(p, q) for p, q in zip(first, second))


@pytest.mark.skipif(not (os.path.exists(REF_H5) and os.path.exists("/opt/conda/lib/libhdf5.so")),
                    reason="reference HDF fixture / libhdf5 not on this box")
def test_dropin_hdfdataset_reads_the_reference_file(golden_dir):
    from graphinvent_amd.BlockDatasetLoader import HDFDataset
    ds = HDFDataset(REF_H5)
    fn, fe, fa = _fixture(golden_dir)
    assert len(ds) == fn.shape[0] and np.array_equal(ds.apds, fa)
