"""-m gpu: the input pipeline as shipped — pinned block slices -> vectorised row gather ->
asynchronous H2D on a side stream one batch ahead -> graph_compact's counting phase on that stream — delivers
BIT-EXACT int8 rows (the contract of BlockDatasetLoader.py:32-63, 135-143 is "these rows of the file", integer
work: no tolerance), every row exactly once per epoch, and the prefetched compaction changes no bit.

Every row of the test file carries its own row number (two int8 columns of the APD target), so a delivered row says
which source row it claims to be and is compared with that row in full."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graphinvent_amd import ops, synthetic
from graphinvent_amd.loader import ArraySource, BlockStreamLoader, HDFSource
from tests.h5util import have_libhdf5, write_h5

pytestmark = pytest.mark.gpu
ROWS = 2347                                                     # not a multiple of anything below


def tagged_rows(rows=ROWS, seed=5):
    n8, e8, a8 = synthetic.make_batch(rows, **synthetic.SHAPES["gdb13"], seed=seed)
    a8 = a8.copy()
    ids = np.arange(rows)
    a8[:, 0] = (ids % 100) + 1                                  # (never zero: no row looks like padding)
    a8[:, 1] = (ids // 100) + 1
    return n8, e8, a8


def ids_of(apds: torch.Tensor) -> np.ndarray:
    a = apds.cpu().numpy().astype(np.int64)
    return (a[:, 0] - 1) + 100 * (a[:, 1] - 1)


def check_epoch(batches, src_arrays, expect_rows):
    """every delivered row equals the source row it claims to be; `expect_rows` ids, each exactly once"""
    n8, e8, a8 = src_arrays
    seen = []
    for nodes, edges, apds in batches:
        assert nodes.is_cuda and nodes.dtype == edges.dtype == apds.dtype == torch.int8
        idx = ids_of(apds)
        assert torch.equal(nodes.cpu(), torch.from_numpy(n8[idx]))
        assert torch.equal(edges.cpu(), torch.from_numpy(e8[idx]))
        assert torch.equal(apds.cpu(), torch.from_numpy(a8[idx]))
        seen.append(idx)
    seen = np.concatenate(seen)
    assert len(seen) == len(set(seen.tolist())), "a row was delivered twice"
    if expect_rows is not None:
        assert sorted(seen.tolist()) == sorted(expect_rows)
    return seen


@pytest.mark.parametrize("kind", ["array", "hdf"])
def test_block_stream_loader_delivers_bit_exact_rows_once_per_epoch(tmp_path, kind):
    arrays = tagged_rows()
    if kind == "hdf":
        if not have_libhdf5():
            pytest.skip("no libhdf5 on this box")
        path = str(tmp_path / "train.h5")
        write_h5(path, *arrays)
        src = HDFSource(path)
        assert src.n_rows == ROWS
    else:
        src = ArraySource(*arrays)
    ld = BlockStreamLoader(src, 256, block_size=700, device="cuda", seed=11)       # 4 blocks, ragged tails kept
    assert ld.n_blocks == 4 and len(ld) == 3 + 3 + 3 + 1
    orders = []
    for epoch in (0, 1):
        ld.set_epoch(epoch)
        got = list(ld)
        assert len(got) == len(ld)
        orders.append(check_epoch(got, arrays, range(ROWS)))
    assert not np.array_equal(orders[0], orders[1])                                 # reshuffled
    # abandoned iteration, then a full one on the same buffers
    it = iter(ld)
    next(it); next(it)
    check_epoch(list(ld), arrays, range(ROWS))
    torch.cuda.synchronize()


def test_dropin_blockdataloader_delivers_bit_exact_rows(tmp_path):
    """graphinvent_amd/BlockDatasetLoader.py under the reference's class names and constructor arguments
    (Workflow.py:131-137): same guarantee, and what HDFDataset.__getitem__ hands out is the reference's fp32 row."""
    from graphinvent_amd.BlockDatasetLoader import BlockDataLoader, HDFDataset
    arrays = tagged_rows(rows=1000, seed=6)
    if have_libhdf5():
        path = str(tmp_path / "valid.h5")
        write_h5(path, *arrays)
        ds = HDFDataset(path)
    else:
        ds = HDFDataset.from_arrays(*arrays)
    assert len(ds) == 1000 and ds.nodes.shape == (1000, 13, 8)
    assert torch.equal(ds[17][2], torch.from_numpy(arrays[2][17]).float())
    dl = BlockDataLoader(dataset=ds, batch_size=128, block_size=400, shuffle=True, n_workers=0, pin_memory=True)
    assert len(dl) == 4 + 4 + 2
    e0 = check_epoch(list(dl), arrays, range(1000))
    e1 = check_epoch(list(dl), arrays, range(1000))                                 # next pass: new permutation
    assert not np.array_equal(e0, e1)


def test_prefetched_compaction_on_the_loader_is_bit_identical_to_the_unprefetched_one():
    """The same loader with and without graph_compact's counting phase run one batch ahead on the copy stream: the
    model's outputs must not differ in one bit, and the prefetched run must not block on a read-back."""
    from graphinvent_amd.gnn import mpnn
    from oracle import ggnn_oracle as O
    arrays = tagged_rows(rows=640, seed=8)
    cfg = O.make_config(device="cuda")
    model = mpnn.GGNN(O.as_constants(cfg))
    model.load_state_dict(O.init_params(cfg, seed=2))
    model = model.to("cuda").eval()
    outs = {}
    for prefetch in (True, False):
        ld = BlockStreamLoader(ArraySource(*arrays), 128, block_size=320, device="cuda", seed=4,
                               prefetch_compact=prefetch)
        ops._PREFETCHED.clear()
        before = dict(ops.READBACKS)
        res = []
        with torch.no_grad():
            for nodes, edges, apds in ld:
                res.append((ids_of(apds), model(nodes, edges).clone()))
        delta = {k: ops.READBACKS[k] - before[k] for k in before}
        assert delta == ({"prefetched": 6, "blocking": 0, "bounded": 0} if prefetch else
                         {"prefetched": 0, "blocking": 6, "bounded": 0}), delta
        outs[prefetch] = res
    for (ia, a), (ib, b) in zip(outs[True], outs[False]):
        assert np.array_equal(ia, ib) and torch.equal(a, b)
    # and the int8 rows give the logits of the fp32 rows the reference's loader would have built (:139-141)
    nodes, edges = (torch.from_numpy(arrays[k][outs[True][0][0]]).float().cuda() for k in (0, 1))
    with torch.no_grad():
        assert torch.equal(model(nodes, edges), outs[True][0][1])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_worker(rank, world, port, out_dir, h5_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                                    # 1-GPU box: the ranks share the device (gloo)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphinvent_amd.BlockDatasetLoader import BlockDataLoader, HDFDataset
    arrays = tagged_rows()
    ds = HDFDataset(h5_path) if h5_path else HDFDataset.from_arrays(*arrays)
    dl = BlockDataLoader(dataset=ds, batch_size=128, block_size=700)                # picks rank / world up from dist
    epochs = []
    for _ in range(2):
        sizes, ids = [], []
        for nodes, edges, apds in dl:
            idx = ids_of(apds)
            ok = (torch.equal(nodes.cpu(), torch.from_numpy(arrays[0][idx]))
                  and torch.equal(edges.cpu(), torch.from_numpy(arrays[1][idx]))
                  and torch.equal(apds.cpu(), torch.from_numpy(arrays[2][idx])))
            assert ok, "a delivered row differs from the source row it claims to be"
            sizes.append(len(idx)); ids.extend(idx.tolist())
        epochs.append(dict(sizes=sizes, ids=ids))
    torch.save(dict(epochs=epochs, n=len(dl)), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_get_disjoint_bit_exact_shards_in_lock_step(tmp_path):
    h5 = ""
    if have_libhdf5():
        h5 = str(tmp_path / "train.h5")
        write_h5(h5, *tagged_rows())
    mp.spawn(_rank_worker, args=(2, _free_port(), str(tmp_path), h5), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in (0, 1))
    assert r0["n"] == r1["n"]
    for ep in (0, 1):
        a, b = r0["epochs"][ep], r1["epochs"][ep]
        assert a["sizes"] == b["sizes"] and len(a["sizes"]) == r0["n"]              # lock-step
        assert not set(a["ids"]) & set(b["ids"])                                    # disjoint
        assert len(set(a["ids"])) == len(a["ids"])
        # every block's trailing (rows % world) row is skipped in that epoch, nothing else
        blocks = [min(700, ROWS - s) for s in range(0, ROWS, 700)]
        assert len(a["ids"]) + len(b["ids"]) == sum(2 * (n // 2) for n in blocks)
    assert set(r0["epochs"][0]["ids"]) != set(r0["epochs"][1]["ids"])               # slices rotate over epochs
