"""-m gpu: the pass-0 row cache of inference loops (gi_graph.p0_cache, gi_p0_cache_words).  In the first message
pass h = [x | 0] (gnn/summation_mpnn.py:121-126), so a message row depends only on (bond type, feature pattern of
the source node) and the weights; GraphGenerator.build_graphs (GraphGenerator.py:118-157) calls the model thousands
of times between weight updates.  The cache must never change a logit beyond the last bit and must notice every
way this package changes weights."""
import numpy as np
import pytest
import torch

from graphinvent_amd import synthetic
from graphinvent_amd.gnn import mpnn
from graphinvent_amd.optim import FusedAdam
from oracle import ggnn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(kind, shape, seed=3, **over):
    sh = synthetic.SHAPES[shape]
    cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], **over)
    P = O.init_params(cfg, seed=seed, model=kind)
    cls = mpnn.AttentionGGNN if kind == "AttGGNN" else mpnn.GGNN
    m = cls(O.as_constants(dict(cfg, device="cuda")))
    m.load_state_dict(P)
    return m.to(DEV).eval(), cfg, sh


def _batch(sh, B, seed):
    n8, e8, _ = synthetic.make_batch(B, **sh, seed=seed)
    return torch.from_numpy(n8).to(DEV), torch.from_numpy(e8).to(DEV)


def _uncached(model, nodes, edges):
    model.cache_pass0 = False
    try:
        with torch.no_grad():
            return model(nodes, edges)
    finally:
        model.cache_pass0 = True


@pytest.mark.parametrize("kind,shape,B", [("GGNN", "gdb13", 1000), ("AttGGNN", "gdb13", 300), ("GGNN", "zinc", 200)])
@pytest.mark.parametrize("sync_free", [False, True])
def test_cached_forward_equals_the_uncached_forward(kind, shape, B, sync_free):
    model, cfg, sh = _model(kind, shape)
    model.sync_free = sync_free
    batches = [_batch(sh, B, seed) for seed in (5, 6, 7)]
    refs = [_uncached(model, *b) for b in batches]
    with torch.no_grad():
        first = model(*batches[0])                   # empty table: the stack runs, rows are added
        st = model.pass0_cache_stats()
        assert st["forwards"] == 1 and st["hits"] == 0 and 0 < st["rows"] <= 4096
        assert torch.equal(first, refs[0])
        again = model(*batches[0])                   # every row is in the table: the stack launch exits at once
        st2 = model.pass0_cache_stats()
        assert st2["hits"] == 1 and st2["rows"] == st["rows"]
        assert torch.equal(again, refs[0])
        for b, ref in zip(batches[1:] + batches, refs[1:] + refs):      # other batches: hits or partial misses
            out = model(*b)
            # a row's arithmetic does not depend on the batch it was computed in (same MFMA k order)
            assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        st3 = model.pass0_cache_stats()
        assert st3["forwards"] == 7 and st3["hits"] >= 4       # the second visit of each batch is a hit
    if sync_free:
        assert model.last_bounded_error() == 0


def test_every_weight_update_empties_the_table():
    model, cfg, sh = _model("GGNN", "gdb13")
    nodes, edges = _batch(sh, 400, 11)
    with torch.no_grad():
        model(nodes, edges); model(nodes, edges)
    assert model.pass0_cache_stats()["hits"] == 1

    def check(tag):
        ref = _uncached(model, nodes, edges)
        with torch.no_grad():
            out = model(nodes, edges)
        assert model.pass0_cache_stats() == {"forwards": 1, "hits": 0, "rows": model.pass0_cache_stats()["rows"]}, tag
        assert torch.equal(out, ref), tag

    # (1) an in-place update that bumps the version counter (torch optimizers, load_state_dict, copy_)
    with torch.no_grad():
        getattr(model.msg_nns[0].seq, "0").weight.mul_(1.5)
    check("in-place op")
    # (2) FusedAdam writes through raw pointers: lib.WEIGHTS_EPOCH
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-2)
    model(nodes.float(), edges.float()).square().mean().backward()
    opt.step()
    model.eval()
    check("FusedAdam.step")
    # (3) load_state_dict
    P2 = O.init_params(cfg, seed=9, model="GGNN")
    model.load_state_dict(P2)
    check("load_state_dict")
    # (4) a write the version counter cannot see needs the explicit reset
    with torch.no_grad():
        model(nodes, edges)
    getattr(model.msg_nns[1].seq, "0").weight.data.mul_(0.5)
    model.reset_pass0_cache()
    check("reset_pass0_cache")
    # training-mode forwards never touch the table
    before = model.pass0_cache_stats()
    model.train()
    model(nodes.float(), edges.float()).sum().backward()
    model.eval()
    assert model.pass0_cache_stats() == before


def test_a_full_table_starts_over():
    model, cfg, sh = _model("GGNN", "gdb13")
    nodes, edges = _batch(sh, 300, 21)
    ref = _uncached(model, nodes, edges)
    with torch.no_grad():
        model(nodes, edges)
        rows = model.pass0_cache_stats()["rows"]
        buf = model._p0_state["buf"]
        buf.zero_(); buf[1] = 4096 - 1               # an (artificially) almost full table without this batch's rows
        out = model(nodes, edges)                    # miss -> would not fit -> table emptied, this batch's rows kept
        assert torch.equal(out, ref)
        assert model.pass0_cache_stats()["rows"] == rows
        out = model(nodes, edges)
        assert torch.equal(out, ref) and model.pass0_cache_stats()["hits"] == 1


def test_cache_inside_a_captured_hipgraph():
    model, cfg, sh = _model("GGNN", "gdb13")
    model.sync_free = True
    B = 500
    rounds = [_batch(sh, B, 30 + k) for k in range(4)]
    refs = [_uncached(model, *r) for r in rounds]
    nodes, edges = rounds[0][0].clone(), rounds[0][1].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            model(nodes, edges)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        out = model(nodes, edges)
    for k in (1, 2, 3, 1, 0):
        nodes.copy_(rounds[k][0]); edges.copy_(rounds[k][1])
        g.replay()
        assert float((out - refs[k]).abs().max()) <= 1e-6 * float(refs[k].abs().max())
    st = model.pass0_cache_stats()
    assert st["hits"] >= 2 and model.last_bounded_error() == 0


@pytest.mark.parametrize("kind,shape,B", [("GGNN", "gdb13", 1000), ("AttGGNN", "gdb13", 300)])
@pytest.mark.parametrize("sync_free", [False, True])
def test_weights_cache_is_bitwise_transparent_and_follows_the_weights(kind, shape, B, sync_free):
    """gi_graph.wcache (round 6): the fp16x2 forward chains' weight image, the max |W| cells and the weights' range check
    are derived ONCE per weight version and kept in a device buffer across the forwards of an inference / generation
    loop.  Logits with the cache == logits without it, bit for bit (first forward = derive, later forwards = reuse);
    every way this package changes weights invalidates it (in-place write, FusedAdam through raw pointers,
    load_state_dict); a training forward in between neither uses nor disturbs it."""
    model, cfg, sh = _model(kind, shape)
    model.sync_free = sync_free
    nodes, edges = _batch(sh, B, 5)

    def fwd(cache):
        model.cache_weights = cache
        with torch.no_grad():
            return model(nodes, edges).clone()

    ref = fwd(False)
    assert torch.equal(fwd(True), ref) and model.__dict__["_w_state"]["valid"]          # derive
    assert torch.equal(fwd(True), ref) and torch.equal(fwd(True), ref)                 # reuse
    # (a) in-place write under no_grad: the version counter moves
    with torch.no_grad():
        dict(model.named_parameters())["msg_nns.0.seq.3.weight"].mul_(1.25)
    ref2 = fwd(False)
    assert not torch.equal(ref2, ref)
    assert torch.equal(fwd(True), ref2) and torch.equal(fwd(True), ref2)
    # (b) a training step through FusedAdam (raw-pointer update, lib.WEIGHTS_EPOCH) with a training forward in between
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-2)
    out = model(nodes.float(), edges.float())
    out.square().mean().backward()
    opt.step()
    model.eval()
    ref3 = fwd(False)
    assert not torch.equal(ref3, ref2)
    assert torch.equal(fwd(True), ref3) and torch.equal(fwd(True), ref3)
    # (c) load_state_dict
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["gru.weight_hh"] = sd["gru.weight_hh"] * 0.5
    model.load_state_dict(sd)
    ref4 = fwd(False)
    assert torch.equal(fwd(True), ref4) and torch.equal(fwd(True), ref4)
    if sync_free:
        model.last_bounded_error()


def test_weights_cache_derived_by_a_tiny_batch_serves_a_large_one():
    """The cache's first forward may be a B = 1 call (too few rows for the 16-bit-pipe layers): what it derives must be
    complete — a following B = 1000 forward of the same weights, which does take those launches, reads the max |W| cells
    from the cache (round 6: they had not been written, every graph of the large batch got the same logits)."""
    model, cfg, sh = _model("GGNN", "gdb13")
    small, big = _batch(sh, 1, 3), _batch(sh, 1000, 4)
    model.cache_weights = False
    with torch.no_grad():
        want = model(*big).clone()
    model.cache_weights = True
    with torch.no_grad():
        model(*small)
        assert model.__dict__["_w_state"]["valid"]
        assert torch.equal(model(*big), want)
