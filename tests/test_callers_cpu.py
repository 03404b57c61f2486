"""CPU: the callers of the hot path (Workflow.py, GraphGenerator.py).  Needs the reference checkout for the parts that
run its UNMODIFIED code (skipped where it is absent, e.g. on the GPU box):

* oracle/callers_oracle.py reproduces the unmodified ``Workflow`` training path and ``GraphGenerator.build_graphs`` bit
  for bit (the same check tests/golden/make_golden_callers.py makes before it writes the golden files);
* the unmodified ``Workflow``, bound to the DROP-IN ``gnn`` / ``BlockDatasetLoader`` modules (graphinvent_amd/ first on
  sys.path), constructs the model, the loader, the optimizer and the scheduler and computes its loss on the drop-in
  loader's int8 batches — everything up to the forward itself, which has no CPU fallback and says so."""
import os
import shutil

import numpy as np
import pytest
import torch

from oracle import callers_oracle as CO
from oracle import ggnn_oracle as O
from tests.golden import ref_callers as RC


@pytest.fixture(autouse=True)
def _restore_process_state():
    """ref_callers.load / pin_multinomial replace modules and torch.distributions.Multinomial process-wide."""
    with RC.isolated():
        yield


FIXTURE = "/root/reference/data/pre-training/gdb13_1K-debug/valid.h5"
needs_ref = pytest.mark.skipif(not (RC.have_reference() and os.path.exists(FIXTURE)
                                    and os.path.exists("/opt/conda/lib/libhdf5.so")),
                               reason="reference checkout / libhdf5 not on this box")
SMALL = dict(hidden_node_features=24, message_size=24, enn_hidden_dim=32, gather_width=24, gather_att_hidden_dim=32,
             gather_emb_hidden_dim=32, mlp1_hidden_dim=48, mlp2_hidden_dim=48)


@pytest.fixture
def data_dir(tmp_path):
    for name in ("train", "valid", "test"):
        shutil.copy(FIXTURE, str(tmp_path / f"{name}.h5"))
    return str(tmp_path)


def _train(make, consts, steps):
    torch.manual_seed(3)
    wf = make(consts)
    wf.train_dataloader = wf.get_dataloader(wf.train_h5_path, "training set")
    wf.valid_dataloader = wf.get_dataloader(wf.valid_h5_path, "validation set")
    start, end = wf.define_model_and_optimizer()
    out = []
    for epoch in range(start, end):
        wf.current_epoch = epoch
        out.append((float(wf.train_epoch()), float(wf.validation_epoch())))
    return out, wf


@needs_ref
def test_restated_callers_equal_the_unmodified_reference_bit_for_bit(data_dir):
    consts = RC.as_constants(dict(RC.constants_dict("cpu", O.make_config(**SMALL), data_dir, batch_size=100, epochs=3),
                                  init_lr=1e-3))
    WF, GG = RC.load("reference", consts)
    ref, wf_ref = _train(lambda c: WF.Workflow(constants=c), consts, 3)
    mine, wf_mine = _train(lambda c: CO.WorkflowOracle(c), consts, 3)
    assert ref == mine
    assert all(torch.equal(a, b) for a, b in zip(wf_ref.model.state_dict().values(), wf_mine.model.state_dict().values()))
    outs = []
    for make in (lambda m, b, draw: GG.GraphGenerator(model=m, batch_size=b),
                 lambda m, b, draw: CO.GeneratorOracle(m, b, consts, draw)):
        draw = CO.InverseCdfDraws(0, 24)
        RC.pin_multinomial(draw)
        with torch.no_grad():
            gen = make(wf_ref.model.eval(), 24, draw)
            n = gen.build_graphs()
        outs.append((n, draw.round, gen.generated_nodes.clone(), gen.generated_edges.clone(), gen.generated_n_nodes.clone(),
                     gen.generated_likelihoods.clone(), gen.properly_terminated.clone()))
    assert outs[0][:2] == outs[1][:2] and all(torch.equal(a, b) for a, b in zip(outs[0][2:], outs[1][2:]))


@needs_ref
def test_unmodified_workflow_binds_the_dropin_modules(data_dir):
    consts = RC.as_constants(RC.constants_dict("cpu", O.make_config(**SMALL), data_dir, batch_size=32, epochs=2))
    WF, _ = RC.load("dropin", consts)
    wf = WF.Workflow(constants=consts)
    wf.train_dataloader = wf.get_dataloader(wf.train_h5_path, "training set")
    assert type(wf.train_dataloader).__module__ == "BlockDatasetLoader"
    assert "graphinvent_amd" in __import__("BlockDatasetLoader").__file__
    assert len(wf.train_dataloader) == 4                           # 100 rows, ragged last minibatch kept like the reference
    start, end = wf.define_model_and_optimizer()                   # create_model + Adam + OneCycleLR on the drop-in
    assert (start, end) == (1, 3) and type(wf.model).__module__ == "gnn.mpnn"
    assert "graphinvent_amd" in __import__("gnn.mpnn").mpnn.__file__
    ref_keys = list(O.param_shapes(O.make_config(**SMALL)).keys())
    assert list(wf.model.state_dict().keys()) == ref_keys          # the checkpoint wire format
    nodes, edges, apds = next(iter(wf.train_dataloader))
    assert nodes.dtype == torch.int8 and nodes.shape == (32, 13, 8)
    loss = wf.loss(output=torch.randn(32, 625), target_output=apds)      # Workflow.loss on int8 targets, unchanged
    assert torch.isfinite(loss)
    with pytest.raises(RuntimeError):                              # no CPU fallback: the forward needs the MI355X
        wf.model(nodes, edges)
