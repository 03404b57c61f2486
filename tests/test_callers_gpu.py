"""-m gpu: the reference's CALLERS drive the MI355X drop-ins unchanged (BASELINE.json north_star: "keeps the existing
model-class constructor / forward API so Workflow.py and GraphGenerator.py call it unchanged").

With ``graphinvent_amd/`` ahead of the reference's ``graphinvent/`` on ``sys.path``, ``import gnn.mpnn`` and
``from BlockDatasetLoader import BlockDataLoader, HDFDataset`` (Workflow.py:19, 23) resolve to the drop-in modules.
The caller code that then runs is

* the UNMODIFIED ``Workflow`` / ``GraphGenerator`` classes when a reference checkout is visible (``GI_REFERENCE_DIR``
  or /root/reference — never on the GPU pool's boxes; the run of this file with the two reference files staged beside
  the snapshot is logged in profiles/r04/unchanged_callers_gpu.log, tools/prove_unchanged_callers.sh), otherwise
* ``oracle/callers_oracle.py``, the restatement that tests/golden/make_golden_callers.py proved bit-identical to the
  unmodified methods on CPU.

Both are compared with what the unmodified callers + the reference's own ``gnn`` produced on CPU
(tests/golden/golden_workflow.npz, golden_generator.npz): first-step loss 1e-4 (north_star's bar), the 4-step loss
trajectory within 2e-3 (Adam turns a rounding-noise gradient into a +-lr step, see test_dp_gpu.py), the validation loss
after every step likewise; the generation loop — 22 rounds, 50 molecules of 10-11 atoms with the draws pinned — produces
the SAME graphs, node for node and bond for bond, and the same per-action likelihoods to 1e-4."""
import os
import sys

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

from tests.h5util import have_libhdf5, write_h5

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dropin_modules():
    """``gnn.mpnn`` / ``BlockDatasetLoader`` as the reference's callers import them, bound to graphinvent_amd/."""
    for name in list(sys.modules):
        if name in ("BlockDatasetLoader", "gnn") or name.startswith("gnn."):
            del sys.modules[name]
    first = os.path.join(ROOT, "graphinvent_amd")
    sys.path.insert(0, first)
    try:
        import BlockDatasetLoader
        import gnn.mpnn
    finally:
        sys.path.remove(first)
    assert gnn.mpnn.__file__.startswith(first) and BlockDatasetLoader.__file__.startswith(first)
    return gnn.mpnn, BlockDatasetLoader


def callers(constants):
    """(Workflow class, GraphGenerator factory(model, batch, draw), what): unmodified when the reference is visible."""
    if RC.have_reference():
        WF, GG = RC.load("dropin", constants)
        return WF.Workflow, (lambda m, b, draw: GG.GraphGenerator(model=m, batch_size=b)), "unmodified reference callers"
    dropin_modules()
    return CO.WorkflowOracle, (lambda m, b, draw: CO.GeneratorOracle(m, b, constants, draw)), "oracle/callers_oracle.py"


@pytest.fixture
def data_dir(tmp_path, golden_dir):
    if not have_libhdf5():
        pytest.skip("no libhdf5 on this box")
    d = np.load(os.path.join(golden_dir, "gdb13_1K-debug_valid.npz"))
    for name in ("train", "valid", "test"):                       # the shipped valid.h5 under all three names
        write_h5(str(tmp_path / f"{name}.h5"), d["nodes"], d["edges"], d["APDs"])
    return str(tmp_path)


def test_workflow_training_path_on_the_dropins_matches_the_reference_run(data_dir, golden_dir):
    G = np.load(os.path.join(golden_dir, "golden_workflow.npz"))
    consts = RC.as_constants(RC.constants_dict("cuda", O.make_config(), data_dir, batch_size=int(G["batch_size"]),
                                               epochs=int(G["epochs"])))
    Workflow, _, what = callers(consts)
    torch.manual_seed(int(G["seed"]))
    wf = Workflow(constants=consts) if what.startswith("unmodified") else Workflow(consts)
    wf.train_dataloader = wf.get_dataloader(wf.train_h5_path, "training set")
    wf.valid_dataloader = wf.get_dataloader(wf.valid_h5_path, "validation set")
    assert type(wf.train_dataloader).__module__ == "BlockDatasetLoader" and len(wf.train_dataloader) == 1
    start, end = wf.define_model_and_optimizer()
    assert type(wf.model).__module__ == "gnn.mpnn" and next(wf.model.parameters()).is_cuda
    from tests.golden.spec import digest
    init = np.stack([digest(v.cpu()) for v in wf.model.state_dict().values()])
    assert np.allclose(init, G["init_digest"], rtol=1e-6, atol=1e-7)       # seed-for-seed the reference's initial weights
    train, valid = [], []
    for epoch in range(start, end):
        wf.current_epoch = epoch
        train.append(float(wf.train_epoch()))
        valid.append(float(wf.validation_epoch()))
    train, valid = np.array(train), np.array(valid)
    print(f"\n[{what}] train {train} (reference {G['train']})  valid {valid} (reference {G['valid']})")
    assert abs(train[0] - G["train"][0]) < 1e-4 * abs(G["train"][0])
    assert np.all(np.abs(train - G["train"]) < 2e-3 * np.abs(G["train"]))
    assert np.all(np.abs(valid - G["valid"]) < 2e-3 * np.abs(G["valid"]))
    lr_steps = 4 * 1e-4
    for k in ("APDReadout.fTermNet2.seq.12.bias", "gru.bias_hh", "msg_nns.0.seq.0.weight"):
        got = wf.model.state_dict()[k].cpu().numpy()
        assert np.abs(got - G["final::" + k]).max() <= 2 * lr_steps + 1e-6, k


@pytest.mark.parametrize("sync_free", [False, True])
def test_generation_loop_on_the_dropin_model_builds_the_reference_graphs(golden_dir, sync_free):
    G = np.load(os.path.join(golden_dir, "golden_generator.npz"))
    cfg = O.make_config(**{str(k): int(v) for k, v in zip(G["cfg_keys"], G["cfg_vals"])})
    consts = RC.as_constants(RC.constants_dict("cuda", cfg, "/nonexistent", batch_size=100, epochs=1))
    _, make_generator, what = callers(consts)
    import gnn.mpnn                                               # the drop-in, under the reference's name
    model = gnn.mpnn.GGNN(constants=consts)
    model.load_state_dict({k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("w::")})
    model = model.to("cuda").eval()
    model.sync_free = sync_free                                   # the host-sync-free forward built for this loop
    draw = CO.InverseCdfDraws(int(G["draw_seed"]), int(G["batch"]))
    RC.pin_multinomial(draw)
    with torch.no_grad():
        gen = make_generator(model, int(G["batch"]), draw)
        n = gen.build_graphs()
    if sync_free:
        assert model.last_bounded_error() == 0                    # one read-back for all 22 rounds
    assert (n, draw.round) == (int(G["n_generated"]), int(G["rounds"])), (what, n, draw.round)
    assert draw.margin > 1e-5
    assert np.array_equal(gen.generated_n_nodes.cpu().numpy(), G["n_nodes"])
    assert np.array_equal(gen.generated_nodes.cpu().numpy().astype(np.int8), G["nodes"])
    assert np.array_equal(gen.generated_edges.cpu().numpy().astype(np.int8), G["edges"])
    assert np.array_equal(gen.properly_terminated.cpu().numpy(), G["terminated"])
    like, ref = gen.generated_likelihoods.cpu().numpy(), G["likelihoods"]
    assert np.abs(like - ref).max() < 1e-4 * ref.max()
    print(f"\n[{what}, sync_free={sync_free}] {n} molecules in {draw.round} rounds, node counts "
          f"{np.bincount(G['n_nodes'][:n].astype(int))}, closest draw to a CDF boundary {draw.margin:.2e}")
