"""Build-container helper (needs /root/reference): import the UNMODIFIED reference callers of the hot path —
``Workflow`` (Workflow.py) and ``GraphGenerator`` (GraphGenerator.py) — with stub modules for what cannot be imported
here (rdkit, h5py, tensorboard, matplotlib-using ``util``, ``Analyzer``, ``DataProcesser``, ``GraphGeneratorRL``,
``ScoringFunction``, ``MolecularGraph``, ``parameters.constants``; none of them is touched by the methods under test:
``get_dataloader``, ``define_model_and_optimizer``, ``create_model``, ``train_epoch``, ``validation_epoch``, ``loss``,
``build_graphs`` and what it calls).  ``which`` decides which ``gnn`` / ``BlockDatasetLoader`` the unmodified callers
bind: "reference" (``/root/reference/graphinvent`` first on ``sys.path``: golden generation on CPU) or "dropin"
(``graphinvent_amd/`` first: the MI355X modules under the reference's names)."""
import os
import sys
import types
from collections import namedtuple

import numpy as np

REF = os.environ.get("GI_REFERENCE_DIR", "/root/reference/graphinvent")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# GDB-13 preprocessing parameters of the shipped fixtures (data/pre-training/gdb13_1K-debug/preprocessing_params.csv)
N, ATOMS, CHARGES, BONDS = 13, 5, 3, 3


def have_reference() -> bool:
    return os.path.exists(os.path.join(REF, "Workflow.py"))


def constants_dict(device: str, model_cfg: dict, data_dir: str, batch_size: int, epochs: int) -> dict:
    """Every ``constants`` field the methods under test read (parameters/constants.py:159-211, defaults.py)."""
    d = dict(model_cfg)
    d.update(
        device=device, model="GGNN", job_type="train", restart=False, job_dir=data_dir + "/", dataset_dir=data_dir + "/",
        training_set=os.path.join(data_dir, "train.smi"), validation_set=os.path.join(data_dir, "valid.smi"),
        test_set=os.path.join(data_dir, "test.smi"), batch_size=batch_size, block_size=100000, n_workers=0,
        init_lr=1e-4, max_rel_lr=1.0, min_rel_lr=1e-4, epochs=epochs,
        dim_nodes=[N, ATOMS + CHARGES], dim_edges=[N, N, BONDS], dim_f_add=[N, ATOMS, CHARGES, BONDS],
        dim_f_conn=[N, BONDS], n_atom_types=ATOMS, n_formal_charge=CHARGES, n_imp_H=0, n_chirality=0,
        use_explicit_H=False, ignore_H=True, use_chirality=False, tensorboard_dir=data_dir + "/tb/")
    return d


def as_constants(d: dict):
    return namedtuple("CONSTANTS", sorted(d))(**d)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _H5File:
    """h5py.File stand-in over graphinvent_amd.loader's libhdf5 reader: ``.get(name)`` returns the dataset as an int8
    array (the reference slices / indexes it and calls ``torch.from_numpy`` on the result, BlockDatasetLoader.py:128-141)."""

    def __init__(self, path, mode="r", swmr=False):
        from graphinvent_amd.loader import read_hdf_int8
        n, e, a = read_hdf_int8(path)
        self._d = {"nodes": n, "edges": e, "APDs": a}

    def get(self, name):
        return self._d[name]


def load(which: str, constants):
    """-> (Workflow module, GraphGenerator module), freshly imported with `constants` as ``parameters.constants``."""
    assert which in ("reference", "dropin") and have_reference()
    for name in list(sys.modules):
        if name in ("Workflow", "GraphGenerator", "BlockDatasetLoader", "gnn") or name.startswith("gnn."):
            del sys.modules[name]
    if ROOT not in sys.path:
        sys.path.append(ROOT)                                   # graphinvent_amd as a package (h5 reader, drop-ins)
    _stub("rdkit")
    _stub("h5py", File=_H5File)
    _stub("Analyzer", Analyzer=object)
    _stub("DataProcesser", DataProcesser=object)
    _stub("GraphGeneratorRL", GraphGeneratorRL=object)
    _stub("ScoringFunction", ScoringFunction=object)
    _stub("MolecularGraph", GenerationGraph=object)
    _stub("util")
    _stub("parameters")
    _stub("parameters.constants", constants=constants)
    import torch.utils
    if "torch.utils.tensorboard" not in sys.modules:
        try:
            import torch.utils.tensorboard  # noqa: F401
        except Exception:
            torch.utils.tensorboard = _stub("torch.utils.tensorboard", SummaryWriter=object)
    first = REF if which == "reference" else os.path.join(ROOT, "graphinvent_amd")
    path0 = list(sys.path)
    sys.path[:] = [first, REF] + [p for p in path0 if p not in (first, REF)]
    try:
        import Workflow
        import GraphGenerator
        import BlockDatasetLoader
        import gnn.mpnn
    finally:
        sys.path[:] = path0
    assert Workflow.__file__.startswith(REF) and GraphGenerator.__file__.startswith(REF)
    assert gnn.mpnn.__file__.startswith(first) and BlockDatasetLoader.__file__.startswith(first), \
        (gnn.mpnn.__file__, BlockDatasetLoader.__file__)
    return Workflow, GraphGenerator


_TOUCHED = ("Workflow", "GraphGenerator", "BlockDatasetLoader", "gnn", "rdkit", "h5py", "Analyzer", "DataProcesser",
            "GraphGeneratorRL", "ScoringFunction", "MolecularGraph", "util", "parameters", "parameters.constants",
            "torch.utils.tensorboard")


class isolated:
    """Context manager / pytest-fixture body: whatever ``load`` and ``pin_multinomial`` change process-wide — the stub
    and caller modules in ``sys.modules`` (``gnn.*`` included), ``sys.path``, ``torch.distributions.Multinomial`` — is put
    back on exit, so that tests run in any order (a later ``import gnn.mpnn`` must not find the drop-in registered
    under the reference's name; round-4 advisor finding)."""

    def __enter__(self):
        import torch
        self._mods = {k: v for k, v in sys.modules.items() if k in _TOUCHED or k.startswith("gnn.")}
        self._path = list(sys.path)
        self._multinomial = torch.distributions.Multinomial
        self._tb = getattr(torch.utils, "tensorboard", None)
        return self

    def __exit__(self, *exc):
        import torch
        for k in [k for k in sys.modules if k in _TOUCHED or k.startswith("gnn.")]:
            del sys.modules[k]
        sys.modules.update(self._mods)
        sys.path[:] = self._path
        torch.distributions.Multinomial = self._multinomial
        if self._tb is None and hasattr(torch.utils, "tensorboard") and "torch.utils.tensorboard" not in sys.modules:
            del torch.utils.tensorboard
        return False


def pin_multinomial(draw):
    """Replace ``torch.distributions.Multinomial`` (GraphGenerator.py:533-537) by the seeded inverse-CDF draw."""
    import torch

    class PinnedMultinomial:
        def __init__(self, total_count, probs):
            assert total_count == 1
            self.probs = probs

        def sample(self):
            p = self.probs.detach().cpu().numpy()
            idx = draw(p)
            one_hot = torch.zeros(p.shape, dtype=self.probs.dtype)
            one_hot[torch.arange(p.shape[0]), torch.from_numpy(idx)] = 1
            return one_hot.to(self.probs.device)

    torch.distributions.Multinomial = PinnedMultinomial
