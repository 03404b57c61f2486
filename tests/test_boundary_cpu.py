"""CPU: the drop-in boundary — C ABI exports, module API, checkpoint wire format, loud failure
without a GPU.  No device compute is issued."""
import copy
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

from graphinvent_amd import lib as L
from graphinvent_amd.gnn import mpnn
from oracle import ggnn_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/graphinvent"


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "graphinvent_amd.h")).read()
    declared = set(re.findall(r"^(?:int|long long)\s+(gi_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gi_abi_version() == L.ABI_VERSION == 18


def test_host_side_planning_functions():
    lib = L.load()
    d = mpnn._dims_from_constants(O.as_constants(O.make_config()), 1000)
    assert lib.gi_ggnn_num_params(C.byref(d)) == 104
    lay = L.CompactLayout()
    assert lib.gi_compact_layout(1000, 13, 3, C.byref(lay)) == 0 and lay.total_ints > 13000 * 8
    assert lib.gi_compact_layout(4, 200, 3, C.byref(lay)) == -2          # N > GI_MAX_NODES
    ws = lib.gi_ggnn_workspace_floats(C.byref(d), 6900, 12600, 8000, 45)
    ws0 = lib.gi_ggnn_workspace_floats(C.byref(d), 0, 0, 0, 0)
    assert ws > ws0 > 0 and ws % 4 == 0
    was = lib.gi_bf3_enable(-1)                            # bf16x3 switch: query, set, restore; the workspace does
    assert was in (0, 1) and lib.gi_bf3_enable(0) == was   # not depend on it (a tape of either setting feeds either)
    ws_off = lib.gi_ggnn_workspace_floats(C.byref(d), 6900, 12600, 8000, 45)
    assert lib.gi_bf3_enable(1) == 0 and lib.gi_ggnn_workspace_floats(C.byref(d), 6900, 12600, 8000, 45) == ws_off == ws
    lib.gi_bf3_enable(was)
    assert lib.gi_bf3_image_elems(500, 500) == 3 * 500 * 512 and lib.gi_bf3_image_elems(0, 5) < 0
    words = lib.gi_p0_cache_words(C.byref(d))            # pass-0 row cache: header + hash + 4096 rows of ldM floats
    assert words > 4096 * 100 and words % 4 == 0
    assert lib.gi_ggnn_workspace_floats(C.byref(d), 6900, 12600, 12601, 0) == -1   # U > E
    Ut = (C.c_int * 3)(6000, 1900, 100)
    assert lib.gi_ggnn_slab_floats(C.byref(d), 6900, 8000, Ut) > 0
    da = mpnn._dims_from_constants(O.as_constants(O.make_config()), 1000, L.KIND_ATTGGNN)
    assert lib.gi_ggnn_num_params(C.byref(da)) == 134                     # + 3 energy MLPs x 10
    assert lib.gi_ggnn_workspace_floats(C.byref(da), 6900, 12600, 8000, 0) > ws
    da.kind = 7
    assert lib.gi_ggnn_num_params(C.byref(da)) == -1                      # unknown model kind
    d.Fn = d.H + 1
    assert lib.gi_ggnn_workspace_floats(C.byref(d), 1, 1, 1, 0) == -1    # GI_EINVAL


def test_state_dict_is_the_reference_wire_format():
    cfg = O.make_config()
    model = mpnn.GGNN(O.as_constants(cfg))
    sd = model.state_dict()
    shapes = O.param_shapes(cfg)
    assert list(sd) == list(shapes)
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert [n for n, _ in model.named_parameters()] == list(shapes)
    model.load_state_dict(O.init_params(cfg, seed=3))
    clone = copy.deepcopy(model)
    assert all(torch.equal(a, b) for a, b in zip(model.parameters(), clone.parameters()))
    model.eval(); model.train(); model.zero_grad()
    att = mpnn.AttentionGGNN(O.as_constants(cfg))
    shapes = O.param_shapes(cfg, "AttGGNN")
    assert [n for n, _ in att.named_parameters()] == list(shapes) == list(att.state_dict())
    assert all(tuple(p.shape) == shapes[n] for n, p in att.named_parameters())


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this box")
def test_constructor_is_seed_for_seed_identical_to_the_reference():
    for m in [k for k in sys.modules if k == "gnn" or k.startswith("gnn.")]:
        del sys.modules[m]                      # (whatever an earlier test registered under the reference's name)
    sys.path.insert(0, REF)
    try:
        import gnn.mpnn as ref_mpnn
    finally:
        sys.path.remove(REF)
    assert ref_mpnn.__file__.startswith(REF)
    c = O.as_constants(O.make_config())
    for name in ("GGNN", "AttentionGGNN"):
        torch.manual_seed(123)
        ref = getattr(ref_mpnn, name)(c)
        torch.manual_seed(123)
        ours = getattr(mpnn, name)(c)
        a, b = ref.state_dict(), ours.state_dict()
        assert list(a) == list(b), name
        assert all(torch.equal(a[k], b[k]) for k in a), name
    for m in [k for k in sys.modules if k == "gnn" or k.startswith("gnn.")]:
        del sys.modules[m]


def test_forward_fails_loudly_off_gpu():
    cfg = O.make_config()
    model = mpnn.GGNN(O.as_constants(cfg))
    with pytest.raises(RuntimeError, match="no CPU"):
        model(torch.zeros(2, 13, 8), torch.zeros(2, 13, 13, 3))


def test_dropout_in_training_mode_takes_the_hip_path_too():
    """AlphaDropout p > 0 is implemented in HIP (tests/test_dropout_gpu.py): no eager fallback here
    either — off the GPU it fails like every other forward; the seed logic is host-side."""
    cfg = O.make_config(mlp1_dropout_p=0.1)
    model = mpnn.GGNN(O.as_constants(cfg)).train()
    with pytest.raises(RuntimeError, match="no CPU"):
        model(torch.zeros(2, 13, 8), torch.zeros(2, 13, 13, 3))
    model.dropout_seed = 5
    assert model._next_dropout_seed() == 5 and model.last_dropout_seed == 5
    assert model.eval()._next_dropout_seed() is None
    assert mpnn.GGNN(O.as_constants(O.make_config())).train()._next_dropout_seed() is None


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or"):
        L.load()


def test_drop_in_import_as_top_level_gnn_package():
    """Workflow.py:23 does `import gnn.mpnn`: with graphinvent_amd/ first on sys.path that must
    resolve to this implementation (fresh interpreter so the reference's `gnn` cannot interfere)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import gnn.mpnn as m; "
            "assert m.__file__.startswith(%r), m.__file__; "
            "assert hasattr(m, 'GGNN'); print('ok')") % (os.path.join(ROOT, "graphinvent_amd"),
                                                         os.path.join(ROOT, "graphinvent_amd"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-800:]


def test_graft_entry_build_runs_without_a_gpu():
    """The driver's "does it build" check: compiles every HIP source for gfx950 and binds the ABI."""
    import __graft_entry__ as entry
    entry.build()


def test_graph_capture_flag_state_is_recorded_at_import():
    """Round-3 advisor finding: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 only works when it is in the environment before
    the HIP runtime initialises; the package records whether that can be relied on and refuses capture otherwise."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import graphinvent_amd as g, os; ok, why = g.graph_capture_safe(); "
            "print(int(ok), os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'));\n"
            "try:\n    g.assert_graph_safe(); print('fine')\nexcept RuntimeError as e:\n    print('refused')")
    def run(env_extra):
        env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
        env.update(env_extra, PYTHONPATH=root)
        return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                              check=True).stdout.split()
    assert run({}) == ["1", "0", "fine"]                                       # set by the package, in time
    assert run({"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}) == ["1", "0", "fine"]  # the caller's own 0
    assert run({"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"}) == ["0", "1", "refused"]


def test_ctypes_structs_have_the_sizes_and_offsets_of_the_header(tmp_path):
    """graphinvent_amd/lib.py mirrors the header's structs by hand (ctypes): a field added on one side only shifts
    everything behind it silently.  A C program compiled from include/graphinvent_amd.h with gcc prints sizeof of every
    mirrored struct and the offsets of the fields added last; they must equal ctypes' view."""
    import shutil
    import subprocess
    import sys
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("gi_compact_layout_t", L.CompactLayout), ("gi_gemm_params", L.GemmParams), ("gi_chain_layer", L.ChainLayer),
             ("gi_chain_params", L.ChainParams), ("gi_reduce_desc", L.ReduceDesc), ("gi_absmax_desc", L.AbsmaxDesc),
             ("gi_bf3_pack_desc", L.Bf3PackDesc), ("gi_graph", L.Graph), ("gi_ggnn_dims", L.GgnnDims),
             ("gi_dropout_params", L.DropoutParams)]
    offs = [("gi_gemm_params", "c_amax", L.GemmParams.c_amax.offset),
            ("gi_gemm_params", "x2_guard_host", L.GemmParams.x2_guard_host.offset),
            ("gi_graph", "x2_guard_host", L.Graph.x2_guard_host.offset),
            ("gi_graph", "wcache_valid", L.Graph.wcache_valid.offset),
            ("gi_gemm_params", "m_dev", L.GemmParams.m_dev.offset),
            ("gi_chain_params", "x2_wamax", L.ChainParams.x2_wamax.offset),
            ("gi_chain_params", "x_amax", L.ChainParams.x_amax.offset),
            ("gi_chain_params", "x2_rows32", L.ChainParams.x2_rows32.offset),
            ("gi_chain_layer", "out_amax", L.ChainLayer.out_amax.offset),
            ("gi_chain_params", "image_stride", L.ChainParams.image_stride.offset),
            ("gi_graph", "p0_cache", L.Graph.p0_cache.offset),
            ("gi_ggnn_dims", "drop_seed", L.GgnnDims.drop_seed.offset)]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "graphinvent_amd.h"', 'int main(void) {']
    src += [f'  printf("%zu\\n", sizeof({c}));' for c, _ in pairs]
    src += [f'  printf("%zu\\n", offsetof({c}, {f}));' for c, f, _ in offs]
    src += ['  printf("%d %d %d %d\\n", GI_ABI_VERSION, GI_MAX_GROUPS, GI_CHAIN_MAXL, GI_AMAX_WORDS);', '  return 0;', '}']
    cfile, exe = tmp_path / "sizes.c", tmp_path / "sizes"
    cfile.write_text("\n".join(src))
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-I", inc, str(cfile), "-o", str(exe)], check=True)          # (the header is plain C)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    got = [int(x) for x in out[:len(pairs) + len(offs)]]
    for (c, t), n in zip(pairs, got):
        assert C.sizeof(t) == n, (c, C.sizeof(t), n)
    for (c, f, o), n in zip(offs, got[len(pairs):]):
        assert o == n, (c, f, o, n)
    abi, groups, maxl, words = (int(x) for x in out[len(pairs) + len(offs)].split())
    assert (abi, groups, maxl, words) == (L.ABI_VERSION, L.GI_MAX_GROUPS, L.CHAIN_MAXL, L.AMAX_WORDS)


def test_every_environment_switch_the_library_reads_is_documented_in_the_header():
    """`getenv("GI_…")` in csrc/ against the conventions block of include/graphinvent_amd.h (lab-only macros of the
    tools/ programs — GI_LAB_* — are not the library's)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    csrc = os.path.join(root, "graphinvent_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".h")):
            read |= set(re.findall(r'getenv\("(GI_[A-Z0-9_]+)"\)', open(os.path.join(csrc, name)).read()))
    header = open(os.path.join(root, "include", "graphinvent_amd.h")).read()
    block = header[:header.index("#ifndef GRAPHINVENT_AMD_H")]
    documented = set(re.findall(r"\b(GI_[A-Z0-9_]+)\b", block))
    assert read and read <= documented, sorted(read - documented)
