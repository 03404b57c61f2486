"""-m gpu: the fp16x2 arithmetic (csrc/gi_x2.h: two scaled fp16 planes per fp32 operand, ONE power-of-two scale per
tensor, three f16 MFMA products) on trial — round-4 verdict, item 1.

What the per-tensor scale gives up against fp32 (`MLP.forward`, gnn/modules.py:166-170, in the reference) and against
the bf16x3 split is DYNAMIC RANGE INSIDE ONE TENSOR: an element keeps 22 bits while it lies within 2^16 of the tensor's
largest magnitude; below that its absolute error stays at 2^-38 of that maximum.  Measured here, next to the fp32 MFMA
chain and the bf16x3 split on the same operands, per TENSOR (max |d| / max |ref|, the bar of every other test) and per
ROW (max_j |d_ij| / max_j sum_k |a_ik| |w_jk|: what a caller who looks at single graphs sees):

  * kernel level — 1 % of A's rows x 1e4, x 1e-6, one outlier element x 1e5, both together (rows 1e-11 below the
    maximum: the case fp16x2 cannot represent), a dominant / tiny rows of W; the weight-gradient layout with a few
    rows carrying the whole gradient and with a dead output channel;
  * the dynamic-range GUARD (gi_gemm_params.x2_guard, gi_x2_weight_guard): counts exactly the rows / weight lines the
    scale leaves with fewer than ~14 bits, sets the host-mapped flag, ignores exactly-zero rows;
  * model level — a GGNN TRAINED for 300 FusedAdam steps at lr 1e-3 (weights and gradients of a fitted model, not of
    an initialisation), then logits / loss / every gradient tensor against the oracle fed the same weights at 1e-4
    (both pins) in all three arithmetic modes; the guard stays silent.

`python tests/test_x2_trial_gpu.py` prints the table committed as profiles/r05/x2_trial.txt."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphinvent_amd import lib as L, ops, synthetic                      # noqa: E402
from graphinvent_amd.gnn import mpnn                                        # noqa: E402
from graphinvent_amd.loss import apd_kl_loss                                # noqa: E402
from graphinvent_amd.optim import FusedAdam                                 # noqa: E402
from oracle import ggnn_oracle as O                                         # noqa: E402
from tests import pins                                                      # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
MODES = ("fp32", "bf16x3", "fp16x2")
M, N, K = 2600, 500, 500


def _operands(case: str, seed: int = 0):
    """A [M, K] and W [N, K] with the intra-tensor range of `case`; returns (A, W, indices of the scaled rows)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    rows = torch.randperm(M, generator=g)[:M // 100]
    wrows = torch.randperm(N, generator=g)[:max(N // 100, 1)]
    if case == "rows_x1e4":
        A[rows] *= 1e4
    elif case == "rows_x1e-6":
        A[rows] *= 1e-6
    elif case == "outlier_x1e5":
        A[7, 11] *= 1e5
    elif case == "rows_x1e-6_and_outlier_x1e5":
        A[rows] *= 1e-6
        free = [i for i in range(M) if i not in set(rows.tolist())][0]
        A[free, 11] = 3e5
    elif case == "zero_rows":
        A[rows] = 0.0
    elif case == "w_dominant_row":
        W[3] *= 1e4
    elif case == "w_rows_x1e-6":
        W[wrows] *= 1e-6
    elif case == "w_rows_x1e-9":
        W[wrows] *= 1e-9
    elif case != "uniform":
        raise KeyError(case)
    return A, W, rows, wrows


def _product(mode: str, Ad, Wd, guard=None, flag=0):
    """Y = A W^T on the device in one of the three arithmetics (forward layout, plain store)."""
    Y = torch.empty(M, ops.r4(N), device=DEV)
    if mode == "fp32":
        ops.gemm(Ad, Wd, Y, M, N, K, K, K, ops.r4(N), flags=0)
        return Y[:, :N]
    F = L.GEMM_BF3 | L.GEMM_BF3B_F32
    if mode == "bf16x3":
        ops.gemm(Ad, Wd, Y, M, N, K, K, K, ops.r4(N), flags=F)
        return Y[:, :N]
    cells = torch.zeros(2, L.AMAX_WORDS, device=DEV)
    ops.absmax([Ad, Wd], cells)
    ops.gemm(Ad, Wd, Y, M, N, K, K, K, ops.r4(N), flags=F | L.GEMM_X2, a_amax=cells[0], b_amax=cells[1],
             x2_guard=guard, x2_guard_host=flag)
    return Y[:, :N]


def _errors(Y, A, W):
    ref = A.double() @ W.double().t()
    S = A.double().abs() @ W.double().abs().t()                  # sum_k |a_ik| |w_jk|
    d = (Y.double().cpu() - ref).abs()
    tensor = float(d.max() / ref.abs().max())
    row = float((d.max(1).values / S.max(1).values.clamp_min(1e-300)).max())
    col = float((d.max(0).values / S.max(0).values.clamp_min(1e-300)).max())
    return tensor, row, col


CASES = ("uniform", "rows_x1e4", "rows_x1e-6", "outlier_x1e5", "rows_x1e-6_and_outlier_x1e5", "w_dominant_row",
         "w_rows_x1e-6", "w_rows_x1e-9")


def measure(case: str):
    A, W, rows, wrows = _operands(case)
    Ad, Wd = A.to(DEV), W.to(DEV)
    return {mode: _errors(_product(mode, Ad, Wd), A, W) for mode in MODES}


@pytest.mark.parametrize("case", CASES)
def test_forward_product_under_intra_tensor_dynamic_range(case):
    e = measure(case)
    print(f"\n[x2 trial] {case}: " + "; ".join(f"{m}: tensor {e[m][0]:.1e} row {e[m][1]:.1e} col {e[m][2]:.1e}" for m in MODES))
    for mode in ("fp32", "bf16x3"):                              # fp32's exponent range: every row, every column
        assert max(e[mode]) < 3e-6, (mode, e[mode])              # (the fp32 MFMA chain itself: 2.0e-6 on the outlier's column)
    t, r, c = e["fp16x2"]
    assert t < 2e-6, t                                           # per tensor: always (the bar of the parity suite)
    if case in ("uniform", "rows_x1e4", "w_dominant_row"):
        assert r < 2e-6 and c < 2e-6, (r, c)                     # everything within 2^16 of the maximum: 22 bits
    elif case == "outlier_x1e5":                                 # every other row 2^-16.6 below the maximum: the border
        assert r < 4e-6 and c < 2e-6, (r, c)
    elif case == "rows_x1e-6":                                   # 2^-20 below: ~18 bits left
        assert r < 2e-5, r
    elif case == "w_rows_x1e-6":
        assert c < 2e-5, c
    elif case == "rows_x1e-6_and_outlier_x1e5":                  # 2^-38 below the maximum: flushed — the guard's case
        assert r > 1e-2, r
    elif case == "w_rows_x1e-9":                                 # 2^-30 below: a handful of bits
        assert c > 1e-4, c


@pytest.mark.parametrize("kernel", ["r3", "b3p"])
def test_guard_counts_exactly_the_rows_outside_the_range(kernel):
    """gi_gemm_params.x2_guard in both fp16x2 forward kernels (gi_gemm_bf3_kernel, the pipelined gi_b3p_kernel): rows
    more than 2^24 below max |A| are counted once per launch and set the host-mapped flag; rows 2^20 below (18 bits
    left), a tensor with an outlier only, and exactly-zero rows are not."""
    lib = L.load()
    prev = lib.gi_b3p_enable(-1), lib.gi_b3v_enable(-1)
    lib.gi_b3p_enable(1 if kernel == "b3p" else 0); lib.gi_b3v_enable(0)
    os.environ["GI_B3P_ALL"] = "1"
    flag = ops.HostFlag()
    try:
        for case, expect in (("uniform", 0), ("rows_x1e-6", 0), ("outlier_x1e5", 0), ("zero_rows", 0),
                             ("rows_x1e-6_and_outlier_x1e5", M // 100)):
            A, W, rows, _ = _operands(case)
            counter = torch.zeros(4, dtype=torch.int32, device=DEV)
            flag.value = 0
            _product("fp16x2", A.to(DEV), W.to(DEV), guard=counter, flag=flag.dev)
            torch.cuda.synchronize()
            assert counter.tolist() == [expect, 0, 0, 0], (case, counter.tolist())
            assert flag.value == (1 if expect else 0), case
    finally:
        flag.close()
        lib.gi_b3p_enable(prev[0]); lib.gi_b3v_enable(prev[1])
        os.environ.pop("GI_B3P_ALL", None)


def test_weight_guard_counts_rows_and_columns_outside_the_range():
    g = torch.Generator().manual_seed(5)
    W = torch.randn(500, 250, generator=g)
    W2 = W.clone(); W2[[3, 77, 499]] *= 1e-9; W2[:, [0, 249]] *= 1e-9          # 3 rows and 2 columns 2^-30 below
    W3 = W.clone(); W3[[5]] *= 1e-6; W3[9] = 0.0                                # 2^-20 below / exactly zero: not counted
    flag = ops.HostFlag()
    try:
        for mats, expect in (([W], 0), ([W3], 0), ([W2], 5), ([W, W2, W3, W2.t().contiguous()], 10)):
            dev = [m.to(DEV) for m in mats]
            cells = torch.zeros(len(dev), L.AMAX_WORDS, device=DEV)
            ops.absmax(dev, cells)
            counter = torch.zeros(1, dtype=torch.int32, device=DEV)
            flag.value = 0
            ops.x2_weight_guard(dev, cells, counter, flag.dev)
            torch.cuda.synchronize()
            assert int(counter) == expect, (expect, int(counter))
            assert flag.value == (1 if expect else 0)
    finally:
        flag.close()


def measure_wgrad(case: str):
    """[dW | db] = dZ^T [X | 1] (weight-gradient layout of the pipelined kernel) — per tensor and per OUTPUT ROW
    (max_i |d_oi| / max_i sum_r |dz_ro| |x_ri|)."""
    rows, n_out, n_in, nsplit = 7258, 500, 500, 8
    g = torch.Generator().manual_seed(3)
    dZ = torch.randn(rows, n_out, generator=g) * 1e-3
    X = torch.randn(rows, n_in, generator=g)
    hot = torch.randperm(rows, generator=g)[:rows // 100]
    if case == "few_rows_carry_it":
        dZ[hot] *= 1e4                                           # a few graphs carry the whole gradient (late training)
    elif case == "rows_x1e-8":
        dZ[hot] *= 1e-8                                          # well-fitted graphs: next to nothing
    elif case == "dead_channel":
        dZ[:, 17] *= 2.0 ** -30                                  # one output unit 2^30 below the rest, for every row
    ldc = ops.r4(n_in + 1)
    stride = ops.r4(n_out * ldc)
    ref = torch.cat([dZ.double().t() @ X.double(), dZ.double().sum(0)[:, None]], 1)
    S = torch.cat([dZ.double().abs().t() @ X.double().abs(), dZ.double().abs().sum(0)[:, None]], 1)
    dZd, Xd = dZ.to(DEV), X.to(DEV)
    cells = torch.zeros(2, L.AMAX_WORDS, device=DEV)
    ops.absmax([dZd, Xd], cells)
    out = {}
    for mode in MODES:
        C = torch.zeros(nsplit, stride, device=DEV)
        extra = {"fp32": 0, "bf16x3": L.GEMM_BF3, "fp16x2": L.GEMM_BF3 | L.GEMM_X2}[mode]
        kw = dict(a_amax=cells[0], b_amax=cells[1]) if mode == "fp16x2" else {}
        ops.gemm(dZd, Xd, C, n_out, n_in + 1, rows, n_out, n_in, ldc, flags=L.GEMM_SPLITK | extra, a_major=True,
                 b_major=True, ones_col=n_in, nsplit=nsplit, c_split_stride=stride, **kw)
        got = C[:, :n_out * ldc].view(nsplit, n_out, ldc)[:, :, :n_in + 1].double().sum(0).cpu()
        d = (got - ref).abs()
        out[mode] = (float(d.max() / ref.abs().max()), float((d.max(1).values / S.max(1).values).max()),
                     float(d[17].max() / ref.abs().max()))
    return out


@pytest.mark.parametrize("case", ["uniform", "few_rows_carry_it", "rows_x1e-8", "dead_channel"])
def test_weight_gradient_product_under_intra_tensor_dynamic_range(case):
    """Everything the backward outputs is a SUM over rows: rows far below the tensor's maximum contribute far below the
    sum's own rounding, whatever their relative accuracy — per tensor AND per output row fp16x2 stays at the bf16x3 /
    fp32 level.  The one structure it cannot follow is a whole output channel 2^30 below the others (`dead_channel`): that
    row of dW keeps its ABSOLUTE accuracy (2e-6 of the tensor's maximum, asserted) and loses its relative one (printed)."""
    e = measure_wgrad(case)
    print(f"\n[x2 trial, wgrad] {case}: " + "; ".join(f"{m}: tensor {e[m][0]:.1e} out-row {e[m][1]:.1e}" for m in MODES))
    for mode in MODES:
        assert e[mode][0] < 2e-6, (mode, e[mode])
        if case != "dead_channel" or mode != "fp16x2":
            assert e[mode][1] < 2e-6, (mode, e[mode])
    assert e["fp16x2"][2] < 2e-6                                  # the dead channel's row, in units of the tensor's maximum


# ---- model level: a TRAINED checkpoint -----------------------------------------------------------------------------
TRAIN_STEPS = 300


def _trained_model(shape: str, B: int, over: dict):
    """A GGNN fitted for TRAIN_STEPS FusedAdam steps (lr 1e-3: the weights really move) on one batch of B graphs."""
    sh = synthetic.SHAPES[shape]
    cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], **over)
    model = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda")))
    model.load_state_dict(O.init_params(cfg, seed=4))
    model = model.to(DEV).train()
    n8, e8, a8 = synthetic.make_batch(B, **sh, seed=0)
    nodes, edges, tgt = (torch.from_numpy(np.ascontiguousarray(x)).float().to(DEV) for x in (n8, e8, a8))
    opt = FusedAdam(model.parameters(), lr=1e-3)
    losses = []
    for step in range(TRAIN_STEPS):
        opt.zero_grad(set_to_none=True)
        loss = apd_kl_loss(model(nodes, edges), tgt)
        loss.backward()
        opt.step()
        if step % 50 == 0 or step == TRAIN_STEPS - 1:
            losses.append(float(loss))
    return model, cfg, (n8, e8, a8), (nodes, edges, tgt), losses


_CACHE = {}


def _checkpoint(shape, B, over):
    key = (shape, B, tuple(sorted(over.items())))
    if key not in _CACHE:
        model, cfg, host, dev, losses = _trained_model(shape, B, over)
        P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        P0 = O.init_params(cfg, seed=4)
        moved = max(float((P[k] - P0[k]).abs().max() / P0[k].abs().max().clamp_min(1e-12)) for k in P)
        _CACHE[key] = (model, cfg, host, dev, P, losses, moved)
    return _CACHE[key]


def _set_mode(lib, mode):
    lib.gi_bf3_enable(0 if mode == "fp32" else 1)
    lib.gi_x2_enable(1 if mode == "fp16x2" else 0)


def _live_only(n8, e8, a8):
    keep = np.nonzero(e8.reshape(e8.shape[0], -1).any(1))[0]               # graphs that are not fully masked
    return n8[keep], e8[keep], a8[keep]


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


#: What the measurement found (profiles/r05/x2_trial.txt, gpu_trained_checkpoint.log), on models fitted to a loss of
#: 0.17-0.23 (from 6.0 / 7.4), gradient tensors' maxima between 1e-10 and 1e-1:
#:   * On a fitted model d loss / d logits = softmax - target is a difference of nearly equal numbers: an error of 1e-6
#:     in a logit (every fp32 evaluation has one: HIP 1-2e-6 of the largest logit in all three modes, ATen's blocked CPU
#:     sums 1e-7) comes back multiplied by p / |p - t|.  Tensors fed by few, well-fitted outputs (the 1-output
#:     termination stack) showed 2e-4 with one checkpoint of this recipe and 1.7e-3 with another — the same in all three
#:     arithmetic modes, and a property of the loss, not of the backward: it depends on which weights 300 noisy steps
#:     happened to reach.  The test therefore separates the two things:
#:       (a) logits and loss against the fp32 and the fp64 oracle at 1e-4 as everywhere;
#:       (b) the BACKWARD OPERATOR: HIP's gradients against J^T v of the fp64 oracle for the SAME v (the d loss / d logits
#:           of the HIP logits), tensor by tensor, next to the fp32 oracle's own distance from fp64 for that v;
#:       (c) end to end (each evaluation's own logits -> own gradient): the gradient as a whole, reported per tensor.
#:   * (b), GDB-13 shape: gradient as a whole 1.0-1.1e-6 from fp64 in all three modes (the fp32 oracle: 0.7e-6); worst
#:     single tensor 5.0e-6 fp16x2 / 6.8e-6 bf16x3 / 1.1e-5 fp32 MFMA only (oracle 4-5e-6).  ZINC shape: as a whole
#:     1.6e-6 in both modes (oracle 1.0e-6); worst tensor 5.3e-6 fp16x2 / 6.6e-6 fp32 MFMA only (oracle 3.3e-6).  The
#:     backward operator is at the reference's own fp32 arithmetic on a fitted model, and fp16x2 is NOT behind the fp32
#:     MFMA anywhere.  Bars: the gradient as a whole within max(2e-5, 3 x the fp32 oracle's distance from fp64), every
#:     single tensor within 1e-4, and fp16x2's worst tensor no further from fp64 than 1.5 x the fp32-MFMA-only mode's
#:     (test_fp16x2_is_no_further_from_fp64_than_the_fp32_mfma).
#:   * (c): gradient as a whole 8e-6 (GDB-13) / 6e-5 (ZINC) from the fp64 oracle's own, worst tensor 3e-5 / 8e-5 with
#:     these checkpoints (the same in all modes; the fp32 oracle's own end-to-end distance is ~10x smaller because
#:     ATen's logits are).  Bar: the gradient as a whole within 1e-3.
_WORST = {}


@pytest.mark.parametrize("shape,B,over,mode", [
    ("gdb13", 1000, dict(hidden_node_features=128, message_size=128), "fp16x2"),
    ("gdb13", 1000, dict(hidden_node_features=128, message_size=128), "bf16x3"),
    ("gdb13", 1000, dict(hidden_node_features=128, message_size=128), "fp32"),
    ("zinc", 1000, {}, "fp16x2"),
    ("zinc", 1000, {}, "fp32"),
])
def test_trained_checkpoint_parity_in_every_arithmetic_mode(shape, B, over, mode):
    """Weights, activations and gradients of a FITTED model (300 Adam steps at lr 1e-3 on the bench batch: the loss
    halves, every weight tensor has moved) instead of an initialisation.  On the batch's live graphs: logits and loss
    1e-4 against the fp32 AND the fp64 oracle fed the same weights; every gradient tensor against the fp64 oracle's
    J^T v for the v the HIP backward was fed, on the SELU branches of the HIP forward (tests/pins.py), next to the fp32
    oracle's own distance from it; the gradient as a whole end to end — in the fp16x2 / bf16x3-only / fp32-MFMA-only
    modes; the dynamic-range guard stays silent on the trained model.  (Why two comparisons: the note above _WORST.)"""
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    model, cfg, host, _, P, losses, moved = _checkpoint(shape, B, over)
    assert np.isfinite(losses).all() and losses[-1] < 0.6 * losses[0] and moved > 0.05, (losses, moved)
    n8, e8, a8 = _live_only(*host)
    nodes, edges, tgt = (torch.from_numpy(np.ascontiguousarray(x)).float().to(DEV) for x in (n8, e8, a8))
    lib = L.load()
    was = lib.gi_bf3_enable(-1), lib.gi_x2_enable(-1)
    try:
        _set_mode(lib, mode)
        model.x2_guard_reset()
        params = list(model.parameters())
        out, tape = mpnn.ggnn_forward_raw(model.constants, nodes, edges, params, want_backward=True,
                                          guard=model._x2_guard_state(nodes.device))
        dims, graph, ws = tape
        signs = pins.signs_from_hip(dims, graph, ws, out, attn=False)
        g = pins.graph_arrays(graph)
        act_max = float(ops.ws_view(ws, dims, graph, "add1_act", graph.S + 1, j=1).abs().max())
        o_leaf = out.detach().clone().requires_grad_(True)
        loss = O.kl_loss(o_leaf, tgt)
        loss.backward()
        grads, _ = mpnn.ggnn_backward_raw(tape, out, o_leaf.grad, params)
        names = [k for k, _ in model.named_parameters()]
        t = lambda x, dt: torch.from_numpy(x).to(dt)
        v = o_leaf.grad.detach().cpu()                    # d loss / d logits at the HIP logits: what the HIP backward was fed
        o32, l32, g32, flipped, total = pins.oracle_pinned(O, P, cfg, t(n8, torch.float32), t(e8, torch.float32),
                                                           t(a8, torch.float32), signs, g, "GGNN", upstream=v)
        P64 = {k: v_.double() for k, v_ in P.items()}
        o64, l64, g64, _, _ = pins.oracle_pinned(O, P64, cfg, t(n8, torch.float64), t(e8, torch.float64),
                                                 t(a8, torch.float64), signs, g, "GGNN", upstream=v.double())
        _, _, g64_own, _, _ = pins.oracle_pinned(O, P64, cfg, t(n8, torch.float64), t(e8, torch.float64),
                                                 t(a8, torch.float64), signs, g, "GGNN")
        assert flipped < 1e-5 * total, (flipped, total)
        assert _rel(out, o32) < 1e-4 and _rel(out, o64) < 1e-4, (_rel(out, o32), _rel(out, o64))
        assert abs(float(loss) - float(l64)) < 1e-4 * abs(float(l64))
        # A tensor whose gradient has VANISHED on the fitted model (round 6: a checkpoint whose termination output sits on
        # SELU's floor has max |g| = 7e-16 for the whole fTermNet2 stack, 1e-13 of the largest gradient tensor) has no
        # meaningful relative error — 1.6e-3 "of" 7e-16 is rounding noise at 1e-18.  Denominators are floored at 1e-9 of
        # the largest gradient tensor's maximum: below that a gradient is zero for any optimizer step.
        floor = 1e-9 * max(float(g64[k].abs().max()) for k in names)
        relf = lambda a, b: float((torch.as_tensor(a).detach().double().cpu() - torch.as_tensor(b).detach().double().cpu()).abs().max()
                                  / max(float(torch.as_tensor(b).detach().double().abs().max()), floor))
        rows = [(k, relf(gr, g64[k]), relf(g32[k], g64[k]), relf(gr, g32[k])) for k, gr in zip(names, grads)]
        worst = max(rows, key=lambda r: r[1])
        worst_ref = max(rows, key=lambda r: r[2])
        over_bar = [r for r in rows if r[1] >= 1e-4]
        gmag = [float(g64[k].abs().max()) for k in names]
        stats = model.x2_guard_stats()
        print(f"\n[trained checkpoint, {shape}, {mode}] loss {losses[0]:.4f} -> {losses[-1]:.4f} over {TRAIN_STEPS} steps, "
              f"largest relative weight move {moved:.2f}, max hidden activation {act_max:.1f}, gradient tensors' maxima "
              f"{min(gmag):.1e} .. {max(gmag):.1e}; logits vs fp64 {_rel(out, o64):.1e}; worst gradient tensor vs the fp64 "
              f"oracle: HIP {worst[1]:.2e} ({worst[0]}; the fp32 oracle on it: {worst[2]:.2e}), fp32 oracle's own worst "
              f"{worst_ref[2]:.2e} ({worst_ref[0]}); {len(over_bar)} of {len(rows)} tensors above 1e-4; guard {stats}")
        _WORST[(shape, mode)] = (worst[1], worst_ref[2])
        den = sum(float(g64[k].pow(2).sum()) for k in names)
        l2_hip = (sum(float((gr.double().cpu() - g64[k]).pow(2).sum()) for k, gr in zip(names, grads)) / den) ** 0.5
        l2_ref = (sum(float((g32[k].double() - g64[k]).pow(2).sum()) for k in names) / den) ** 0.5
        den_own = sum(float(g64_own[k].pow(2).sum()) for k in names)
        l2_e2e = (sum(float((gr.double().cpu() - g64_own[k]).pow(2).sum()) for k, gr in zip(names, grads)) / den_own) ** 0.5
        e2e = max(((k, relf(gr, g64_own[k])) for k, gr in zip(names, grads)), key=lambda r: r[1])
        print(f"[trained checkpoint, {shape}, {mode}] the gradient as a whole, relative L2 distance from fp64: backward operator "
              f"(same d loss / d logits) HIP {l2_hip:.2e}, fp32 oracle {l2_ref:.2e}; end to end (the fp64 oracle's own logits "
              f"and gradient) HIP {l2_e2e:.2e}, its worst tensor {e2e[1]:.2e} ({e2e[0]})")
        assert l2_hip < max(2e-5, 3 * l2_ref), (l2_hip, l2_ref)
        assert l2_e2e < 1e-3, l2_e2e
        assert worst[1] < 1e-4, (worst, worst_ref)
        assert stats["forward_rows"] == 0 and stats["weight_lines"] == 0 and not stats["tripped"], stats
    finally:
        lib.gi_bf3_enable(was[0]); lib.gi_x2_enable(was[1])


def test_fp16x2_is_no_further_from_fp64_than_the_fp32_mfma():
    """Across the modes of the test above (same checkpoint, same batch): the fp16x2 mode's worst gradient tensor is not
    further from the fp64 oracle than 1.5 x the fp32-MFMA-only mode's (measured: 0.8-1.1 x)."""
    pairs = [(s, _WORST.get((s, "fp16x2")), _WORST.get((s, "fp32"))) for s in ("gdb13", "zinc")]
    pairs = [p for p in pairs if p[1] and p[2]]
    if not pairs:
        pytest.skip("runs after test_trained_checkpoint_parity_in_every_arithmetic_mode in the same process")
    for shape, x2, f32 in pairs:
        print(f"\n[trained checkpoint, {shape}] worst gradient tensor vs fp64: fp16x2 {x2[0]:.2e}, fp32 MFMA {f32[0]:.2e}, "
              f"fp32 oracle {f32[1]:.2e}")
        assert x2[0] < 1.5 * f32[0] + 1e-5, (shape, x2, f32)


@pytest.mark.parametrize("wname", ["APDReadout.fAddNet1.seq.3.weight",      # a 500 x 500 hidden layer of a node-level stack (GEMM launches)
                                   "msg_nns.0.seq.3.weight"])               # a 250 x 250 hidden layer of a message stack (chain launches)
def test_guard_trips_the_model_into_bf16x3_and_reports_it(wname):
    """End to end: a weight matrix of a fp16x2 layer with three rows 2^-30 below its maximum -> gi_x2_weight_guard counts
    them during the forward and sets the host flag; the NEXT forward runs with GI_RUN_NO_X2 and agrees with the
    process-wide bf16x3 mode bit for bit; x2_guard_reset() re-arms.  Both kinds of fp16x2 forward launches: the GEMMs of
    the node-level stacks and (since the forward chains run as fp16x2) the message stacks' chain launches."""
    sh = synthetic.SHAPES["gdb13"]
    cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], hidden_node_features=128,
                          message_size=128)
    model = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda")))
    model.load_state_dict(O.init_params(cfg, seed=4))
    model = model.to(DEV).eval()
    n8, e8, _ = synthetic.make_batch(1000, **sh, seed=0)
    nodes, edges = (torch.from_numpy(np.ascontiguousarray(x)).float().to(DEV) for x in (n8, e8))
    lib = L.load()
    with torch.no_grad():
        model.cache_pass0 = False
        ref_x2 = model(nodes, edges).clone()
        assert model.x2_guard_stats() == {"forward_rows": 0, "weight_lines": 0, "dgrad_rows": 0, "tripped": False}
        w = dict(model.named_parameters())[wname]
        w[[1, 2, 3]] *= 2.0 ** -30
        first = model(nodes, edges).clone()                     # still fp16x2; the guard notices
        torch.cuda.synchronize()
        st = model.x2_guard_stats()
        assert st["weight_lines"] == 3 and st["tripped"], st
        second = model(nodes, edges).clone()                    # bf16x3 from here on
        was = lib.gi_x2_enable(0)
        try:
            model.x2_guard = False
            plain_b3 = model(nodes, edges).clone()
        finally:
            lib.gi_x2_enable(was)
            model.x2_guard = True
        assert torch.equal(second, plain_b3)
        assert float((first - second).abs().max()) <= 1e-4 * float(second.abs().max())
        w[[1, 2, 3]] *= 2.0 ** 30
        model.x2_guard_reset()
        again = model(nodes, edges)
        assert torch.equal(again, ref_x2) and not model.x2_guard_stats()["tripped"]


def test_sync_free_forward_runs_under_the_same_guard_and_trip_state():
    """Round-5 advisor: the host-sync-free forward called plain gi_ggnn_forward — no guard, no GI_RUN_NO_X2 after a trip —
    so blocking forwards ran bf16x3 while sync-free / generation forwards kept running unguarded fp16x2, and the pass-0
    row cache could serve rows of the other arithmetic.  Now: (a) a planted weight trips the guard FROM a sync-free
    forward, (b) after the trip blocking and sync-free forwards agree bit for bit (both bf16x3), with the row cache on,
    (c) reset re-arms both."""
    sh = synthetic.SHAPES["gdb13"]
    cfg = O.shaped_config(sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], hidden_node_features=128,
                          message_size=128)
    model = mpnn.GGNN(O.as_constants(dict(cfg, device="cuda")))
    model.load_state_dict(O.init_params(cfg, seed=4))
    model = model.to(DEV).eval()
    n8, e8, _ = synthetic.make_batch(600, **sh, seed=3)
    nodes, edges = (torch.from_numpy(np.ascontiguousarray(x)).float().to(DEV) for x in (n8, e8))
    with torch.no_grad():
        model.sync_free = False
        ref_blocking = model(nodes, edges).clone()
        model.sync_free = True
        assert torch.equal(model(nodes, edges), ref_blocking)                 # fp16x2, both paths, cache on
        w = dict(model.named_parameters())["APDReadout.fAddNet1.seq.3.weight"]
        w[[1, 2, 3]] *= 2.0 ** -30
        model.reset_pass0_cache()                                            # (p.data-style write: versions did move here, belt and braces)
        model(nodes, edges)                                                  # sync-free, still fp16x2: the guard notices
        torch.cuda.synchronize()
        st = model.x2_guard_stats()
        assert st["weight_lines"] == 3 and st["tripped"], st
        sf = model(nodes, edges).clone()                                     # bf16x3 now, sync-free
        model.sync_free = False
        bl = model(nodes, edges).clone()                                     # bf16x3, blocking
        assert torch.equal(sf, bl)
        model.last_bounded_error()
        w[[1, 2, 3]] *= 2.0 ** 30
        model.x2_guard_reset()
        model.sync_free = True
        assert torch.equal(model(nodes, edges), ref_blocking) and not model.x2_guard_stats()["tripped"]


if __name__ == "__main__":
    print("forward layout, Y = A W^T, M x N x K = %d x %d x %d; errors against the fp64 product" % (M, N, K))
    print("%-30s %-8s %10s %10s %10s" % ("case", "mode", "tensor", "row", "column"))
    for case in CASES:
        e = measure(case)
        for mode in MODES:
            print("%-30s %-8s %10.1e %10.1e %10.1e" % (case, mode, *e[mode]))
    print("\nweight-gradient layout, [dW | db] = dZ^T [X | 1], 7258 rows, 500 x 501")
    print("%-30s %-8s %10s %10s %14s" % ("case", "mode", "tensor", "out-row", "row 17/max|ref|"))
    for case in ("uniform", "few_rows_carry_it", "rows_x1e-8", "dead_channel"):
        e = measure_wgrad(case)
        for mode in MODES:
            print("%-30s %-8s %10.1e %10.1e %14.1e" % (case, mode, *e[mode]))
