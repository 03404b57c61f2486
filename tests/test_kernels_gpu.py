"""-m gpu: every building-block kernel, called through the C ABI, against its CPU reference
(tests/ref_dataflow.py).  Integer work is compared bit-exactly; fp32 GEMMs within 2e-5 relative of
an fp64 CPU product (exact-fp32 MFMA = fmaf chain, so only the summation order differs)."""
import os

import numpy as np
import pytest
import torch

from graphinvent_amd import lib as L
from graphinvent_amd import ops, synthetic
from tests import ref_dataflow as D
from tests.golden.spec import tiny_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


# ------------------------------------------------------------------------------------------------
def _compact_case(n8, e8, H=16, nodedup=False):
    ref = D.compact(n8, e8, nodedup=nodedup)
    g, hx0 = ops.compact(torch.from_numpy(n8).float().to(DEV), torch.from_numpy(e8).float().to(DEV), H,
                         class_csr=True, nodedup=nodedup)
    if nodedup:      # every slot its own row, every edge its own message row, no pass-0 rows
        assert (g.S, g.U, g.D0) == (n8.shape[0] * n8.shape[1], g.E, 0)
        assert np.array_equal(g.mu_off.cpu().numpy(), np.arange(g.E + 1))
    assert (g.S, g.E, g.U) == (ref["S"], ref["E"], ref["U"])
    assert list(g.Ut) == np.diff(ref["type_off"]).tolist()
    for name in ("cidx", "slot_of", "u_src", "in_perm", "mu_off", "mu_dst", "mu_slot", "out_perm",
                 "type_off"):
        got = getattr(g, name).cpu().numpy()
        assert np.array_equal(got, ref[name]), name            # integer work: bit-exact
    for name in ("seg_off", "src_off"):                        # [R + 1] = [S + 2] entries
        assert np.array_equal(getattr(g, name).cpu().numpy(), ref[name]), name
    # pass-0 rows: (feature class, bond type) pairs, their gather rows and the edge-count matrix
    assert g.D0 == ref["D0"]
    if g.D0:
        assert np.array_equal(g.type_off0.cpu().numpy(), ref["type_off0"])
        assert np.array_equal(g.d_src.cpu().numpy(), ref["d_src"])
        assert np.array_equal(g.cmat.cpu().numpy()[:, :g.D0], ref["cmat"])
        assert float(g.cmat.sum()) == g.E
        # AttentionGGNN's pass 0: edge slot -> pass-0 row, and the row -> edge slots CSR (stable order)
        assert g.class_csr
        for name in ("e2d", "cls_off", "cls_edges"):
            assert np.array_equal(getattr(g, name).cpu().numpy(), ref[name]), name
    assert np.array_equal(g.node_mask.cpu().numpy(), ref["node_mask"].astype(np.int32))
    B, N, Fn = n8.shape
    x = np.zeros((g.S + 1, hx0.shape[1]), dtype=np.float32)
    rows = n8.reshape(B * N, Fn)[ref["slot_of"]].astype(np.float32)
    x[:g.S, :Fn] = rows
    x[:g.S, H:H + Fn] = rows
    assert np.array_equal(hx0.cpu().numpy(), x)


def test_compact_tiny_edge_cases():
    n8, e8, _ = tiny_inputs()
    _compact_case(n8, e8)
    n8[4] = 0; e8[4] = 0; n8[4, 0, 0] = 1; e8[4, 0, 5, 1] = 1          # asymmetric, inactive neighbour
    _compact_case(n8, e8)


def test_compact_without_row_sharing():
    """gi_compact_count_ex(nodedup): the layout of AlphaDropout's training mode."""
    n8, e8, _ = tiny_inputs()
    _compact_case(n8, e8, nodedup=True)
    e0 = e8.copy(); e0[:] = 0
    _compact_case(n8, e0, nodedup=True)
    for shape, B in (("gdb13", 200), ("chembl", 16)):
        a, b, _ = synthetic.make_batch(B, **synthetic.SHAPES[shape], seed=7)
        _compact_case(a, b, H=100, nodedup=True)


def test_compact_no_edges_at_all():
    n8, e8, _ = tiny_inputs()
    e8[:] = 0
    _compact_case(n8, e8)


def test_compact_fixture_and_shapes(golden_dir):
    d = np.load(os.path.join(golden_dir, "gdb13_1K-debug_train.npz"))
    _compact_case(d["nodes"], d["edges"], H=100)
    for shape, B in (("zinc", 300), ("chembl", 64)):
        n8, e8, _ = synthetic.make_batch(B, **synthetic.SHAPES[shape], seed=5)
        _compact_case(n8, e8, H=100)


def test_compact_pass0_shortcut_is_switched_off_for_non_binary_features():
    n8, e8, _ = tiny_inputs()
    nodes = torch.from_numpy(n8).float().to(DEV)
    nodes[1, 0, 0] = 0.5                                       # a real-valued feature
    g, _ = ops.compact(nodes, torch.from_numpy(e8).float().to(DEV), 16)
    assert g.D0 == 0 and g.cmat is None


def multi_bond_inputs(n8, e8):
    """What GraphGenerator.build_graphs feeds the model in slot 0 (GraphGenerator.py:133, 424-427): all-ones feature
    rows, a self-loop, atom pairs with several bond types set at once — plus one such pair in another graph."""
    n8, e8 = n8.copy(), e8.copy()
    n8[0] = 1
    e8[0] = 0
    e8[0, 0, 0, 0] = 1
    e8[0, 0, 3, :] = [1, 0, 1]; e8[0, 3, 0, :] = [1, 0, 1]
    e8[0, 2, 0, :] = 1
    b = int(np.argmax(e8.reshape(e8.shape[0], -1).sum(1) * (np.arange(e8.shape[0]) > 0)))
    i, j = [int(x[0]) for x in np.nonzero(e8[b].sum(2))]
    e8[b, i, j, :] = 1; e8[b, j, i, :] = 1
    return n8, e8


def test_compact_pairs_with_several_bond_types_are_parallel_edges():
    """Every set bond-type entry is an edge of its own (what the reference's masked sum over the bond types computes,
    gnn/mpnn.py:286-294): the index arrays of such inputs, bit-exact against the numpy model; values other than 0 / 1
    are still refused."""
    n8, e8, _ = tiny_inputs()
    _compact_case(*multi_bond_inputs(n8, e8))
    _compact_case(*multi_bond_inputs(n8, e8), nodedup=True)
    for shape, B in (("gdb13", 100), ("chembl", 12)):
        a, b, _ = synthetic.make_batch(B, **synthetic.SHAPES[shape], seed=9)
        rng = np.random.default_rng(1)
        for _ in range(3 * B):                               # random extra bond types on existing bonds
            g = rng.integers(0, B)
            ii, jj = np.nonzero(b[g].sum(2))
            if len(ii):
                k = rng.integers(0, len(ii))
                b[g, ii[k], jj[k], rng.integers(0, b.shape[3])] = 1
        _compact_case(a, b, H=100)
    e8[5, 0, 1, 0] = 2
    with pytest.raises(ValueError, match="0 or 1"):
        ops.compact(torch.from_numpy(n8).float().to(DEV), torch.from_numpy(e8).float().to(DEV), 16)


# ------------------------------------------------------------------------------------------------
TILES = [(1, 1), (1, 2), (2, 2)]


@pytest.fixture
def chain_cfg():
    """gi_mlp_chain_config for one test (row-block height, 64-row variant, weight ring), reset afterwards."""
    lib = L.load()

    def set_cfg(tile_rows=0, rows64=-1, ring=2):
        L.check(lib.gi_mlp_chain_config(tile_rows, rows64, ring, None), "gi_mlp_chain_config")
    yield set_cfg
    lib.gi_mlp_chain_config(0, -1, 2, None)


@pytest.fixture(params=[0, 3], ids=["wg-per-tile", "tile-loop-3wgs"])
def gemm_grid(request):
    """Every gi_gemm launch is a tile loop; launches with more tiles than the device holds workgroups run as a
    persistent grid.  The unit-test problems are far too small for that, so each GEMM test also runs with the
    grid capped at 3 workgroups (gi_gemm_config): every workgroup then walks many tiles — of different
    problems / groups / split-K slabs — through the prefetch-under-epilogue path."""
    lib = L.load()
    lib.gi_gemm_config(-1, request.param)
    yield request.param
    lib.gi_gemm_config(-1, 0)


@pytest.mark.parametrize("tm,tn", TILES)
@pytest.mark.parametrize("M,N,K", [(1000, 250, 100), (333, 500, 685), (130, 45, 500), (64, 1, 500),
                                   (300, 128, 136), (5, 3, 7)])
def test_gemm_forward_bias_selu(M, N, K, tm, tn, gemm_grid):
    g = torch.Generator().manual_seed(M + N + K)
    lda = ops.r4(K) + 4
    X = torch.randn(M, lda, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ldc = ops.r4(N) + 8
    Y = torch.full((M, ldc), 7.0, device=DEV)
    ops.gemm(X.to(DEV), W.to(DEV), Y, M, N, K, lda, K, ldc, flags=L.EPI_BIAS | L.EPI_SELU,
             bias=b.to(DEV), tm=tm, tn=tn)
    ref = D.selu(X[:, :K].double() @ W.double().t() + b.double())
    assert rel(Y[:, :N], ref) < 2e-5
    assert bool((Y[:, N:] == 7.0).all())                # nothing written outside [M, N]


@pytest.mark.parametrize("M,N,K", [(7258, 500, 500), (1000, 250, 250), (130, 200, 685), (257, 192, 36), (64, 129, 1000)])
def test_gemm_bf16x3_split_matches_fp64_like_the_fp32_mfma_does(M, N, K):
    """GI_GEMM_BF3: fp32 operands split into three bf16 each, six bf16 MFMA products per fp32 product, fp32
    accumulate — `MLP.forward`'s Linear + SELU (gnn/modules.py:166-170) and its dgrad on the bf16 pipe.  The result
    must be as close to the fp64 product as the fp32 MFMA chain's (both ~1e-7 of sum |a||b|), far inside the path's
    1e-4 bar; rows / columns / reduction lengths off the 128 x 128 x 32 tile grid, row gather, accumulate."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    lda = ops.r4(K) + 4
    X = torch.randn(M, lda, generator=g)
    X[:, K:] = float("nan")                               # padding beyond K must never reach a product
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ldc = ops.r4(N) + 8
    Xd, Wd = X.to(DEV), W.to(DEV)
    img = ops.bf3_pack(Wd)
    Y3 = torch.full((M, ldc), 7.0, device=DEV)
    Y1 = torch.full((M, ldc), 7.0, device=DEV)
    ops.gemm(Xd, img, Y3, M, N, K, lda, 0, ldc, flags=L.EPI_BIAS | L.EPI_SELU | L.GEMM_BF3, bias=b.to(DEV))
    Xc = Xd.clone(); Xc[:, K:] = 0
    ops.gemm(Xc, Wd, Y1, M, N, K, lda, K, ldc, flags=L.EPI_BIAS | L.EPI_SELU, bias=b.to(DEV))
    ref = D.selu(X[:, :K].double() @ W.double().t() + b.double())
    e3, e1 = rel(Y3[:, :N], ref), rel(Y1[:, :N], ref)
    assert e3 < 2e-6 and e3 < 4 * e1 + 1e-7, (e3, e1)
    assert bool((Y3[:, N:] == 7.0).all())                # nothing written outside [M, N]
    # the other operand forms compute the same planes, hence the same bits: B as the plain fp32 weight (split while
    # staged: what the model's forward launches use) and A as a pre-split image
    Yf = torch.full((M, ldc), 7.0, device=DEV)
    ops.gemm(Xd, Wd, Yf, M, N, K, lda, K, ldc, flags=L.EPI_BIAS | L.EPI_SELU | L.GEMM_BF3 | L.GEMM_BF3B_F32,
             bias=b.to(DEV))
    assert torch.equal(Yf, Y3)
    Ya = torch.full((M, ldc), 7.0, device=DEV)
    ops.gemm(ops.bf3_pack(Xc[:, :K]), img, Ya, M, N, K, 0, 0, ldc,
             flags=L.EPI_BIAS | L.EPI_SELU | L.GEMM_BF3 | L.GEMM_BF3A, bias=b.to(DEV))
    assert torch.equal(Ya, Y3)
    # dgrad: dX = (dZ . W) * selu'(act), W [K_out = K, N_in = N2] through the transposed image; accumulate
    N2 = N
    Wt = torch.randn(K, N2, generator=g) / K ** 0.5       # [out, in]
    act = torch.randn(M, ops.r4(N2), generator=g)
    dX0 = torch.randn(M, ops.r4(N2), generator=g)
    imgT = ops.bf3_pack(Wt.to(DEV), transpose=True)
    dX = dX0.clone().to(DEV)
    ops.gemm(Xc, imgT, dX, M, N2, K, lda, 0, ops.r4(N2), flags=L.EPI_DSELU | L.EPI_ACCUM | L.GEMM_BF3,
             act=act.to(DEV), ldact=ops.r4(N2))
    y = act[:, :N2].double()
    grad = D.selu_grad_from_out(y)
    refd = (X[:, :K].double() @ Wt.double()) * grad + dX0[:, :N2].double()
    assert rel(dX[:, :N2], refd) < 2e-6
    # the model's dgrad launches: W^T as a plain fp32 copy, split while staged — the same bits
    dXf = dX0.clone().to(DEV)
    ops.gemm(Xc, ops.bf3_pack(Wt.to(DEV), transpose=True, as_f32=True), dXf, M, N2, K, lda, ops.r4(K), ops.r4(N2),
             flags=L.EPI_DSELU | L.EPI_ACCUM | L.GEMM_BF3 | L.GEMM_BF3B_F32, act=act.to(DEV), ldact=ops.r4(N2))
    assert torch.equal(dXf, dX)
    # a row gather is refused on the bf16 pipe (32-bit operand offsets cannot be bounded for gathered rows)
    idx = torch.randperm(M, generator=g).int()
    with pytest.raises(RuntimeError, match="GI_EINVAL"):
        ops.gemm(Xc, imgT, dX, M, N2, K, lda, 0, ops.r4(N2), flags=L.EPI_DSELU | L.GEMM_BF3, act=act.to(DEV),
                 ldact=ops.r4(N2), a_idx=idx.to(DEV))
    # a launch mixes bf16x3 problems with fp32 ones: refused, not silently computed in one precision
    p = (L.GemmParams * 2)()
    for q in p:
        q.A, q.B, q.C, q.M, q.N, q.K, q.lda, q.ldb, q.ldc = Xc.data_ptr(), Wd.data_ptr(), Y1.data_ptr(), M, N, K, lda, K, ldc
        q.nsplit, q.ones_col, q.tm, q.tn = 1, -1, 1, 1
    p[0].B, p[0].flags = img.data_ptr(), L.GEMM_BF3
    assert L.load().gi_gemm_batch(p, 2, None) == -1         # GI_EINVAL


@pytest.fixture
def bf3_kernel_switches():
    lib = L.load()
    prev = lib.gi_b3p_enable(-1), lib.gi_b3v_enable(-1)
    yield lib
    lib.gi_b3p_enable(prev[0]); lib.gi_b3v_enable(prev[1])
    os.environ.pop("GI_B3P_ALL", None)


@pytest.mark.parametrize("kernel", ["b3p", "b3v", "r3"])
@pytest.mark.parametrize("M,N,K", [(7258, 500, 500), (1000, 250, 250), (300, 200, 685), (257, 192, 36), (129, 257, 1000)])
def test_gemm_bf16x3_kernels_forward_and_dgrad_layouts_vs_fp64(M, N, K, kernel, bf3_kernel_switches):
    """Every bf16x3 kernel — the software-pipelined 128 x 256 one (gi_gemm_b3p.hip), the 32-deep-tile 128 x 128 one
    (gi_gemm_b3v.hip), the round-3 kernel — on the forward layout (bias + SELU) and on the dgrad layouts (W^T as a
    contiguous copy; W AS STORED, reduction-major, transposed on the way into LDS) against the fp64 product:
    max |d| / max |ref| < 2e-6, edges off every tile grid, nothing written outside [M, N]."""
    lib = bf3_kernel_switches
    lib.gi_b3p_enable(1 if kernel == "b3p" else 0)
    lib.gi_b3v_enable(1 if kernel == "b3v" else 0)
    os.environ["GI_B3P_ALL"] = "1"                      # (the pipelined kernel also for launches of few tiles)
    g = torch.Generator().manual_seed(M + 7 * N + K)
    lda, ldc = ops.r4(K) + 4, ops.r4(N) + 8
    X = torch.randn(M, lda, generator=g); X[:, K:] = float("nan")
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    Xd, Wd = X.to(DEV), W.to(DEV)
    Y = torch.full((M, ldc), 7.0, device=DEV)
    ops.gemm(Xd, Wd, Y, M, N, K, lda, K, ldc, flags=L.EPI_BIAS | L.EPI_SELU | L.GEMM_BF3 | L.GEMM_BF3B_F32, bias=b.to(DEV))
    ref = D.selu(X[:, :K].double() @ W.double().t() + b.double())
    assert rel(Y[:, :N], ref) < 2e-6
    assert bool((Y[:, N:] == 7.0).all())
    # dgrad: dX[M, N] = (dZ[M, K] . Wt[K, N]) * selu'(act) (+ accumulate)
    Wt = torch.randn(K, N, generator=g) / K ** 0.5
    act = torch.randn(M, ops.r4(N), generator=g)
    dX0 = torch.randn(M, ops.r4(N), generator=g)
    refd = (X[:, :K].double() @ Wt.double()) * D.selu_grad_from_out(act[:, :N].double()) + dX0[:, :N].double()
    dXc = dX0.clone().to(DEV)
    ops.gemm(Xd, ops.bf3_pack(Wt.to(DEV), transpose=True, as_f32=True), dXc, M, N, K, lda, ops.r4(K), ops.r4(N),
             flags=L.EPI_DSELU | L.EPI_ACCUM | L.GEMM_BF3 | L.GEMM_BF3B_F32, act=act.to(DEV), ldact=ops.r4(N))
    assert rel(dXc[:, :N], refd) < 2e-6
    if kernel != "r3":                                  # W as stored: only the round-4 kernels transpose while staging
        dXm = dX0.clone().to(DEV)
        ldw = N + (N & 1) if kernel == "b3v" else N     # (b3v reads column PAIRS of a major operand: even row pitch)
        Wp = torch.zeros(K, ldw); Wp[:, :N] = Wt
        ops.gemm(Xd, Wp.to(DEV), dXm, M, N, K, lda, ldw, ops.r4(N), flags=L.EPI_DSELU | L.EPI_ACCUM | L.GEMM_BF3,
                 act=act.to(DEV), ldact=ops.r4(N), b_major=True)
        assert rel(dXm[:, :N], refd) < 2e-6


@pytest.mark.parametrize("kernel", ["b3p", "b3v"])
@pytest.mark.parametrize("rows,n_out,n_in,nsplit", [(7258, 500, 500, 8), (2600, 250, 250, 3), (1000, 192, 301, 1),
                                                    (517, 257, 129, 5), (40, 130, 200, 2)])
def test_gemm_bf16x3_weight_gradient_layout_vs_fp64(rows, n_out, n_in, nsplit, kernel, bf3_kernel_switches):
    """The A_MAJOR, B_MAJOR = true, true layout on the bf16 pipe: [dW | db] = dZ^T [X | 1] (autograd of
    gnn/modules.py:166-170), both operands read along rows and transposed in registers on the way into LDS, the
    ones column for the bias gradient, split-K slabs summed afterwards — against the fp64 product (< 2e-6 of
    max |ref|, the slabs' sum; the fp32 MFMA kernel's own distance is printed beside it), slab tails beyond the
    reduction range exactly zero, nothing written outside [n_out, n_in + 1]."""
    lib = bf3_kernel_switches
    lib.gi_b3p_enable(1 if kernel == "b3p" else 0)
    lib.gi_b3v_enable(1 if kernel == "b3v" else 0)
    g = torch.Generator().manual_seed(rows + n_out + n_in)
    lddz, ldx, ldc = ops.r4(n_out) + 4, ops.r4(n_in), ops.r4(n_in + 1) + 4
    dZ = torch.randn(rows, lddz, generator=g) * 1e-3; dZ[:, n_out:] = float("nan")
    X = torch.randn(rows, ldx, generator=g)
    stride = ops.r4(n_out * ldc)
    ref = torch.cat([dZ[:, :n_out].double().t() @ X[:, :n_in].double(), dZ[:, :n_out].double().sum(0)[:, None]], 1)
    outs = {}
    for name, extra in (("bf16x3", L.GEMM_BF3), ("fp32", 0)):
        C = torch.full((nsplit, stride), 7.0, device=DEV)
        ops.gemm(dZ.to(DEV), X.to(DEV), C, n_out, n_in + 1, rows, lddz, ldx, ldc, flags=L.GEMM_SPLITK | extra,
                 a_major=True, b_major=True, ones_col=n_in, nsplit=nsplit, c_split_stride=stride)
        S = C[:, :n_out * ldc].view(nsplit, n_out, ldc)
        assert bool((S[:, :, n_in + 1:] == 7.0).all())
        outs[name] = S[:, :, :n_in + 1].double().sum(0).cpu()
    e3, e1 = rel(outs["bf16x3"], ref), rel(outs["fp32"], ref)
    assert e3 < 2e-6 and e3 < 4 * e1 + 2e-7, (e3, e1)


@pytest.mark.parametrize("kernel", ["b3p", "r3"])
@pytest.mark.parametrize("M,N,K,big", [(7258, 500, 500, 1.0), (1000, 250, 250, 3e4), (257, 192, 36, 1e-6), (129, 257, 1000, 1.0)])
def test_gemm_fp16x2_split_vs_fp64_with_amax_cells_from_the_producers(M, N, K, big, kernel, bf3_kernel_switches):
    """GI_GEMM_X2 (csrc/gi_x2.h): two scaled fp16 planes per operand, three f16 MFMA products.  The scales come from the
    operands' amax cells — max |W| from gi_absmax, max |X| from the c_amax of the launch that PRODUCED X (an fp32 GEMM
    here) — and the result is as close to the fp64 product as the bf16x3 split (< 2e-6 of max |ref|) at every
    magnitude of the tensors (`big` scales X: the per-tensor power-of-two scale keeps fp16's 5-bit exponent in range).
    Forward layout, dgrad layout (W^T copy, and W as stored for the pipelined kernel), and the weight-gradient
    layout of the pipelined kernel; the launch's own c_amax cell == max |stored|."""
    lib = bf3_kernel_switches
    lib.gi_b3p_enable(1 if kernel == "b3p" else 0); lib.gi_b3v_enable(0)
    os.environ["GI_B3P_ALL"] = "1"
    g = torch.Generator().manual_seed(M + 3 * N + K)
    lda, ldc = ops.r4(K), ops.r4(N) + 8
    X0 = torch.randn(M, 64, generator=g) * big
    P = torch.randn(K, 64, generator=g) / 8
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * big
    cells = torch.zeros(4, L.AMAX_WORDS, device=DEV)
    Xd = torch.zeros(M, lda, device=DEV)
    ops.gemm(X0.to(DEV), P.to(DEV), Xd, M, K, 64, 64, 64, lda, flags=0, c_amax=cells[0])     # the producer of X
    X = Xd.cpu()
    assert float(cells[0].max()) == float(X[:, :K].abs().max())
    Wd = W.to(DEV)
    ops.absmax([Wd], cells[1:2])
    assert float(cells[1].max()) == float(W.abs().max())
    Y = torch.full((M, ldc), 7.0, device=DEV)
    F = L.GEMM_BF3 | L.GEMM_BF3B_F32 | L.GEMM_X2
    # (tiny magnitudes without the SELU: its fp32 exp(x) - 1 — ATen's form, gi_common.h — cancels at |x| ~ 1e-6)
    selu = big >= 1.0
    ops.gemm(Xd, Wd, Y, M, N, K, lda, K, ldc, flags=L.EPI_BIAS | (L.EPI_SELU if selu else 0) | F, bias=b.to(DEV),
             a_amax=cells[0], b_amax=cells[1], c_amax=cells[2])
    ref = X[:, :K].double() @ W.double().t() + b.double()
    ref = D.selu(ref) if selu else ref
    assert rel(Y[:, :N], ref) < 2e-6
    assert bool((Y[:, N:] == 7.0).all())
    assert float(cells[2].max()) == float(Y[:, :N].abs().max())
    # dgrad layouts
    Wt = torch.randn(K, N, generator=g) / K ** 0.5
    act = torch.randn(M, ops.r4(N), generator=g)
    refd = (X[:, :K].double() @ Wt.double()) * D.selu_grad_from_out(act[:, :N].double())
    ops.absmax([Wt.to(DEV)], cells[3:4])
    dXc = torch.zeros(M, ops.r4(N), device=DEV)
    ops.gemm(Xd, ops.bf3_pack(Wt.to(DEV), transpose=True, as_f32=True), dXc, M, N, K, lda, ops.r4(K), ops.r4(N),
             flags=L.EPI_DSELU | F, act=act.to(DEV), ldact=ops.r4(N), a_amax=cells[0], b_amax=cells[3])
    assert rel(dXc[:, :N], refd) < 2e-6
    if kernel == "b3p":
        dXm = torch.zeros(M, ops.r4(N), device=DEV)
        ops.gemm(Xd, Wt.to(DEV), dXm, M, N, K, lda, N, ops.r4(N), flags=L.EPI_DSELU | L.GEMM_BF3 | L.GEMM_X2,
                 act=act.to(DEV), ldact=ops.r4(N), b_major=True, a_amax=cells[0], b_amax=cells[3])
        assert rel(dXm[:, :N], refd) < 2e-6
        # weight-gradient layout: [dW | db] = dZ^T [X | 1], dZ = act here (its amax by gi_absmax)
        ops.absmax([act.to(DEV)], cells[3:4])
        nsplit, ldw = 3, ops.r4(K + 1) + 4
        stride = ops.r4(N * ldw)
        C = torch.full((nsplit, stride), 7.0, device=DEV)
        ops.gemm(act.to(DEV), Xd, C, N, K + 1, M, ops.r4(N), lda, ldw, flags=L.GEMM_SPLITK | L.GEMM_BF3 | L.GEMM_X2,
                 a_major=True, b_major=True, ones_col=K, nsplit=nsplit, c_split_stride=stride,
                 a_amax=cells[3], b_amax=cells[0])
        S = C[:, :N * ldw].view(nsplit, N, ldw)
        refw = torch.cat([act[:, :N].double().t() @ X[:, :K].double(), act[:, :N].double().sum(0)[:, None]], 1)
        assert bool((S[:, :, K + 1:] == 7.0).all())
        assert rel(S[:, :, :K + 1].double().sum(0).cpu(), refw) < 2e-6


@pytest.mark.parametrize("route", ["fp32", "bf3_r3", "bf3_b3v", "bf3_b3p"])
@pytest.mark.parametrize("layout", ["forward", "dgrad_wt_copy", "dgrad_w_as_stored"])
def test_every_gemm_route_publishes_c_amax(route, layout, bf3_kernel_switches):
    """gi_gemm_params.c_amax: whichever kernel a launch is routed to, the cell holds max |stored value| afterwards — the
    fp16x2 launch that reads the tensor next takes its scale from it, and a cell left at zero reads as scale 1 (fp16
    planes of raw values).  Round 6 found the bf16x3 kernel of gi_gemm_b3v.hip not publishing: in a model whose
    node-level stacks END in a layer >= 192 wide (A = 432, tests/test_dims_gpu.py) that kernel produces the dZ the
    fp16x2 layers below read, and every gradient upstream was ~5e-4 off."""
    lib = bf3_kernel_switches
    lib.gi_b3p_enable(1 if route == "bf3_b3p" else 0)
    lib.gi_b3v_enable(1 if route == "bf3_b3v" else 0)
    if route == "bf3_b3p": os.environ["GI_B3P_ALL"] = "1"
    M, N, K = 700, 250, 300
    g = torch.Generator().manual_seed(5)
    X = (torch.randn(M, ops.r4(K), generator=g) * 1e-3).to(DEV)
    cell = torch.zeros(1, L.AMAX_WORDS, device=DEV)
    Y = torch.zeros(M, ops.r4(N), device=DEV)
    bf3 = 0 if route == "fp32" else L.GEMM_BF3
    if layout == "forward":
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
        ops.gemm(X, W, Y, M, N, K, ops.r4(K), K, ops.r4(N), flags=L.EPI_BIAS | L.EPI_SELU | bf3 | (L.GEMM_BF3B_F32 if bf3 else 0),
                 bias=torch.randn(N, generator=g).to(DEV), c_amax=cell[0])
    else:
        Wt = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
        act = torch.randn(M, ops.r4(N), generator=g).to(DEV)
        if layout == "dgrad_wt_copy" and bf3:
            ops.gemm(X, ops.bf3_pack(Wt, transpose=True, as_f32=True), Y, M, N, K, ops.r4(K), ops.r4(K), ops.r4(N),
                     flags=L.EPI_DSELU | bf3 | L.GEMM_BF3B_F32, act=act, ldact=ops.r4(N), c_amax=cell[0])
        else:
            if bf3 and route != "bf3_b3p" and route != "bf3_b3v":
                pytest.skip("W as stored on the 16-bit pipe: the 32-deep-tile kernels only")
            ops.gemm(X, Wt, Y, M, N, K, ops.r4(K), N, ops.r4(N), flags=L.EPI_DSELU | bf3, act=act, ldact=ops.r4(N),
                     b_major=True, c_amax=cell[0])
    assert float(Y[:, :N].abs().max()) > 0
    assert float(cell.max()) == float(Y[:, :N].abs().max())


@pytest.mark.parametrize("kernel", ["b3p", "t128"])
@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("rows,n_out,n_in,nsplit", [(7111, 250, 250, 16), (1300, 384, 128, 3), (517, 257, 129, 5), (600, 128, 250, 2),
                                                    (40, 130, 200, 2)])
def test_gemm_fp16x2_weight_gradient_layout_both_kernels_gathered_and_grouped(rows, n_out, n_in, nsplit, gather, kernel,
                                                                              bf3_kernel_switches):
    """[dW | db] = dZ^T [X | 1] as fp16x2 launches (round 6: EVERY weight gradient of the model can take this form): the
    pipelined 128 x 256 kernel (gi_gemm_b3p.hip) and the 128 x 128-tile kernel (gi_gemm_b3v.hip, GI_GEMM_T128); plain
    split-K slabs and bond-type groups (three groups, the middle one empty, own slab counts); X rows read as stored or
    gathered through b_idx (the first layer of a message stack reads h[u_src]); the ones column exactly 1.0 whatever the
    tensor's scale; dZ of magnitude 1e-3, X of magnitude 30.  Against the fp64 product at 2e-6 of max |ref|; slab tails
    beyond the reduction range exactly zero, nothing written outside [n_out, n_in + 1]."""
    lib = bf3_kernel_switches
    lib.gi_b3p_enable(1); lib.gi_b3v_enable(1)
    g = torch.Generator().manual_seed(rows + n_out + n_in + (7 if gather else 0))
    lddz, ldx, ldc = ops.r4(n_out) + 4, ops.r4(n_in), ops.r4(n_in + 1) + 4
    dZ = torch.randn(rows, lddz, generator=g) * 1e-3; dZ[:, n_out:] = float("nan")
    src_rows = 900 if gather else rows
    Xs = torch.randn(src_rows, ldx, generator=g) * 30
    idx = torch.randint(0, src_rows, (rows,), generator=g, dtype=torch.int32) if gather else None
    X = Xs[idx.long()] if gather else Xs
    cells = torch.zeros(2, L.AMAX_WORDS, device=DEV)
    ops.absmax([dZ[:, :n_out].to(DEV)], cells[0:1])
    ops.absmax([X[:, :n_in].contiguous().to(DEV)], cells[1:2])
    F = L.GEMM_SPLITK | L.GEMM_BF3 | L.GEMM_X2 | (L.GEMM_T128 if kernel == "t128" else 0)
    stride = ops.r4(n_out * ldc)
    common = dict(flags=F, a_major=True, b_major=True, ones_col=n_in, c_split_stride=stride, a_amax=cells[0], b_amax=cells[1],
                  b_idx=idx.to(DEV) if gather else None)
    # plain slabs
    C = torch.full((nsplit, stride), 7.0, device=DEV)
    ops.gemm(dZ.to(DEV), Xs.to(DEV), C, n_out, n_in + 1, rows, lddz, ldx, ldc, nsplit=nsplit, **common)
    S = C[:, :n_out * ldc].view(nsplit, n_out, ldc)
    ref = torch.cat([dZ[:, :n_out].double().t() @ X[:, :n_in].double(), dZ[:, :n_out].double().sum(0)[:, None]], 1)
    assert bool((S[:, :, n_in + 1:] == 7.0).all())
    assert rel(S[:, :, :n_in + 1].double().sum(0).cpu(), ref) < 2e-6
    # grouped: rows [0, a) | empty | [a, rows), slab counts 2 / 1 / nsplit
    a = rows // 3
    off = torch.tensor([0, a, a, rows], dtype=torch.int32)
    gs = (2, 1, nsplit)
    Cg = [torch.full((n, stride), 7.0, device=DEV) for n in gs]
    ops.gemm(dZ.to(DEV), Xs.to(DEV), None, n_out, n_in + 1, rows, lddz, ldx, ldc, grp_off=off.to(DEV), ngroups=3, Cg=Cg,
             gsplit=gs, **common)
    for t, (lo, hi) in enumerate(((0, a), (a, a), (a, rows))):
        St = Cg[t][:, :n_out * ldc].view(gs[t], n_out, ldc)
        rt = torch.cat([dZ[lo:hi, :n_out].double().t() @ X[lo:hi, :n_in].double(), dZ[lo:hi, :n_out].double().sum(0)[:, None]], 1)
        assert bool((St[:, :, n_in + 1:] == 7.0).all()), t
        got = St[:, :, :n_in + 1].double().sum(0).cpu()
        assert float((got - rt).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1e-30), t


def test_absmax_of_pitched_tensors_in_one_launch_ignores_the_padding_columns():
    """gi_absmax on activation-like tensors (row pitch > columns, padding columns never written — NaN here): every
    tensor's cell == max |x| over its real columns; odd column counts, unaligned bases, a single row, many rows."""
    g = torch.Generator().manual_seed(2)
    specs = [(7111, 250, 252), (1000, 501, 504), (3, 37, 40), (1, 128, 136), (513, 129, 132), (2048, 3, 4)]
    ts = []
    for rows, cols, ld in specs:
        t = torch.randn(rows, ld, generator=g) * (1 + len(ts))
        t[:, cols:] = float("nan")
        ts.append(t.to(DEV)[:, :cols])
    base = torch.randn(50, 24, generator=g).to(DEV)
    ts.append(base[:, 1:20])                                       # base not 16-byte aligned
    cells = torch.zeros(len(ts), L.AMAX_WORDS, device=DEV)
    ops.absmax(ts, cells)
    for i, t in enumerate(ts):
        assert float(cells[i].max()) == float(t.abs().max()), i


def test_gemm_fp16x2_needs_both_amax_cells():
    X = torch.randn(256, 64, device=DEV); W = torch.randn(128, 64, device=DEV); Y = torch.zeros(256, 128, device=DEV)
    with pytest.raises(RuntimeError, match="GI_EINVAL"):
        ops.gemm(X, W, Y, 256, 128, 64, 64, 64, 128, flags=L.GEMM_BF3 | L.GEMM_BF3B_F32 | L.GEMM_X2)


@pytest.mark.parametrize("b_major", [False, True])
def test_gemm_split_k_forward_and_dgrad_with_slab_epilogue(b_major, gemm_grid):
    """The skinny, long-reduction layers of the graph-level stacks (B x 500 outputs, K = N*A + G): split-K into
    slabs + gi_slab_epilogue (bias + SELU, or the SELU-backward factor) against the fp64 product."""
    import ctypes as C
    lib = L.load()
    M, N, K, ns = 250, 500, 9252, 16
    g = torch.Generator().manual_seed(3)
    X = torch.randn(M, ops.r4(K), generator=g)
    W = (torch.randn(K, N, generator=g) if b_major else torch.randn(N, K, generator=g)) / K ** 0.5
    b = torch.randn(N, generator=g)
    act = torch.selu(torch.randn(M, ops.r4(N), generator=g))
    ld = ops.r4(N)
    stride = ops.r4(M * ld)
    slabs = torch.full((ns * stride,), float("nan"), device=DEV)
    ops.gemm(X.to(DEV), W.to(DEV), slabs, M, N, K, X.shape[1], W.shape[1], ld, flags=L.GEMM_SPLITK,
             b_major=b_major, nsplit=ns, c_split_stride=stride)
    out = torch.full((M, ld + 4), 7.0, device=DEV)
    flags = L.EPI_DSELU if b_major else (L.EPI_BIAS | L.EPI_SELU)
    bd, ad = b.to(DEV), act.to(DEV)
    L.check(lib.gi_slab_epilogue(slabs.data_ptr(), ns, stride, M, N, ld, flags, bd.data_ptr(),
                                 ad.data_ptr(), ad.shape[1], out.data_ptr(), out.shape[1],
                                 torch.cuda.current_stream().cuda_stream), "gi_slab_epilogue")
    prod = X[:, :K].double() @ (W.double() if b_major else W.double().t())
    ref = prod * D.selu_grad_from_out(act[:, :N].double()) if b_major else D.selu(prod + b.double())
    assert rel(out[:, :N], ref) < 2e-5
    assert bool((out[:, N:] == 7.0).all())


def test_gemm_forward_gather_and_groups(gemm_grid):
    g = torch.Generator().manual_seed(1)
    R, K, N, E = 200, 100, 250, 777
    h = torch.randn(R, 104, generator=g)
    idx = torch.randint(0, R, (E,), generator=g, dtype=torch.int32)
    off = torch.tensor([0, 600, 600, 777], dtype=torch.int32)          # middle group empty
    Ws = [torch.randn(N, K, generator=g) / 10 for _ in range(3)]
    bs = [torch.randn(N, generator=g) for _ in range(3)]
    Y = torch.zeros(E, 252, device=DEV)
    ops.gemm(h.to(DEV), None, Y, E, N, K, 104, K, 252, flags=L.EPI_BIAS | L.EPI_SELU,
             a_idx=idx.to(DEV), tm=1, tn=2, grp_off=off.to(DEV), ngroups=3, max_group_rows=600,
             Bg=[w.to(DEV) for w in Ws], biasg=[b.to(DEV) for b in bs])
    ref = torch.zeros(E, N, dtype=torch.float64)
    for t in range(3):
        lo, hi = int(off[t]), int(off[t + 1])
        ref[lo:hi] = D.selu(h[idx[lo:hi].long(), :K].double() @ Ws[t].double().t() + bs[t].double())
    assert rel(Y[:, :N], ref) < 2e-5


def _chain_case(sizes, off, seed, two=False, x2=False, in_scale=1.0, rows32=False):
    """Resident-activation MLP chain (gi_mlp_chain) against fp64 torch, forward and dZ chain: grouped
    rows (ragged, an empty group, a 1-row group), gathered input, widths that are not multiples of 4
    or 32, every hidden activation / dZ buffer checked, nothing written outside [rows, N]."""
    g = torch.Generator().manual_seed(seed)
    G = len(off) - 1
    E, R = off[-1], 300
    K0 = sizes[0]
    ldx = ops.r4(K0) + 4
    h = torch.randn(R, ldx, generator=g) * in_scale
    idx = torch.randint(0, R, (E,), generator=g, dtype=torch.int32)
    offt = torch.tensor(off, dtype=torch.int32)
    rows_g = [off[t + 1] - off[t] for t in range(G)]
    nchains = 2 if two else 1
    Ws = [[[torch.randn(o, i, generator=g) / i ** 0.5 for _ in range(G)]
           for i, o in zip(sizes, sizes[1:])] for _ in range(nchains)]
    bs = [[[torch.randn(o, generator=g) * 0.3 for _ in range(G)] for o in sizes[1:]]
          for _ in range(nchains)]
    dev = lambda t: t.to(DEV)
    specs, outs = [], []
    for c in range(nchains):
        bufs = [torch.full((E, ops.r4(o) + 4), 7.0, device=DEV) for o in sizes[1:]]
        outs.append(bufs)
        specs.append(dict(X=dev(h), x_idx=dev(idx), grp_off=dev(offt), group_rows=rows_g, rows=E,
                          layers=[dict(W=[dev(w) for w in Ws[c][l]], bias=[dev(b) for b in bs[c][l]],
                                       out=bufs[l], K=sizes[l], N=sizes[l + 1])
                                  for l in range(len(sizes) - 1)]))
    # fp16x2 kernels: amax cells for the input rows and every layer's output (gi_chain_params.x_amax,
    # gi_chain_layer.out_amax — what the stack's fp16x2 weight-gradient launches take their operand scales from)
    fcells = torch.zeros(nchains, len(sizes), L.AMAX_WORDS, device=DEV)
    if x2:
        for c in range(nchains):
            specs[c]["x_amax"] = fcells[c, 0]
            for l in range(len(sizes) - 1):
                specs[c]["layers"][l]["out_amax"] = fcells[c, l + 1]
    ops.mlp_chain(specs, backward=False, x2=x2, rows32=rows32)
    if x2:
        for c in range(nchains):
            assert float(fcells[c, 0].max()) == float(h[idx.long(), :K0].abs().max()), c
            for l in range(len(sizes) - 1):
                assert float(fcells[c, l + 1].max()) == float(outs[c][l][:, :sizes[l + 1]].abs().max()), (c, l)
    refs = []
    for c in range(nchains):
        acts = []
        x = h[idx.long(), :K0].double()
        for l in range(len(sizes) - 1):
            y = torch.zeros(E, sizes[l + 1], dtype=torch.float64)
            for t in range(G):
                lo, hi = off[t], off[t + 1]
                y[lo:hi] = D.selu(x[lo:hi] @ Ws[c][l][t].double().t() + bs[c][l][t].double())
            acts.append(y)
            x = y
            assert rel(outs[c][l][:, :sizes[l + 1]], y) < 3e-5, (c, l)
            assert bool((outs[c][l][:, sizes[l + 1]:] == 7.0).all()), (c, l)
        refs.append(acts)
    # ---- dZ chain: dZ_{l-1} = (dZ_l W_l) * selu'(act_{l-1}); first layer: plain input gradient
    L_ = len(sizes) - 1
    bspecs, bouts = [], []
    for c in range(nchains):
        dZ = torch.randn(E, ops.r4(sizes[-1]), generator=g) * in_scale
        douts = [torch.full((E, ops.r4(sizes[l]) + 4), 7.0, device=DEV) for l in range(L_)]
        layers = []
        for l in range(L_ - 1, -1, -1):
            act = None
            if l > 0:
                act = refs[c][l - 1].float()
                pad = torch.zeros(E, ops.r4(sizes[l]) - sizes[l])
                act = dev(torch.cat([act, pad], 1))
            layers.append(dict(W=[dev(w) for w in Ws[c][l]], out=douts[l], act=act, K=sizes[l + 1],
                               N=sizes[l]))
        bspecs.append(dict(X=dev(dZ), x_idx=None, grp_off=dev(offt), group_rows=rows_g, rows=E,
                           layers=layers))
        bouts.append((dZ, douts))
    bcells = torch.zeros(nchains, len(sizes), L.AMAX_WORDS, device=DEV)
    if x2:
        for c in range(nchains):
            bspecs[c]["x_amax"] = bcells[c, 0]
            for i in range(L_):
                bspecs[c]["layers"][i]["out_amax"] = bcells[c, i + 1]
    ops.mlp_chain(bspecs, backward=True, x2=x2, rows32=rows32)
    if x2:
        for c in range(nchains):
            assert float(bcells[c, 0].max()) == float(bouts[c][0][:, :sizes[-1]].abs().max()), c
            for i in range(L_):                              # chain layer i writes douts[L_ - 1 - i]
                l = L_ - 1 - i
                assert float(bcells[c, i + 1].max()) == float(bouts[c][1][l][:, :sizes[l]].abs().max()), (c, l)
    for c in range(nchains):
        dZ, douts = bouts[c]
        z = dZ[:, :sizes[-1]].double()
        for l in range(L_ - 1, -1, -1):
            nxt = torch.zeros(E, sizes[l], dtype=torch.float64)
            for t in range(G):
                lo, hi = off[t], off[t + 1]
                nxt[lo:hi] = z[lo:hi] @ Ws[c][l][t].double()
            if l > 0:
                nxt = nxt * D.selu_grad_from_out(refs[c][l - 1].float().double())
            assert rel(douts[l][:, :sizes[l]], nxt) < 3e-5, (c, l)
            assert bool((douts[l][:, sizes[l]:] == 7.0).all()), (c, l)
            z = nxt


@pytest.mark.parametrize("sizes,off", [
    ((128, 250, 250, 250, 250, 128), [0, 600, 600, 777]),       # bench config, middle group empty
    ((100, 250, 250, 250, 250, 100), [0, 33, 34, 131]),         # reference default dims
    ((16, 24, 24, 12), [0, 5, 7, 8]),                           # tiny config
    ((37, 256, 7, 130), [0, 64]),                               # one group, awkward widths, depth 2
    ((128, 100), [0, 31, 95]),                                  # single layer
])
def test_mlp_chain_forward_and_dz_chain(sizes, off):
    _chain_case(sizes, off, seed=sum(sizes))


def test_mlp_chain_two_chains_one_launch():
    _chain_case((100, 250, 250, 100), [0, 90, 130, 131], seed=5, two=True)


@pytest.mark.parametrize("in_scale", [1.0, 1e-4, 3e3])
@pytest.mark.parametrize("sizes,off", [
    ((128, 250, 250, 250, 250, 128), [0, 600, 600, 777]),       # bench config, middle group empty
    ((100, 250, 250, 250, 250, 100), [0, 33, 34, 131]),         # reference default dims
    ((100, 250, 250, 250, 250, 100), [0, 1000, 1001, 2500]),    # several 64-row blocks per group, ragged tails
    ((16, 24, 24, 12), [0, 5, 7, 8]),                           # tiny config
    ((37, 256, 7, 130), [0, 64]),                               # one group, awkward widths, exactly one block
    ((128, 100), [0, 31, 95]),                                  # single layer
])
def test_mlp_chain_fp16x2_forward_and_dz_chain(sizes, off, in_scale):
    """The fp16x2 chain (gi_chain_params.x2_wamax; csrc/gi_x2.h): 64-row blocks, weights as two scaled fp16 planes
    (scale per layer and group from gi_mlp_chain_pack's own amax pass), activations split per row block and layer in
    the epilogue — against the same fp64 reference and at the same 3e-5 as the fp32-MFMA chain, for inputs (and
    upstream gradients) of magnitude 1, 1e-4 and 3e3, single and two chains per launch."""
    _chain_case(sizes, off, seed=sum(sizes), x2=True, in_scale=in_scale)
    if in_scale == 1.0:
        _chain_case(sizes, off, seed=sum(sizes) + 1, two=True, x2=True)


@pytest.mark.parametrize("in_scale", [1.0, 1e-4, 3e3])
@pytest.mark.parametrize("sizes,off", [
    ((128, 250, 250, 250, 250, 128), [0, 600, 600, 777]),       # bench config, middle group empty
    ((100, 250, 250, 250, 250, 100), [0, 33, 34, 131]),         # reference default dims
    ((100, 250, 250, 250, 250, 100), [0, 1000, 1001, 2500]),    # many 32-row blocks per group, ragged tails
    ((16, 24, 24, 12), [0, 5, 7, 8]),                           # tiny config
    ((37, 256, 7, 130), [0, 64]),                               # one group, awkward widths
    ((128, 100), [0, 31, 95]),                                  # single layer
])
def test_mlp_chain_fp16x2_row_independent_variant(sizes, off, in_scale):
    """gi_chain_params.x2_rows32 (round 5): 32-row blocks, every ROW scaled by its own power of two, the product formed
    transposed so that a lane holds one row — the same fp64 reference and the same 3e-5 as the other two chain
    kernels, forward and dZ chain, magnitudes 1 / 1e-4 / 3e3, one and two chains per launch."""
    _chain_case(sizes, off, seed=sum(sizes), x2=True, in_scale=in_scale, rows32=True)
    if in_scale == 1.0:
        _chain_case(sizes, off, seed=sum(sizes) + 1, two=True, x2=True, rows32=True)


@pytest.mark.parametrize("backward", [False, True])
def test_mlp_chain_fp16x2_pack_in_one_launch_equals_the_three_launches(backward, monkeypatch):
    """gi_mlp_chain_pack of an fp16x2 chain (round 6): ONE launch that takes max |W| of every (layer, group) matrix,
    writes its amax cell and packs the two-plane image — against memset + gi_absmax + the pack kernel
    (GI_CHAIN_PACK_FUSED=0): the cells' maxima and the images bit for bit, forward (W as stored) and dZ chain (W^T),
    ragged widths, an empty group."""
    sizes, off = (100, 250, 136, 250, 36), [0, 300, 300, 1000]
    g = torch.Generator().manual_seed(77)
    G, E = len(off) - 1, off[-1]
    order = sizes[::-1] if backward else sizes
    Ws = [[(torch.randn(o, i, generator=g) * 10.0 ** (l - 2)).to(DEV) for _ in range(G)]
          for l, (i, o) in enumerate(zip(sizes, sizes[1:]))]
    if backward:
        Ws = Ws[::-1]
    X = torch.randn(E, ops.r4(order[0]), generator=g).to(DEV)
    got = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("GI_CHAIN_PACK_FUSED", fused)
        outs = [torch.zeros(E, ops.r4(n), device=DEV) for n in order[1:]]
        layers = []
        for l in range(len(order) - 1):
            ly = dict(W=Ws[l], out=outs[l], K=order[l], N=order[l + 1])
            if not backward:
                ly["bias"] = [torch.zeros(order[l + 1], device=DEV) for _ in range(G)]
            layers.append(ly)
        spec = dict(X=X, grp_off=torch.tensor(off, dtype=torch.int32, device=DEV),
                    group_rows=[off[t + 1] - off[t] for t in range(G)], rows=E, layers=layers)
        cells, image = ops.mlp_chain([spec], backward=backward, x2=True)
        torch.cuda.synchronize()
        slots = cells.view(-1, L.AMAX_WORDS)[:, ::L.AMAX_WORDS // 64]          # (64 slots a 128-byte line apart; the rest is never written)
        # the image buffer has the fp32 layout's size; the fp16x2 tiles (16-deep, 16 KB each) fill the front of each group's share
        per_group = image.numel() // G
        written = sum((k + 15) // 16 for k in order[:-1]) * 4096
        img = image.view(torch.int32).view(G, per_group)[:, :written]
        got[fused] = (slots.max(1).values.cpu(), img.cpu(), [o.cpu() for o in outs])
    want = torch.stack([w.abs().max().cpu() for lw in Ws for w in lw])
    assert torch.equal(got["1"][0], want) and torch.equal(got["0"][0], want)
    assert torch.equal(got["0"][1], got["1"][1])
    for a, b in zip(got["0"][2], got["1"][2]):
        assert torch.equal(a, b)


def test_mlp_chain_fp16x2_rows32_two_workgroups_per_cu():
    """More row blocks than CUs -> the launcher takes the two-workgroups-per-CU build of the kernel (two-slot weight ring,
    outputs stored from registers): 264 and 700 blocks in three ragged groups, forward and dZ chain, one and two chains."""
    _chain_case((128, 250, 250, 128), [0, 2816, 5632, 8448], seed=8448, x2=True, rows32=True)
    _chain_case((100, 250, 250, 250, 250, 100), [0, 15000, 15001, 22400], seed=22400, x2=True, rows32=True)
    _chain_case((128, 250, 128), [0, 7000, 7000, 8600], seed=8600, two=True, x2=True, rows32=True)


def test_mlp_chain_fp16x2_rows32_is_bitwise_row_independent():
    """What the FORWARD needs of its chain (pass-0 row cache, blocking == host-sync-free, tape == no tape): a row's
    outputs depend on the row and the weights only, bit for bit — whichever rows share its block, whatever their
    magnitudes.  The same rows run (a) in their original order, (b) permuted, with every other row multiplied by 1e4
    so that each block's companions change completely; the 64-row per-block variant is NOT invariant (asserted, so that
    the test would notice if it were handed the wrong kernel)."""
    g = torch.Generator().manual_seed(3)
    sizes, E = (128, 250, 250, 128), 500
    X = torch.randn(E, 128, generator=g)
    Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).to(DEV) for i, o in zip(sizes, sizes[1:])]
    bs = [(torch.randn(o, generator=g) * 0.3).to(DEV) for o in sizes[1:]]

    def run(Xin, rows32):
        n = Xin.shape[0]
        outs = [torch.zeros(n, ops.r4(o), device=DEV) for o in sizes[1:]]
        spec = dict(X=Xin.to(DEV), x_idx=None, grp_off=None, group_rows=[n], rows=n,
                    layers=[dict(W=[Ws[l]], bias=[bs[l]], out=outs[l], K=sizes[l], N=sizes[l + 1]) for l in range(3)])
        ops.mlp_chain([spec], backward=False, x2=True, rows32=rows32)
        torch.cuda.synchronize()
        return [o.cpu() for o in outs]

    perm = torch.randperm(E, generator=g)
    big = torch.zeros(2 * E, 128)
    big[0::2] = X[perm]
    big[1::2] = X * 1e4                                         # loud neighbours in every block
    for rows32, expect_equal in ((True, True), (False, False)):
        a, b = run(X, rows32), run(big, rows32)
        same = all(torch.equal(b[l][0::2], a[l][perm]) for l in range(3))
        assert same == expect_equal, (rows32, same)
    # ... and of the build of the kernel: the same rows inside a 9 000-row input (282 blocks: two workgroups per CU)
    a = run(X, True)
    huge = torch.randn(9000, 128, generator=g) * 30
    at = torch.randperm(9000, generator=g)[:E]
    huge[at] = X
    b = run(huge, True)
    assert all(torch.equal(b[l][at], a[l]) for l in range(3))


@pytest.mark.parametrize("tile_rows", [33, 34, 36])
@pytest.mark.parametrize("sizes,off", [
    ((128, 250, 250, 250, 250, 128), [0, 600, 600, 777]),
    ((100, 250, 250, 100), [0, 33, 34, 131]),
    ((37, 256, 7, 130), [0, 70]),
])
def test_mlp_chain_taller_row_blocks(chain_cfg, tile_rows, sizes, off):
    """Row blocks of 32 + x rows (x <= 4 extra rows computed on the VALU; what the launcher picks when
    it saves a round of workgroups): forced here so that every height is exercised on ragged groups —
    blocks with 0, some and all of their extra rows valid."""
    chain_cfg(tile_rows=tile_rows)
    _chain_case(sizes, off, seed=tile_rows + sum(sizes))
    _chain_case(sizes, off, seed=tile_rows, two=True)


@pytest.mark.parametrize("sizes,off", [
    ((128, 250, 250, 250, 250, 128), [0, 600, 600, 777]),
    ((100, 250, 250, 100), [0, 33, 34, 131]),
    ((100, 250, 250, 250, 250, 100), [0, 1000, 1001, 2500]),
    ((37, 256, 7, 130), [0, 70]),
])
def test_mlp_chain_64_row_blocks(chain_cfg, sizes, off):
    """The <2, 2> variant (64 rows per workgroup, two weight tiles in LDS; what the launcher picks for
    batches of several rounds of row blocks), forced here on ragged groups: full blocks, a 1-row group, an
    empty group, blocks whose second half is partly / completely out of range — forward and dZ chain."""
    chain_cfg(rows64=1)
    _chain_case(sizes, off, seed=64 + sum(sizes))
    _chain_case(sizes, off, seed=64, two=True)


@pytest.mark.parametrize("tile_rows", [None, 34])
@pytest.mark.parametrize("sizes,off", [
    ((128, 250, 250, 250, 250, 128), [0, 600, 600, 777]),
    ((100, 250, 250, 100), [0, 33, 34, 131]),
    ((37, 256, 7, 130), [0, 70]),
])
def test_mlp_chain_three_slot_ring(chain_cfg, tile_rows, sizes, off):
    """gi_chain_kernel<BWD, 1, 3>: 32-row blocks with the three-slot weight ring (146 KB of LDS) — the default
    until round 3, now gi_mlp_chain_config(ring = 3) for measurements (the two-slot ring, 114 KB, lets a GEMM
    workgroup share the CU and is what every other chain test runs)."""
    chain_cfg(ring=3, tile_rows=tile_rows or 0)
    _chain_case(sizes, off, seed=2 + sum(sizes))
    _chain_case(sizes, off, seed=2, two=True)


def test_class_sum_dselu_long_segments():
    """gi_class_sum_dselu: per output row the sum of hundreds of indexed rows, times selu'(y), two
    problems in one launch, run-to-run bit identical (fixed tree)."""
    g = torch.Generator().manual_seed(3)
    E, D0, M = 5000, 37, 100
    ld = ops.r4(M)
    cls = torch.randint(0, D0, (E,), generator=g)
    cls[cls == 5] = 6                                        # an empty segment
    idx = torch.argsort(cls, stable=True).int()
    off = torch.cat([torch.zeros(1, dtype=torch.long), torch.bincount(cls, minlength=D0).cumsum(0)]).int()
    v0, v1 = torch.randn(E, ld, generator=g), torch.randn(E, ld, generator=g)
    y0, y1 = D.selu(torch.randn(D0, ld, generator=g)), D.selu(torch.randn(D0, ld, generator=g))
    outs = []
    v0d, v1d, idxd, offd = v0.to(DEV), v1.to(DEV), idx.to(DEV), off.to(DEV)
    for _ in range(2):
        a, b = y0.clone().to(DEV), y1.clone().to(DEV)
        L.check(L.load().gi_class_sum_dselu(v0d.data_ptr(), v1d.data_ptr(), ld, idxd.data_ptr(),
                                            offd.data_ptr(), D0, M, a.data_ptr(), b.data_ptr(), ld,
                                            torch.cuda.current_stream().cuda_stream), "class_sum")
        torch.cuda.synchronize()
        outs.append((a.cpu(), b.cpu()))
    for got, v, y in ((outs[0][0], v0, y0), (outs[0][1], v1, y1)):
        ref = D.seg_sum(v[:, :M].double(), idx, off, D0) * D.selu_grad_from_out(y[:, :M].double())
        assert rel(got[:, :M], ref) < 1e-5
        assert torch.equal(got[:, M:], y[:, M:])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_mlp_chain_limits_are_reported():
    x = torch.zeros(8, 260, device=DEV)
    w = torch.zeros(300, 256, device=DEV)
    spec = dict(X=x, rows=8, layers=[dict(W=[w], bias=[torch.zeros(300, device=DEV)],
                                         out=torch.zeros(8, 300, device=DEV), K=256, N=300)])
    with pytest.raises(RuntimeError, match="GI_ELIMIT"):
        ops.mlp_chain([spec])


@pytest.mark.parametrize("tm,tn", TILES)
def test_gemm_dgrad_dselu_inplace_and_accumulate(tm, tn, gemm_grid):
    g = torch.Generator().manual_seed(2)
    R, n_out, n_in = 517, 250, 500
    dZ = torch.randn(R, 252, generator=g)
    W = torch.randn(n_out, n_in, generator=g) / 16
    act = D.selu(torch.randn(R, n_in, generator=g))
    buf = act.clone().to(DEV)                                           # in place: C == act
    ops.gemm(dZ.to(DEV), W.to(DEV), buf, R, n_in, n_out, 252, n_in, n_in, flags=L.EPI_DSELU,
             act=buf, ldact=n_in, b_major=True, tm=tm, tn=tn)
    ref = (dZ[:, :n_out].double() @ W.double()) * D.selu_grad_from_out(act.double())
    assert rel(buf, ref) < 2e-5
    acc = torch.ones(R, 104, device=DEV)                                 # first 100 columns only, +=
    ops.gemm(dZ.to(DEV), W.to(DEV), acc, R, 100, n_out, 252, n_in, 104, flags=L.EPI_ACCUM,
             b_major=True, tm=tm, tn=tn)
    ref2 = 1.0 + dZ[:, :n_out].double() @ W.double()[:, :100]
    assert rel(acc[:, :100], ref2) < 2e-5
    assert bool((acc[:, 100:] == 1.0).all())


@pytest.mark.parametrize("tn", [1, 2])
@pytest.mark.parametrize("R,n_out,n_in,nsplit", [(1000, 250, 100, 4), (999, 45, 500, 7),
                                                 (70, 500, 128, 1), (33, 1, 500, 3)])
def test_gemm_wgrad_slabs_and_reduce(R, n_out, n_in, nsplit, tn, gemm_grid):
    g = torch.Generator().manual_seed(R)
    dZ = torch.randn(R, ops.r4(n_out), generator=g)
    X = torch.randn(R, ops.r4(n_in) + 4, generator=g)
    ld = ops.r4(n_in + 1)
    stride = ops.r4(n_out * ld)
    slabs = torch.full((nsplit * stride,), float("nan"), device=DEV)
    ops.gemm(dZ.to(DEV), X.to(DEV), slabs, n_out, n_in + 1, R, dZ.shape[1], X.shape[1], ld,
             flags=L.GEMM_SPLITK, a_major=True, b_major=True, tm=1, tn=tn, nsplit=nsplit,
             c_split_stride=stride, ones_col=n_in)
    dW = torch.empty(n_out, n_in, device=DEV)
    db = torch.empty(n_out, device=DEV)
    ops.reduce_slabs([(slabs, dW, db, stride, nsplit, n_out, n_in, ld)])
    assert rel(dW, dZ[:, :n_out].double().t() @ X[:, :n_in].double()) < 2e-5
    assert rel(db, dZ[:, :n_out].double().sum(0)) < 2e-5


def test_gemm_wgrad_grouped_gather(gemm_grid):
    g = torch.Generator().manual_seed(3)
    Rn, E, n_out, n_in, nsplit = 150, 901, 250, 100, 3
    h = torch.randn(Rn, 104, generator=g)
    idx = torch.randint(0, Rn, (E,), generator=g, dtype=torch.int32)
    dZ = torch.randn(E, 252, generator=g)
    off = torch.tensor([0, 700, 890, 890, 901], dtype=torch.int32)       # 4 groups, one empty
    ld = ops.r4(n_in + 1)
    stride = ops.r4(n_out * ld)
    slabs = [torch.full((nsplit * stride,), float("nan"), device=DEV) for _ in range(4)]
    ops.gemm(dZ.to(DEV), h.to(DEV), None, n_out, n_in + 1, E, 252, 104, ld, flags=L.GEMM_SPLITK,
             a_major=True, b_major=True, b_idx=idx.to(DEV), tm=1, tn=2, nsplit=nsplit,
             c_split_stride=stride, ones_col=n_in, grp_off=off.to(DEV), ngroups=4, Cg=slabs)
    for t in range(4):
        lo, hi = int(off[t]), int(off[t + 1])
        dW = torch.empty(n_out, n_in, device=DEV)
        db = torch.empty(n_out, device=DEV)
        ops.reduce_slabs([(slabs[t], dW, db, stride, nsplit, n_out, n_in, ld)])
        refW = dZ[lo:hi, :n_out].double().t() @ h[idx[lo:hi].long(), :n_in].double()
        if hi > lo:
            assert rel(dW, refW) < 2e-5 and rel(db, dZ[lo:hi, :n_out].double().sum(0)) < 2e-5
        else:
            assert float(dW.abs().max()) == 0.0 and float(db.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------
def test_seg_sum_and_accumulate():
    n8, e8, _ = synthetic.make_batch(200, **synthetic.SHAPES["gdb13"], seed=9)
    ref = D.compact(n8, e8)
    S, E, U = ref["S"], ref["E"], ref["U"]
    g = torch.Generator().manual_seed(4)
    for cols in (128, 100, 12):
        ld = ops.r4(cols)
        vals = torch.randn(U, ld, generator=g)                 # message rows
        for perm_k, off_k in (("in_perm", "seg_off"), ("out_perm", "src_off")):
            perm = torch.from_numpy(ref[perm_k])
            off = torch.from_numpy(ref[off_k])
            out = torch.full((S + 1, ld), 3.0, device=DEV)
            ops.seg_sum(vals.to(DEV), perm.to(DEV), off.to(DEV), S + 1, cols, out)
            want = D.seg_sum(vals.double(), perm, off, S + 1)
            assert rel(out[:, :cols], want[:, :cols]) < 1e-6
            assert float(out[S].abs().max()) == 0.0
            ops.seg_sum(vals.to(DEV), perm.to(DEV), off.to(DEV), S + 1, cols, out, accumulate=True)
            assert rel(out[:, :cols], 2 * want[:, :cols]) < 1e-6
        # backward of the aggregation over the message CSR, fused with the SELU backward (in place)
        dagg = torch.randn(S + 1, ld, generator=g)
        y = D.selu(torch.randn(U, ld, generator=g))
        buf = y.clone().to(DEV)
        lib = L.load()
        dagg_d = dagg.to(DEV)                                  # keep the device copies alive
        mu_dst_d, mu_off_d = (torch.from_numpy(ref[k]).to(DEV) for k in ("mu_dst", "mu_off"))
        L.check(lib.gi_seg_sum_dselu(dagg_d.data_ptr(), ld, mu_dst_d.data_ptr(), mu_off_d.data_ptr(),
                                     U, cols, buf.data_ptr(), ld,
                                     torch.cuda.current_stream().cuda_stream), "dselu")
        want = D.seg_sum(dagg.double(), torch.from_numpy(ref["mu_dst"]), torch.from_numpy(ref["mu_off"]), U) \
            * D.selu_grad_from_out(y.double())
        assert rel(buf[:, :cols], want[:, :cols]) < 1e-6


# ---- GI_FUSE variants (launch-count reductions): each against the launches it replaces -----------
def _stream():
    return torch.cuda.current_stream().cuda_stream


def _gates_case(H, Fn, seed):
    """A real graph batch's CSRs plus random GRU tensors; rows without incoming edges included."""
    n8, e8, _ = synthetic.make_batch(37, **synthetic.SHAPES["gdb13"], seed=seed)
    g = D.compact(n8, e8)
    S, U = g["S"], g["U"]
    R = S + 1
    gen = torch.Generator().manual_seed(seed)
    ldg, ldhx, ldH = ops.r4(3 * H), ops.r4(H + Fn), ops.r4(H)
    gi = torch.randn(R, ldg, generator=gen)
    gh = torch.randn(R, ldg, generator=gen)
    hx = torch.randn(R, ldhx, generator=gen)
    seg = torch.from_numpy(g["seg_off"])
    has_edge = (seg[1:R + 1] - seg[:R]) > 0
    assert bool(has_edge.any()) and not bool(has_edge.all())
    return g, R, U, gi, gh, hx, seg, has_edge, ldg, ldhx, ldH, gen


@pytest.mark.parametrize("H,Fn", [(128, 8), (100, 8), (16, 5), (18, 5)])
def test_gru_gates_forward_and_backward_kernels(H, Fn):
    """gi_gru_gates_fwd / gi_gru_gates_bwd / gi_gru_gates_bwd_ex (16-byte vector kernels when H % 4 == 0,
    scalar otherwise — H = 18) against the fp64 dataflow model; the vector and the scalar backward agree."""
    lib = L.load()
    g, R, U, gi, gh, hx, seg, has_edge, ldg, ldhx, ldH, gen = _gates_case(H, Fn, 3)
    h_new, saved = D.gru_gates(gi[:, :3 * H].double(), gh[:, :3 * H].double(), hx[:, :H].double(), has_edge)
    gi_d, gh_d, hx_d, seg_d = gi.to(DEV), gh.to(DEV), hx.to(DEV), seg.to(DEV)
    hx_new = torch.full((R, ldhx), 7.0, device=DEV)
    L.check(lib.gi_gru_gates_fwd(gi_d.data_ptr(), gh_d.data_ptr(), ldg, hx_d.data_ptr(), hx_new.data_ptr(),
                                 ldhx, seg_d.data_ptr(), R, H, Fn, _stream()), "gates_fwd")
    assert rel(hx_new[:, :H], h_new) < 2e-6
    assert torch.equal(hx_new[:, H:].cpu(), hx[:, H:])                      # feature tail + padding copied
    e = has_edge.numpy()
    for got, want in ((gi_d[:, :H], saved[0]), (gi_d[:, H:2 * H], saved[1]), (gi_d[:, 2 * H:3 * H], saved[2])):
        assert rel(got[e], want[e]) < 2e-6
    assert torch.equal(gi_d[~has_edge.to(DEV)].cpu(), gi[~has_edge])       # untouched without edges
    assert torch.equal(gh_d.cpu(), gh)
    # backward: d h = dh + dhb + dhc + dhd
    dhs = [torch.randn(R, ldH, generator=gen) for _ in range(4)]
    want_gi, want_gh, want_dh = D.gru_gates_bwd(sum(t[:, :H].double() for t in dhs),
                                                tuple(t.double() for t in saved), hx[:, :H].double(), has_edge)
    outs = []
    for entry in ("gi_gru_gates_bwd", "gi_gru_gates_bwd_ex"):
        a, b = gi_d.clone(), gh_d.clone()
        dprev = torch.full((R, ldH), 7.0, device=DEV)
        dd = [t.to(DEV) for t in dhs]
        args = [a.data_ptr(), b.data_ptr(), ldg, hx_d.data_ptr(), ldhx] + [t.data_ptr() for t in dd] + \
               [dprev.data_ptr(), ldH, seg_d.data_ptr(), R, H]
        if entry.endswith("_ex"):
            args += [None, None, 0, None, None]
        L.check(getattr(lib, entry)(*args, _stream()), entry)
        assert rel(a[:, :3 * H], want_gi) < 2e-6 and rel(b[:, :3 * H], want_gh) < 2e-6
        assert rel(dprev[:, :H], want_dh) < 2e-6
        outs.append((a, b, dprev))
    for x, y in zip(*outs):
        assert rel(x[:, :3 * H] if x.shape[1] >= 3 * H else x[:, :H],
                   y[:, :3 * H] if y.shape[1] >= 3 * H else y[:, :H]) < 1e-6


@pytest.mark.parametrize("R,H,M,Fn", [(7258, 128, 128, 8), (1000, 100, 100, 8), (333, 256, 64, 19), (65, 24, 36, 4), (1, 128, 128, 8)])
def test_fused_gru_update_vs_fp64_and_vs_the_two_launch_path(R, H, M, Fn):
    """gi_gru_forward (csrc/gi_gru.hip, round 6): projections + gates of torch.nn.GRUCell (gnn/mpnn.py:296-297) with the
    node mask (summation_mpnn.py:146) in one launch, against (a) the fp64 arithmetic — h' and the stored gates r, z, n,
    gh_n at 2e-6 — and (b) the two-launch path it replaces (gi_gemm x 2 + gi_gru_gates_fwd) at 2e-6; the feature tail /
    padding of the state rows is copied, rows of nodes without incoming edges keep their state bit for bit, nothing is
    written outside [R, 3H] of gi / gh."""
    lib = L.load()
    g = torch.Generator().manual_seed(R + H + M)
    lda, ldh, ldg = ops.r4(M) + 4, ops.r4(H + Fn), ops.r4(3 * H) + 4
    agg = torch.randn(R, lda, generator=g); agg[:, M:] = float("nan")
    hx = torch.randn(R, ldh, generator=g)
    Wih = torch.randn(3 * H, M, generator=g) / M ** 0.5; Whh = torch.randn(3 * H, H, generator=g) / H ** 0.5
    bih = torch.randn(3 * H, generator=g) * 0.3; bhh = torch.randn(3 * H, generator=g) * 0.3
    deg = torch.randint(0, 3, (R,), generator=g)
    seg = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).int()
    live = deg > 0
    # fp64 reference
    gi64 = agg[:, :M].double() @ Wih.double().t() + bih.double()
    gh64 = hx[:, :H].double() @ Whh.double().t() + bhh.double()
    r = torch.sigmoid(gi64[:, :H] + gh64[:, :H]); z = torch.sigmoid(gi64[:, H:2 * H] + gh64[:, H:2 * H])
    n = torch.tanh(gi64[:, 2 * H:] + r * gh64[:, 2 * H:])
    h_new = torch.where(live[:, None], (1 - z) * n + z * hx[:, :H].double(), hx[:, :H].double())
    dev = lambda t: t.to(DEV)
    gi_d = torch.full((R, ldg), 7.0, device=DEV); gh_d = torch.full((R, ldg), 7.0, device=DEV)
    hx_new = torch.full((R, ldh), 7.0, device=DEV)
    hx_d, seg_d = dev(hx), dev(seg)
    agg_d, Wih_d, Whh_d, bih_d, bhh_d = dev(agg), dev(Wih), dev(Whh), dev(bih), dev(bhh)     # (kept alive: raw pointers below)
    L.check(lib.gi_gru_forward(agg_d.data_ptr(), lda, hx_d.data_ptr(), ldh, Wih_d.data_ptr(), Whh_d.data_ptr(),
                               bih_d.data_ptr(), bhh_d.data_ptr(), gi_d.data_ptr(), gh_d.data_ptr(), ldg,
                               hx_new.data_ptr(), seg_d.data_ptr(), R, H, M, _stream()), "gi_gru_forward")
    assert rel(hx_new[:, :H], h_new) < 2e-6
    assert torch.equal(hx_new[:, H:].cpu(), hx[:, H:])
    assert torch.equal(hx_new[~live.to(DEV), :H].cpu(), hx[~live, :H])
    e = live
    if bool(e.any()):
        for got, want in ((gi_d[:, :H], r), (gi_d[:, H:2 * H], z), (gi_d[:, 2 * H:3 * H], n), (gh_d[:, 2 * H:3 * H], gh64[:, 2 * H:])):
            assert rel(got.cpu()[e], want[e]) < 2e-6
    assert bool((gi_d[:, 3 * H:] == 7.0).all()) and bool((gh_d[:, 3 * H:] == 7.0).all()) and bool((gh_d[:, :2 * H] == 7.0).all())
    assert bool((gi_d[~live.to(DEV)] == 7.0).all())
    # the two-launch path on the same operands
    gi2 = torch.zeros(R, ldg, device=DEV); gh2 = torch.zeros(R, ldg, device=DEV); hx2 = torch.full((R, ldh), 7.0, device=DEV)
    ops.gemm(agg_d, Wih_d, gi2, R, 3 * H, M, lda, M, ldg, flags=L.EPI_BIAS, bias=bih_d)
    ops.gemm(hx_d, Whh_d, gh2, R, 3 * H, H, ldh, H, ldg, flags=L.EPI_BIAS, bias=bhh_d)
    L.check(lib.gi_gru_gates_fwd(gi2.data_ptr(), gh2.data_ptr(), ldg, hx_d.data_ptr(), hx2.data_ptr(), ldh,
                                 seg_d.data_ptr(), R, H, Fn, _stream()), "gates_fwd")
    assert rel(hx_new[:, :H], hx2[:, :H]) < 2e-6
    # limits are reported
    assert lib.gi_gru_forward(agg_d.data_ptr(), lda, hx_d.data_ptr(), ldh, Wih_d.data_ptr(), Whh_d.data_ptr(),
                              bih_d.data_ptr(), bhh_d.data_ptr(), gi_d.data_ptr(), gh_d.data_ptr(), ldg,
                              hx_new.data_ptr(), seg_d.data_ptr(), R, H - 2, M, _stream()) == -1


@pytest.mark.parametrize("two", [False, True])
def test_gru_gates_backward_with_fused_scatter_is_bit_exact(two):
    """gi_gru_gates_bwd_ex(sc0[, sc1]) == gi_seg_sum(sc, out_perm, src_off, dh, accumulate) launches in front
    of the plain gate backward, bit for bit (same summation order)."""
    lib = L.load()
    H, Fn = 100, 8
    g, R, U, gi, gh, hx, seg, has_edge, ldg, ldhx, ldH, gen = _gates_case(H, Fn, 5)
    perm, off = torch.from_numpy(g["out_perm"]).to(DEV), torch.from_numpy(g["src_off"]).to(DEV)
    sc = [torch.randn(U, ldH, generator=gen).to(DEV) for _ in range(2 if two else 1)]
    dh = torch.randn(R, ldH, generator=gen)
    hx_d, seg_d = hx.to(DEV), seg.to(DEV)
    res = []
    for fused in (False, True):
        a, b, d = gi.to(DEV), gh.to(DEV), dh.to(DEV)
        dprev = torch.full((R, ldH), 7.0, device=DEV)
        if not fused:
            for t in sc:
                ops.seg_sum(t, perm, off, R, H, d, accumulate=True)
        tail = [None, None, 0, None, None] if not fused else \
            [sc[0].data_ptr(), sc[1].data_ptr() if two else None, ldH, perm.data_ptr(), off.data_ptr()]
        L.check(lib.gi_gru_gates_bwd_ex(a.data_ptr(), b.data_ptr(), ldg, hx_d.data_ptr(), ldhx, d.data_ptr(),
                                        None, None, None, dprev.data_ptr(), ldH, seg_d.data_ptr(), R, H,
                                        *tail, _stream()), "gates_bwd_ex")
        res.append((a[:, :3 * H], b[:, :3 * H], dprev[:, :H]))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    # the scalar-kernel shapes cannot take the fused scatter: reported, not silently ignored
    assert lib.gi_gru_gates_bwd_ex(gi.to(DEV).data_ptr(), gh.to(DEV).data_ptr(), ldg, hx_d.data_ptr(), ldhx,
                                   dh.to(DEV).data_ptr(), None, None, None, dprev.data_ptr(), ldH,
                                   seg_d.data_ptr(), R, 98, sc[0].data_ptr(), None, ldH, perm.data_ptr(),
                                   off.data_ptr(), _stream()) == -1


def test_selu_bwd_cols3_matches_three_launches():
    lib = L.load()
    gen = torch.Generator().manual_seed(8)
    B, n0, n1, n2 = 57, 585, 39, 1
    W = n0 + n1 + n2
    Y = D.selu(torch.randn(B, W, generator=gen)).to(DEV)
    dY = torch.randn(B, W + 3, generator=gen).to(DEV)
    lds = (ops.r4(n0), ops.r4(n1), 4)
    a = [torch.full((B, ld), 7.0, device=DEV) for ld in lds]
    b = [torch.full((B, ld), 7.0, device=DEV) for ld in lds]
    start = 0
    for t, n in zip(a, (n0, n1, n2)):
        L.check(lib.gi_selu_bwd_rows(dY.data_ptr() + 4 * start, dY.stride(0), None, Y.data_ptr() + 4 * start,
                                     Y.stride(0), t.data_ptr(), t.stride(0), B, n, _stream()), "selu_bwd_rows")
        start += n
    L.check(lib.gi_selu_bwd_cols3_f(dY.data_ptr(), dY.stride(0), Y.data_ptr(), Y.stride(0), 0, B, n0,
                                    b[0].data_ptr(), lds[0], n1, b[1].data_ptr(), lds[1], n2, b[2].data_ptr(),
                                    lds[2], _stream()), "selu_bwd_cols3")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert rel(b[1][:, :n1], dY[:, n0:n0 + n1].double() * D.selu_grad_from_out(Y[:, n0:n0 + n1].double().cpu()).to(DEV)) < 1e-6


@pytest.mark.parametrize("shape,B", [("gdb13", 61), ("chembl", 9)])
def test_slot_glue_pairs_match_single_launches(shape, B):
    """gi_expand_slots2 / gi_compress_slots2_f == two gi_expand_slots / gi_compress_slots launches."""
    lib = L.load()
    sh = synthetic.SHAPES[shape]
    n8, e8, _ = synthetic.make_batch(B, **sh, seed=12)
    ref = D.compact(n8, e8)
    S, N = ref["S"], sh["max_n_nodes"]
    cidx = torch.from_numpy(ref["cidx"]).to(DEV)
    gen = torch.Generator().manual_seed(13)
    Wa, Wb = sh["n_atom_types"] * sh["n_formal_charge"] * sh["n_edge_features"], sh["n_edge_features"]
    t1 = [D.selu(torch.randn(S + 1, ops.r4(W), generator=gen)).to(DEV) for W in (Wa, Wb)]
    ldc = [ops.r4(N * W + 100) for W in (Wa, Wb)]
    cat_a = [torch.full((B, ld), 7.0, device=DEV) for ld in ldc]
    cat_b = [torch.full((B, ld), 7.0, device=DEV) for ld in ldc]
    for t, c, W in zip(t1, cat_a, (Wa, Wb)):
        L.check(lib.gi_expand_slots(t.data_ptr(), t.stride(0), cidx.data_ptr(), B, N, W, c.data_ptr(),
                                    c.stride(0), _stream()), "expand")
    L.check(lib.gi_expand_slots2(t1[0].data_ptr(), t1[0].stride(0), Wa, cat_b[0].data_ptr(), ldc[0],
                                 t1[1].data_ptr(), t1[1].stride(0), Wb, cat_b[1].data_ptr(), ldc[1],
                                 cidx.data_ptr(), B, N, _stream()), "expand2")
    for x, y in zip(cat_a, cat_b):
        assert torch.equal(x, y)
    assert torch.equal(cat_b[1][:, :N * Wb].reshape(B, N, Wb), t1[1][cidx.view(B, N).long(), :Wb])
    dcat = [torch.randn(B, ld, generator=gen).to(DEV) for ld in ldc]
    outs = []
    for pair in (False, True):
        y = [t.clone() for t in t1]
        z = [torch.full((B, ops.r4(W)), 7.0, device=DEV) for W in (Wa, Wb)]
        if pair:
            L.check(lib.gi_compress_slots2_f(y[0].data_ptr(), y[0].stride(0), Wa, dcat[0].data_ptr(), ldc[0],
                                             z[0].data_ptr(), z[0].stride(0), y[1].data_ptr(), y[1].stride(0),
                                             Wb, dcat[1].data_ptr(), ldc[1], z[1].data_ptr(), z[1].stride(0),
                                             cidx.data_ptr(), B, N, S, 0, _stream()), "compress2")
        else:
            for k, W in enumerate((Wa, Wb)):
                L.check(lib.gi_compress_slots(y[k].data_ptr(), y[k].stride(0), cidx.data_ptr(), B, N, W, S,
                                              dcat[k].data_ptr(), ldc[k], z[k].data_ptr(), z[k].stride(0),
                                              _stream()), "compress")
        outs.append(y + z)
    for x, y in zip(*outs):
        assert torch.equal(x, y)


def test_selu_bwd_rows_gather():
    g = torch.Generator().manual_seed(5)
    dA = torch.randn(50, 128, generator=g)
    idx = torch.randint(0, 50, (300,), generator=g, dtype=torch.int32)
    Y = D.selu(torch.randn(300, 128, generator=g))
    buf = Y.clone().to(DEV)
    ops.selu_bwd_rows(dA.to(DEV), idx.to(DEV), buf, buf, 300, 128)
    assert rel(buf, dA[idx.long()].double() * D.selu_grad_from_out(Y.double())) < 1e-6


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W", [625, 3193, 9769])          # GDB-13 / ZINC-shaped / ChEMBL-shaped APD widths:
@pytest.mark.parametrize("tdtype", ["f32", "i8"])          # > 2048 logits runs kl_loss_kernel<., 1024>
def test_fused_kl_loss_matches_torch_expression(W, tdtype):
    """`Workflow.loss` (Workflow.py:850-858) — the fused kernel against the torch expression (`O.kl_loss`'s
    formula), both instantiations (rows up to / wider than 2 048 logits), fp32 and int8 (HDF dtype) targets."""
    from graphinvent_amd.loss import apd_kl_loss, apd_kl_loss_torch
    import oracle.ggnn_oracle as O
    g = torch.Generator().manual_seed(6 + W)
    B = 257 if W == 625 else 67
    out = (torch.randn(B, W, generator=g) * 3).to(DEV)
    tgt = torch.randint(0, 4, (B, W), generator=g)
    tgt = tgt * (torch.rand(B, W, generator=g) < 0.02)       # sparse like real APDs
    tgt[:, 0] += 1                                           # every row has mass
    tgt_ref = tgt.float().to(DEV)
    tgt = tgt.to(torch.int8).to(DEV) if tdtype == "i8" else tgt_ref
    a = out.clone().requires_grad_(True)
    b = out.clone().requires_grad_(True)
    la, lb = apd_kl_loss(a, tgt), apd_kl_loss_torch(b, tgt_ref)
    (2.5 * la).backward()
    (2.5 * lb).backward()
    assert abs(float(la) - float(lb)) < 1e-5 * abs(float(lb))
    assert rel(a.grad, b.grad) < 1e-5
    # the oracle's restatement of the loss on CPU (what every model-level parity test compares with)
    lo = O.kl_loss(out.cpu(), tgt_ref.cpu())
    assert abs(float(la) - float(lo)) < 1e-5 * abs(float(lo))
    tz = tgt.clone(); tz[3] = 0                              # all-zero row: NaN like the reference
    tz_ref = tgt_ref.clone(); tz_ref[3] = 0
    assert torch.isnan(apd_kl_loss(out, tz)) and torch.isnan(apd_kl_loss_torch(out, tz_ref))
    with torch.no_grad():
        assert abs(float(apd_kl_loss(out[:3], tz[:3])) - float(apd_kl_loss_torch(out[:3], tz_ref[:3]))) < 1e-5
    # empty batch: NaN (0 / 0 like the torch expression), never uninitialised memory
    assert torch.isnan(apd_kl_loss(out[:0], tgt[:0]))


def test_fused_adam_matches_torch_adam():
    from graphinvent_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(7)
    shapes = [(250, 100), (250,), (3, 7), (1,), (500, 685)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=3e-3, weight_decay=1e-2)
    o_mine = FusedAdam(mine, lr=3e-3, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.OneCycleLR(o_mine, max_lr=3e-3, total_steps=20)
    sched_ref = torch.optim.lr_scheduler.OneCycleLR(o_ref, max_lr=3e-3, total_steps=20)
    for step in range(6):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g).to(DEV)
            p.grad = gr.clone()
            q.grad = gr.clone() if step % 2 else gr.clone()
        o_ref.step(); o_mine.step(); sched.step(); sched_ref.step()
    for p, q in zip(ref, mine):
        assert rel(q, p) < 2e-6
    assert mine[0].data_ptr() + 4 * 25000 == mine[1].data_ptr()        # flat, 16-byte segments
    # checkpoint / resume: the state_dict has torch.optim.Adam's layout (moments + step count) and a
    # fresh FusedAdam loaded from it continues exactly like the one that kept running
    sd = o_mine.state_dict()
    sd_ref = o_ref.state_dict()
    assert set(sd["state"]) == set(sd_ref["state"])
    for k in sd_ref["state"]:
        assert int(sd["state"][k]["step"]) == int(sd_ref["state"][k]["step"]) == 6
        assert rel(sd["state"][k]["exp_avg"], sd_ref["state"][k]["exp_avg"]) < 2e-6
        assert rel(sd["state"][k]["exp_avg_sq"], sd_ref["state"][k]["exp_avg_sq"]) < 2e-6
    resumed = [torch.nn.Parameter(q.detach().clone()) for q in mine]
    o_res = FusedAdam(resumed, lr=3e-3, weight_decay=1e-2)
    o_res.load_state_dict(sd)
    for p, q, r in zip(ref, mine, resumed):
        gr = torch.randn(p.shape, generator=g).to(DEV)
        p.grad, q.grad, r.grad = gr.clone(), gr.clone(), gr.clone()
    o_ref.step(); o_mine.step(); o_res.step()
    for p, q, r in zip(ref, mine, resumed):
        assert torch.equal(q, r) and rel(q, p) < 2e-6
    with pytest.raises(RuntimeError):
        o_mine.add_param_group({"params": [torch.nn.Parameter(torch.zeros(4, device=DEV))]})
