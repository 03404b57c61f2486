"""
Thin torch-tensor wrappers over the C ABI (``graphinvent_amd/lib.py``).  Every function enqueues
on torch's current HIP stream and takes/returns CUDA tensors; nothing here computes on the host.
"""
from __future__ import annotations

import ctypes as C
import time
import weakref
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import lib as L


def _stream(t: Optional[torch.Tensor] = None) -> int:
    """The current stream of `t`'s device (of the current device without `t`)."""
    return torch.cuda.current_stream(t.device if t is not None else None).cuda_stream


def _on_device_of(argname: str):
    """Run the wrapped op with the device of its tensor argument `argname` current: kernels go to the current
    device's stream, which must be the device the tensors live on (multi-GPU processes, advisor finding)."""
    import functools
    import inspect

    def deco(fn):
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            t = sig.bind(*args, **kwargs).arguments.get(argname)
            if isinstance(t, (list, tuple)) and t:
                t = t[0]
            if isinstance(t, dict):
                t = next((v for v in t.values() if torch.is_tensor(v)), None)
            if torch.is_tensor(t) and t.is_cuda and t.device.index != torch.cuda.current_device():
                with torch.cuda.device(t.device):
                    return fn(*args, **kwargs)
            return fn(*args, **kwargs)
        return wrapper
    return deco


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: the MI355X HIP path has no CPU "
                           "fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _model_input(t: torch.Tensor, name: str):
    """Model inputs may stay int8 (the dtype of the preprocessed HDF) or be fp32 (what the
    reference's HDFDataset converts to, BlockDatasetLoader.py:139-141); returns (tensor, GI_DTYPE)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: the MI355X HIP path has no CPU "
                           "fallback")
    if t.dtype == torch.int8:
        return t.contiguous(), L.DTYPE_I8
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous(), L.DTYPE_F32


def r4(x: int) -> int:
    return (x + 3) & ~3


# ------------------------------------------------------------------------------------------------
# K1 graph_compact
# ------------------------------------------------------------------------------------------------
@dataclass
class CompactGraph:
    B: int
    N: int
    Fn: int
    Fe: int
    S: int                      # active node slots (compact rows 0..S-1; row S = zero row)
    E: int                      # directed edges
    U: int                      # message rows = distinct (source slot, bond type) pairs
    D0: int                     # pass-0 rows = present (feature class, bond type) pairs; 0 = off
    Ut: Sequence[int]           # message rows per bond type
    layout: L.CompactLayout
    gfix: torch.Tensor          # int32, fixed-size part (see include/graphinvent_amd.h)
    gvar: torch.Tensor          # int32, variable-size part: u_src | in_perm | mu_off | mu_dst | mu_slot | out_perm | d_src
    cmat: Optional[torch.Tensor] = None   # fp32 [S+1, ldc0] pass-0 edge-count matrix
    class_csr: bool = False               # e2d / cls_off / cls_edges are filled (AttentionGGNN pass 0)
    bounded: bool = False                 # S, E, U, D0, Ut are BOUNDS; the real sizes stay in gfix (compact_bounded)

    def view(self, name: str, n: int) -> torch.Tensor:
        o = getattr(self.layout, name)
        return self.gfix[o:o + n]

    def _var(self, k: int, n: int) -> torch.Tensor:
        return self.gvar[self._offs[k]:self._offs[k] + n]

    @property
    def _offs(self):
        return _gvar_offsets(self.E, self.U, self.D0)[0]

    @property
    def u_src(self): return self._var(0, self.U)
    @property
    def in_perm(self): return self._var(1, self.E)
    @property
    def mu_off(self): return self._var(2, self.U + 1)
    @property
    def mu_dst(self): return self._var(3, self.E)
    @property
    def mu_slot(self): return self._var(4, self.E)
    @property
    def out_perm(self): return self._var(5, self.U)
    @property
    def d_src(self): return self._var(6, self.D0)
    @property
    def e2d(self): return self._var(7, self.E)
    @property
    def cls_off(self): return self._var(8, self.D0 + 1)
    @property
    def cls_edges(self): return self._var(9, self.E)
    @property
    def type_off0(self): return self.view("type_off0", self.Fe + 1)
    @property
    def cidx(self): return self.view("cidx", self.B * self.N)
    @property
    def node_mask(self): return self.view("node_mask", self.B * self.N)
    @property
    def slot_of(self): return self.view("slot_of", self.S)
    @property
    def seg_off(self): return self.view("seg_off", self.S + 2)
    @property
    def src_off(self): return self.view("src_off", self.S + 2)
    @property
    def type_off(self): return self.view("type_off", self.Fe + 1)

    def c_struct(self) -> "L.Graph":
        """gi_graph for the fused model calls (keeps the host Ut array alive on the struct)."""
        g = L.Graph()
        g.S, g.E, g.U = self.S, self.E, self.U
        g.gfix = self.gfix.data_ptr()
        base, offs = self.gvar.data_ptr(), self._offs
        (g.u_src, g.in_perm, g.mu_off, g.mu_dst, g.mu_slot, g.out_perm,
         g.d_src) = (base + 4 * o for o in offs[:7])
        if self.class_csr and self.D0 > 0:
            g.e2d, g.cls_off, g.cls_edges = (base + 4 * o for o in offs[7:10])
        g.D0 = self.D0
        g.ldc0 = r4(self.D0)
        g.cmat = self.cmat.data_ptr() if self.D0 > 0 else None
        g._ut = (C.c_int * self.Fe)(*self.Ut)
        g.Ut = g._ut
        g.bounded = 1 if self.bounded else 0
        return g


def _gvar_offsets(E: int, U: int, D0: int = 0):
    """Offsets (ints, 16-byte aligned) of u_src[U], in_perm[E], mu_off[U+1], mu_dst[E], mu_slot[E],
    out_perm[U], d_src[D0], e2d[E], cls_off[D0+1], cls_edges[E] inside the variable-size index buffer,
    and its total length (the last three are filled only for AttentionGGNN's pass 0)."""
    sizes = (U, E, U + 1, E, E, U, D0, E, D0 + 1, E)        # + e2d[E], cls_off[D0+1], cls_edges[E]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += max(r4(n), 4)
    return offs, o


@_on_device_of("nodes")
def _count_launch(nodes: torch.Tensor, edges: torch.Tensor, nodedup: bool = False):
    """Enqueue gi_compact_count on the current stream; returns (nodes as the kernels read them,
    layout, gfix).  nodedup: no row sharing (AlphaDropout training mode, gi_compact_count_ex)."""
    lib = L.load()
    nodes, dt_n = _model_input(nodes, "nodes")
    edges, dt_e = _model_input(edges, "edges")
    if dt_n != dt_e:                          # mixed dtypes: promote both to fp32
        nodes, edges, dt_n = nodes.float(), edges.float(), L.DTYPE_F32
    B, N, Fn = nodes.shape
    Fe = edges.shape[3]
    if edges.shape[:3] != (B, N, N):
        raise ValueError(f"edges shape {tuple(edges.shape)} does not match nodes {tuple(nodes.shape)}")
    lay = L.CompactLayout()
    L.check(lib.gi_compact_layout(B, N, Fe, C.byref(lay)), "gi_compact_layout")
    gfix = torch.empty(lay.total_ints, dtype=torch.int32, device=nodes.device)
    L.check(lib.gi_compact_count_ex(nodes.data_ptr(), edges.data_ptr(), dt_n, B, N, Fn, Fe,
                                    gfix.data_ptr(), 1 if nodedup else 0, _stream()),
            "gi_compact_count")
    return nodes, lay, gfix, Fe


ERR_EDGE_VALUE, ERR_OVERFLOW, ERR_NODE_VALUE, ERR_MULTI_BOND = 1, 2, 4, 8     # bits of gfix counts[2]


def _unpack_counts(counts, Fe: int, allow_multi_bond: bool = True):
    S, E, err, U, D0 = counts[0], counts[1], counts[2], counts[3], counts[20]
    if err & ERR_EDGE_VALUE:
        raise ValueError("edges tensor violates the preprocessed-HDF contract: bond-type entries must be 0 or 1 "
                         "(DataProcesser.py / MolecularGraph.py edge features)")
    if (err & ERR_MULTI_BOND) and not allow_multi_bond:
        raise ValueError("an atom pair carries several bond types at once: GGNN treats them as parallel edges like the "
                         "reference does (the generation loop's dummy graph, GraphGenerator.py:133, 424-427), but "
                         "AttentionGGNN's softmax over neighbours (gnn/mpnn.py:370-389) is not defined per bond here")
    return S, E, U, D0, counts[4:4 + Fe]


# ---- compaction one batch ahead -------------------------------------------------------------------
# gi_compact_count ends in the forward pass's only host read-back (S, E, U decide buffer sizes and
# launch grids).  A caller that knows its next batch (the block loader; bench.py) can run the
# counting phase for it on a side stream while the current training step occupies the main stream;
# the forward then finds the result here, already on the host, and neither waits for the device nor
# puts the three counting kernels on its critical path.  Entries are consumed once.
#: how forward passes got their sizes so far: from a finished prefetch (no wait on the step's stream) or by a
#: blocking read-back behind everything queued on it (bench.py reports the split over its timed region)
READBACKS = {"prefetched": 0, "blocking": 0, "bounded": 0}     # "bounded": no read-back at all (compact_bounded)
#: seconds the HOST spent waiting for a prefetched count (the one place a training loop on resident or loader-fed batches
#: blocks): wall time of a loop minus this = what the host needs to enqueue it (bench.py: "host")
HOST_WAIT = [0.0]
_PREFETCHED: "dict" = {}
_PINNED: "list" = []
_PINNED_NEXT = 0
_PREFETCH_STREAMS: "dict" = {}


def _version(t: torch.Tensor) -> int:
    try:
        return t._version
    except RuntimeError:            # inference-mode tensors do not track versions
        return -1


def _batch_key(nodes: torch.Tensor, edges: torch.Tensor):
    return (nodes.data_ptr(), edges.data_ptr(), _version(nodes), _version(edges), tuple(nodes.shape),
            tuple(edges.shape), nodes.dtype, edges.dtype)


def prefetch_compact(nodes: torch.Tensor, edges: torch.Tensor,
                     stream: Optional["torch.cuda.Stream"] = None) -> None:
    """Run the counting phase of graph_compact for a FUTURE forward(nodes, edges) on `stream`
    (default: a per-device side stream that first waits for the current stream)."""
    if not nodes.is_cuda:
        return
    dev = nodes.device
    if stream is None:
        stream = _PREFETCH_STREAMS.get(dev.index)
        if stream is None:
            stream = _PREFETCH_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
    global _PINNED_NEXT
    if len(_PINNED) < 8:                                     # pool of host buffers, round robin; at
        _PINNED.append(torch.empty(L.COUNTS, dtype=torch.int32).pin_memory())   # most 4 are pending
        pinned = _PINNED[-1]
    else:
        pinned = _PINNED[_PINNED_NEXT % 8]
        _PINNED_NEXT += 1
    with torch.cuda.stream(stream):
        nodes_c, lay, gfix, Fe = _count_launch(nodes, edges)
        pinned.copy_(gfix[lay.counts:lay.counts + L.COUNTS], non_blocking=True)
        done = torch.cuda.Event()
        done.record(stream)
    while len(_PREFETCHED) >= 4:                             # never-consumed entries do not pile up
        _PREFETCHED.pop(next(iter(_PREFETCHED)))
    # weak references to the very tensor objects: an address can be recycled by the allocator, an
    # object identity cannot be confused while the entry is alive
    _PREFETCHED[_batch_key(nodes, edges)] = (nodes_c, lay, gfix, Fe, pinned, done,
                                             weakref.ref(nodes), weakref.ref(edges))


def compact_count(nodes: torch.Tensor, edges: torch.Tensor, nodedup: bool = False, allow_multi_bond: bool = True):
    """Phase 1; returns (nodes, layout, gfix, S, E, U, D0, Ut).  One host read-back of 24 ints — the only
    synchronisation point of a forward pass, unless `prefetch_compact` already ran for this batch.
    nodedup: the layout without row sharing (a prefetched, de-duplicated result is discarded)."""
    hit = _PREFETCHED.pop(_batch_key(nodes, edges), None) if _PREFETCHED else None
    if nodedup:
        hit = None
    if hit is not None and not (hit[6]() is nodes and hit[7]() is edges):
        hit = None                                           # same address, different tensors: stale
    if hit is not None:
        nodes_c, lay, gfix, Fe, pinned, done = hit[:6]
        t_wait = time.perf_counter()
        done.synchronize()                                   # normally long finished; a host that runs ahead of the device waits here
        HOST_WAIT[0] += time.perf_counter() - t_wait
        cur = torch.cuda.current_stream(nodes_c.device)
        cur.wait_event(done)
        gfix.record_stream(cur)
        nodes_c.record_stream(cur)
        READBACKS["prefetched"] += 1
        return (nodes_c, lay, gfix) + _unpack_counts(pinned.tolist(), Fe, allow_multi_bond)
    READBACKS["blocking"] += 1
    nodes, lay, gfix, Fe = _count_launch(nodes, edges, nodedup)
    counts = gfix[lay.counts:lay.counts + L.COUNTS].cpu().tolist()
    return (nodes, lay, gfix) + _unpack_counts(counts, Fe, allow_multi_bond)


@_on_device_of("nodes")
def compact_fill(nodes, lay, gfix, S, E, U, D0, Ut, hx0: torch.Tensor, ldhx: int,
                 H: int, class_csr: bool = False) -> CompactGraph:
    lib = L.load()
    B, N, Fn = nodes.shape
    Fe = len(Ut)
    offs, total = _gvar_offsets(E, U, D0)
    gvar = torch.empty(total, dtype=torch.int32, device=nodes.device)
    ldc0 = r4(D0)
    cmat = torch.empty((S + 1, ldc0), dtype=torch.float32, device=nodes.device) if D0 > 0 else None
    dt = L.DTYPE_I8 if nodes.dtype == torch.int8 else L.DTYPE_F32
    base = gvar.data_ptr()
    ptrs = [base + 4 * o for o in offs]
    class_csr = bool(class_csr and D0 > 0 and E > 0)
    L.check(lib.gi_compact_fill(nodes.data_ptr(), dt, B, N, Fn, Fe, gfix.data_ptr(), S, E, U,
                                *ptrs[:6], hx0.data_ptr(), ldhx, H, D0, ptrs[6], _ptr(cmat), ldc0,
                                ptrs[7] if class_csr else None, _stream()), "gi_compact_fill")
    if class_csr:        # pass-0 row -> edge slots CSR (AttentionGGNN's pass-0 backward)
        L.check(lib.gi_compact_class_csr(ptrs[7], E, D0, ptrs[8], ptrs[9], _stream()),
                "gi_compact_class_csr")
    return CompactGraph(B, N, Fn, Fe, S, E, U, D0, list(Ut), lay, gfix, gvar, cmat, class_csr)


def default_bounds(B: int, N: int, Fe: int):
    """(e_bound, d0_bound) of a bounded forward when the caller declares none: 4 directed edges per node slot
    on average (valence-bounded molecular graphs: sum of degrees <= 4 atoms) and 64 feature classes per bond
    type (GDB-13: 15, ZINC-shaped: 27, ChEMBL-shaped: 36)."""
    e = 4 * B * N
    return e, min(64 * Fe, e)


def compact_bounded(nodes: torch.Tensor, edges: torch.Tensor, lay_ws, e_bound: int, d0_bound: int,
                    class_csr: bool = False, sticky_err: Optional[torch.Tensor] = None):
    """graph_compact WITHOUT the host read-back: counting phase, gi_compact_bound, fill — all enqueued; every
    buffer is sized for the bounds (S <= B N, E, U <= e_bound, D0 <= d0_bound) and the real sizes never leave the
    device.  `lay_ws(S_b, E_b, U_b, D0_b)` -> (hx0 view, ldhx, H) provides the workspace slice hx0 is written to.
    Returns (CompactGraph with bounded=True, nodes as the kernels read them).  A batch that exceeds a bound or
    has non-0/1 node features is flagged in gfix counts[2] (bits 1 / 2; `bounded_error`) and produces
    meaningless logits instead of touching memory beyond the buffers.  `sticky_err` (a 1-element int32 CUDA tensor
    the caller keeps): the same bits OR-ed into it on the device, so one read-back after a loop covers every round."""
    lib = L.load()
    nodes_c, lay, gfix, Fe = _count_launch(nodes, edges)
    B, N, Fn = nodes_c.shape
    S_b, E_b, U_b = B * N, max(int(e_bound), 1), max(int(e_bound), 1)
    D0_b = max(min(int(d0_bound), U_b), 1)             # (pass-0 rows are message rows of a kind: D0 <= U)
    with torch.cuda.device(nodes_c.device):
        if sticky_err is not None and not (sticky_err.is_cuda and sticky_err.dtype == torch.int32
                                           and sticky_err.device == nodes_c.device and sticky_err.numel() >= 1):
            raise ValueError("sticky_err must be an int32 tensor on the inputs' device")
        L.check(lib.gi_compact_bound(gfix.data_ptr(), B, N, Fe, E_b, D0_b,
                                     sticky_err.data_ptr() if sticky_err is not None else None,
                                     _stream(nodes_c)), "gi_compact_bound")
        hx0, ldhx, H = lay_ws(S_b, E_b, U_b, D0_b)
        offs, total = _gvar_offsets(E_b, U_b, D0_b)
        gvar = torch.empty(total, dtype=torch.int32, device=nodes_c.device)
        ldc0 = r4(D0_b)
        cmat = torch.empty((S_b + 1, ldc0), dtype=torch.float32, device=nodes_c.device)
        dt = L.DTYPE_I8 if nodes_c.dtype == torch.int8 else L.DTYPE_F32
        ptrs = [gvar.data_ptr() + 4 * o for o in offs]
        L.check(lib.gi_compact_fill(nodes_c.data_ptr(), dt, B, N, Fn, Fe, gfix.data_ptr(), -1, E_b, U_b,
                                    *ptrs[:6], hx0.data_ptr(), ldhx, H, D0_b, ptrs[6], cmat.data_ptr(), ldc0,
                                    ptrs[7] if class_csr else None, _stream(nodes_c)), "gi_compact_fill")
    READBACKS["bounded"] += 1
    g = CompactGraph(B, N, Fn, Fe, S_b, E_b, U_b, D0_b, [U_b] * Fe, lay, gfix, gvar, cmat, bool(class_csr), True)
    return g, nodes_c


def bounded_error(graph: "CompactGraph") -> int:
    """counts[2] of a bounded compaction (one host read-back — call it when you synchronise anyway): bit 0 = an
    edge's feature vector is not one-hot, bit 1 = more edges / pass-0 rows than the bounds, bit 2 = node features
    not 0/1 (the bounded forward needs the pass-0 shortcut).  0 = the logits of that forward are valid."""
    c = graph.layout.counts
    return int(graph.gfix[c + 2].item())


def compact(nodes: torch.Tensor, edges: torch.Tensor, H: int, class_csr: bool = False,
            nodedup: bool = False):
    """Both phases; returns (CompactGraph, hx0[S+1, ldhx])."""
    nodes, lay, gfix, S, E, U, D0, Ut = compact_count(nodes, edges, nodedup)
    ldhx = r4(H + nodes.shape[2])
    hx0 = torch.empty((S + 1, ldhx), dtype=torch.float32, device=nodes.device)
    g = compact_fill(nodes, lay, gfix, S, E, U, D0, Ut, hx0, ldhx, H, class_csr)
    return g, hx0


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
@_on_device_of("W")
def bf3_pack(W: torch.Tensor, transpose: bool = False, as_f32: bool = False) -> torch.Tensor:
    """gi_bf3_pack: the three bf16 planes of a weight matrix for `gemm(..., flags=L.GEMM_BF3)`.  W [rows, cols]
    row-major gives B[n][k] = W[n][k]; with `transpose` B[n][k] = W[k][n] (dgrad of a Linear weight).  `as_f32`: the
    same matrix as plain fp32 [rows][r4(cols)] in the same buffer (for `L.GEMM_BF3 | L.GEMM_BF3B_F32`, ldb = r4(cols))."""
    lib = L.load()
    rows, cols = (W.shape[1], W.shape[0]) if transpose else (W.shape[0], W.shape[1])
    n = lib.gi_bf3_image_elems(rows, cols)
    img = torch.empty(n, dtype=torch.int16, device=W.device)
    d = L.Bf3PackDesc()
    d.W, d.rows, d.cols, d.ld, d.transpose, d.image = W.data_ptr(), rows, cols, W.stride(0), int(transpose), img.data_ptr()
    d.as_f32 = int(as_f32)
    L.check(lib.gi_bf3_pack(C.byref(d), 1, _stream()), "gi_bf3_pack")
    return img



@_on_device_of("A")
def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, *, flags=0, bias=None, act=None, ldact=0,
         a_idx=None, b_idx=None, a_major=False, b_major=False, tm=1, tn=1, grp_off=None,
         ngroups=0, max_group_rows=0, Bg=(), biasg=(), Cg=(), nsplit=1, c_split_stride=0,
         ones_col=-1, gsplit=(), a_amax=None, b_amax=None, c_amax=None, x2_guard=None, x2_guard_host=0):
    lib = L.load()
    p = L.GemmParams()
    p.A, p.B, p.C = _ptr(A), _ptr(B), _ptr(C_out)
    p.bias, p.act, p.a_idx, p.b_idx, p.grp_off = (_ptr(bias), _ptr(act), _ptr(a_idx), _ptr(b_idx),
                                                  _ptr(grp_off))
    p.M, p.N, p.K, p.lda, p.ldb, p.ldc, p.ldact = M, N, K, lda, ldb, ldc, ldact
    p.flags, p.a_major, p.b_major, p.tm, p.tn = flags, int(a_major), int(b_major), tm, tn
    p.ngroups, p.nsplit, p.max_group_rows, p.ones_col = ngroups, nsplit, max_group_rows, ones_col
    p.c_split_stride = c_split_stride
    p.a_amax, p.b_amax, p.c_amax = _ptr(a_amax), _ptr(b_amax), _ptr(c_amax)
    p.x2_guard, p.x2_guard_host = _ptr(x2_guard), (x2_guard_host or None)
    for i, t in enumerate(Bg):
        p.Bg[i] = _ptr(t)
    for i, t in enumerate(biasg):
        p.biasg[i] = _ptr(t)
    for i, t in enumerate(Cg):
        p.Cg[i] = _ptr(t)
    for i in range(ngroups):
        p.gsplit[i] = gsplit[i] if gsplit else nsplit
    L.check(lib.gi_gemm(C.byref(p), _stream()), "gi_gemm")


@_on_device_of("vals")
def seg_sum(vals, perm, off, rows, cols, out, accumulate=False):
    L.check(L.load().gi_seg_sum(vals.data_ptr(), vals.stride(0), _ptr(perm), off.data_ptr(), rows,
                                cols, out.data_ptr(), out.stride(0), int(accumulate), _stream()),
            "gi_seg_sum")


@_on_device_of("dY")
def selu_bwd_rows(dY, idx, Y, out, rows, cols):
    L.check(L.load().gi_selu_bwd_rows(dY.data_ptr(), dY.stride(0), _ptr(idx), Y.data_ptr(),
                                      Y.stride(0), out.data_ptr(), out.stride(0), rows, cols,
                                      _stream()), "gi_selu_bwd_rows")


def absmax(tensors, out: torch.Tensor) -> torch.Tensor:
    """max |x| of each 2-D tensor (row pitch = its stride) into the amax cell out[i] (fp32 [n, AMAX_WORDS], zeroed here;
    the value is ``out[i].max()``): what a GI_GEMM_X2 launch needs for operands no GEMM epilogue produced (the weights)."""
    lib = L.load()
    tensors = list(tensors)
    assert out.dim() == 2 and out.shape[1] == L.AMAX_WORDS and out.is_contiguous() and out.shape[0] >= len(tensors)
    out[:len(tensors)].zero_()
    d = (L.AbsmaxDesc * len(tensors))()
    for i, t in enumerate(tensors):
        d[i].x, d[i].rows, d[i].cols, d[i].ld = t.data_ptr(), t.shape[0], t.shape[1], t.stride(0)
        d[i].out = out[i].data_ptr()
    with torch.cuda.device(out.device):
        L.check(lib.gi_absmax(d, len(tensors), _stream(out)), "gi_absmax")
    return out


class HostFlag:
    """One int of pinned host memory that kernels can write (``gi_host_flag_create``): ``.dev`` is the pointer to hand
    to kernels (``gi_graph.x2_guard_host``), ``.value`` reads / writes it from the host without a synchronisation."""

    def __init__(self, device=None):
        host, dev = C.c_void_p(), C.c_void_p()
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            L.check(L.load().gi_host_flag_create(C.byref(host), C.byref(dev)), "gi_host_flag_create")
        self._host, self.dev = host, dev.value
        self._view = C.cast(host, C.POINTER(C.c_int))

    @property
    def value(self) -> int:
        return int(self._view[0])

    @value.setter
    def value(self, v: int) -> None:
        self._view[0] = int(v)

    def close(self) -> None:
        if self._host is not None:
            L.load().gi_host_flag_destroy(self._host)
            self._host = None
            self._view = None

    def __del__(self):
        try:
            self.close()
        except Exception:            # interpreter shutdown: the library may be gone
            pass

    def __reduce__(self):
        raise TypeError("a HostFlag wraps pinned host memory of this process and cannot be pickled")


def x2_weight_guard(tensors, cells: torch.Tensor, counter: torch.Tensor, host_flag: int = 0) -> None:
    """gi_x2_weight_guard: count in ``counter[0]`` the rows and columns of each matrix whose largest magnitude lies more
    than 2^24 below the matrix's (``cells[i]`` = its amax cell, filled by ``absmax`` on this stream before)."""
    tensors = list(tensors)
    d = (L.AbsmaxDesc * len(tensors))()
    for i, t in enumerate(tensors):
        d[i].x, d[i].rows, d[i].cols, d[i].ld = t.data_ptr(), t.shape[0], t.shape[1], t.stride(0)
        d[i].out = cells[i].data_ptr()
    with torch.cuda.device(counter.device):
        L.check(L.load().gi_x2_weight_guard(d, len(tensors), counter.data_ptr(), host_flag or None, _stream(counter)),
                "gi_x2_weight_guard")


def reduce_slabs(items):
    """items: iterable of (slabs, dW, db, slab_stride, n_slabs, N, K, ld)."""
    items = list(items)
    arr = (L.ReduceDesc * len(items))()
    for d, (slabs, dW, db, stride, n, N, K, ld) in zip(arr, items):
        d.slabs, d.dW, d.db = slabs.data_ptr(), dW.data_ptr(), _ptr(db)
        d.slab_stride, d.n_slabs, d.N, d.K, d.ld = stride, n, N, K, ld
    L.check(L.load().gi_reduce_slabs(arr, len(items), _stream()), "gi_reduce_slabs")


def ws_view(ws: torch.Tensor, dims, graph: "CompactGraph", name: str, rows: int, i: int = 0,
            j: int = 0):
    """Test/debug: a [rows, ld] view of a named workspace buffer (gi_ggnn_ws_query)."""
    off, ld = C.c_longlong(), C.c_int()
    D0 = graph.D0
    L.check(L.load().gi_ggnn_ws_query(C.byref(dims), graph.S, graph.E, graph.U, D0, name.encode(), i,
                                      j, C.byref(off),
                                      C.byref(ld)), f"gi_ggnn_ws_query({name})")
    return ws[off.value:off.value + rows * ld.value].view(rows, ld.value)


def mlp_chain(chains, backward=False, x2=False, rows32=False):
    """gi_mlp_chain (x2: the fp16x2 variant — amax cells for the weights are allocated here, gi_mlp_chain_pack fills
    them and writes the two-plane image).  chains: list (1 or 2) of dicts with keys
    X, x_idx (or None), grp_off (int32 tensor [G+1] or None), group_rows (host ints), rows,
    seg (backward, optional: dict(vals, idx, off) — gi_chain_params.seg_vals: X is formed in place) and
    layers = list of dicts(W=[per-group tensors], bias=[per-group tensors] (forward), out, act (backward
    or None), K, N)."""
    lib = L.load()
    arr = (L.ChainParams * len(chains))()
    for c, spec in zip(arr, chains):
        c.nlayers = len(spec["layers"])
        c.X, c.ldx = spec["X"].data_ptr(), spec["X"].stride(0)
        c.x_idx, c.grp_off = _ptr(spec.get("x_idx")), _ptr(spec.get("grp_off"))
        rows_g = list(spec.get("group_rows") or [spec["rows"]])
        c.ngroups, c.rows, c.backward = len(rows_g), spec["rows"], int(backward)
        seg = spec.get("seg")              # backward: X formed in place by the fused segmented sum
        if seg is not None:
            c.seg_vals, c.ld_seg = seg["vals"].data_ptr(), seg["vals"].stride(0)
            c.seg_idx, c.seg_off = seg["idx"].data_ptr(), seg["off"].data_ptr()
        for t, n in enumerate(rows_g):
            c.group_rows[t] = n
        for y, ly in zip(c.layer, spec["layers"]):
            y.K, y.N = ly["K"], ly["N"]
            y.out, y.ldo = ly["out"].data_ptr(), ly["out"].stride(0)
            act = ly.get("act")
            y.act, y.ldact = _ptr(act), (act.stride(0) if act is not None else 0)
            for t, w in enumerate(ly["W"]):
                y.W[t] = w.data_ptr()
            for t, b in enumerate(ly.get("bias") or ()):
                y.bias[t] = b.data_ptr()
            y.out_amax = _ptr(ly.get("out_amax"))            # fp16x2 kernels: max |out| of the layer (amax cell)
        c.x_amax = _ptr(spec.get("x_amax"))
    images = []
    if x2:
        for c, spec in zip(arr, chains):
            images.append(torch.empty(c.nlayers * c.ngroups, L.AMAX_WORDS, dtype=torch.float32, device=spec["X"].device))
            c.x2_wamax = images[-1].data_ptr()
            c.x2_rows32 = 1 if rows32 else 0                # the row-independent 32-row variant (gi_chain_params.x2_rows32)
    for c, spec in zip(arr, chains):       # packed weight images (kept alive until the launch is queued)
        n = lib.gi_mlp_chain_image_floats(C.byref(c))
        if n < 0:
            L.check(int(n), "gi_mlp_chain_image_floats")
        images.append(torch.empty(int(n), dtype=torch.float32, device=spec["X"].device))
        c.image = images[-1].data_ptr()
    L.check(lib.gi_mlp_chain_pack(arr, len(chains), _stream()), "gi_mlp_chain_pack")
    L.check(lib.gi_mlp_chain(arr, len(chains), _stream()), "gi_mlp_chain")
    for im in images:
        im.record_stream(torch.cuda.current_stream(im.device))
    return images       # (tests: [max |W| cells per chain ...] + [packed image per chain ...] when x2, else the images)
