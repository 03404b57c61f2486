"""
``gnn.mpnn.GGNN`` — MI355X-native gated-graph neural network with the reference's model-class API.

Boundary kept from the reference: ``GGNN(constants)`` (gnn/mpnn.py:229-282) builds
the same sub-modules in the same registration order with the same ``state_dict`` keys and the same
RNG consumption; ``forward(nodes[B,N,Fn], edges[B,N,N,Fe]) -> logits[B, N*A + N*Fe + 1]``
(gnn/summation_mpnn.py:80-149, gnn/mpnn.py:284-303), differentiable w.r.t. the parameters, usable
under ``train()/eval()/no_grad()``, ``.to("cuda")``, ``deepcopy`` and ``load_state_dict``.

What differs is everything underneath: the whole forward is ONE call into hand-written HIP
(``gi_ggnn_forward``) and the whole backward another (``gi_ggnn_backward``), on torch's current
stream, through the C ABI of ``include/graphinvent_amd.h``.  There is no eager / CPU fallback: a
missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from collections import namedtuple
from typing import List

import torch

try:                                        # imported as graphinvent_amd.gnn.mpnn
    from .. import lib as _L
    from .. import ops as _ops
    from . import modules as _modules
except ImportError:                         # imported as top-level `gnn.mpnn` (drop-in layout)
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    from graphinvent_amd import lib as _L
    from graphinvent_amd import ops as _ops
    from graphinvent_amd.gnn import modules as _modules


def _dims_from_constants(c, B: int, kind: int = _L.KIND_GGNN) -> "_L.GgnnDims":
    d = _L.GgnnDims()
    d.kind = kind
    d.B, d.N, d.Fn, d.Fe = B, c.max_n_nodes, c.n_node_features, c.n_edge_features
    d.H, d.M, d.G = c.hidden_node_features, c.message_size, c.gather_width
    d.A, d.C, d.passes = c.len_f_add_per_node, c.len_f_conn_per_node, c.message_passes
    if kind == _L.KIND_ATTGGNN:         # per-bond-type message + energy MLPs (gnn/mpnn.py:319-335)
        d.enn_depth, d.enn_hidden = c.msg_depth, c.msg_hidden_dim
        d.eatt_depth, d.eatt_hidden = c.att_depth, c.att_hidden_dim
    else:
        d.enn_depth, d.enn_hidden = c.enn_depth, c.enn_hidden_dim
    d.att_depth, d.att_hidden = c.gather_att_depth, c.gather_att_hidden_dim
    d.emb_depth, d.emb_hidden = c.gather_emb_depth, c.gather_emb_hidden_dim
    d.mlp1_depth, d.mlp1_hidden = c.mlp1_depth, c.mlp1_hidden_dim
    d.mlp2_depth, d.mlp2_hidden = c.mlp2_depth, c.mlp2_hidden_dim
    d.big_positive = float(c.big_positive)
    return d


def _ptr_table(tensors) -> "C.Array":
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _set_dropout(dims, c, kind: int, seed: int) -> None:
    """AlphaDropout training mode (gnn/modules.py:130-142): per-stack probabilities + the mask seed."""
    attn = kind == _L.KIND_ATTGGNN
    dims.dropout = 1
    dims.drop_enn = float(c.msg_dropout_p if attn else c.enn_dropout_p)
    dims.drop_eatt = float(c.att_dropout_p) if attn else 0.0
    dims.drop_att, dims.drop_emb = float(c.gather_att_dropout_p), float(c.gather_emb_dropout_p)
    dims.drop_mlp1, dims.drop_mlp2 = float(c.mlp1_dropout_p), float(c.mlp2_dropout_p)
    dims.drop_seed = int(seed) & 0xFFFFFFFFFFFFFFFF


def ggnn_forward_raw(consts, nodes, edges, params, kind: int = _L.KIND_GGNN, dropout_seed=None,
                     bounds=None, p0_cache=None, sticky_err=None, want_backward=False, guard=None, no_x2=False,
                     wcache=None):
    """graph_compact + the fused forward.  Returns (logits, tape); the tape
    (dims, CompactGraph, workspace, per-type edge counts) is what backward consumes.

    ``dropout_seed`` (an int): training mode with AlphaDropout p > 0 — the graph is compacted without
    row sharing (every edge and every padded slot draws its own mask, as in the reference), every
    activation gets a stored backward factor (workspace x2), and the logits tensor carries B extra
    rows for the factors of the logits (the returned tensor is the view of the first B).

    ``bounds = (e_bound, d0_bound)``: the HOST-SYNC-FREE forward (inference only) — nothing is read back from the
    device; buffers and launch grids are sized for the bounds (``ops.default_bounds``), the real graph sizes stay
    on the device where every kernel reads them (``gi_compact_bound``, ``gi_graph.bounded``).  For callers that
    cannot prefetch the compaction because they mutate ``nodes`` / ``edges`` in place between forwards
    (``GraphGenerator.build_graphs``, GraphGenerator.py:118-157).  The tape cannot feed a backward.  ``sticky_err``:
    a 1-element int32 CUDA tensor that accumulates the error bits of every bounded forward (``gi_compact_bound``).

    ``p0_cache`` (an int32 CUDA tensor of ``gi_p0_cache_words`` words, zero-filled whenever the weights change):
    the pass-0 row cache of an inference loop (``gi_graph.p0_cache``); the tape cannot feed a backward.

    ``want_backward``: a backward will follow — the weight images it needs (W^T of the 16-bit-pipe layers, the dZ
    chains' image) are packed NOW on the side stream, which idles during a forward (``GI_RUN_PREPACK_BWD``); the tape
    remembers, and ``ggnn_backward_raw`` waits for them instead of packing in front of its first launches.
    ``guard = (counters, host_flag_dev_ptr)``: the fp16x2 dynamic-range guard (``gi_graph.x2_guard``);
    ``no_x2``: this forward (and its backward) as bf16x3 splits (``GI_RUN_NO_X2``).
    ``wcache = state`` (a dict with ``buf`` / ``valid``, see ``_FusedMPNN._weights_cache``): the weights-only data of the
    forward (fp16x2 chain image, max |W| cells) live in ``buf`` across forwards and are re-derived only when ``valid`` is
    False; the tape cannot feed a backward."""
    lib = _L.load()
    if bounds is not None:
        return _forward_bounded(lib, consts, nodes, edges, params, kind, bounds, p0_cache, sticky_err, guard, no_x2,
                                wcache)
    drop = dropout_seed is not None
    nodes, lay, gfix, S, E, U, D0, Ut = _ops.compact_count(nodes, edges, nodedup=drop,
                                                           allow_multi_bond=kind == _L.KIND_GGNN)
    attn = kind != _L.KIND_GGNN
    B = nodes.shape[0]
    dims = _dims_from_constants(consts, B, kind)
    if drop:
        _set_dropout(dims, consts, kind, dropout_seed)
    if lib.gi_ggnn_num_params(C.byref(dims)) != len(params):
        raise RuntimeError("parameter table does not match the model dimensions")
    for p in params:
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
            raise RuntimeError("GGNN parameters must be contiguous fp32 CUDA tensors "
                               "(call model.to('cuda'))")
    dev = nodes.device
    n_ws = lib.gi_ggnn_workspace_floats(C.byref(dims), S, E, U, D0)
    if n_ws < 0:
        _L.check(int(n_ws), "gi_ggnn_workspace_floats")
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    ldhx = lib.gi_ggnn_ldhx(C.byref(dims))
    hx0 = ws[lib.gi_ggnn_hx0_offset(C.byref(dims), S, E, U, D0):]
    graph = _ops.compact_fill(nodes, lay, gfix, S, E, U, D0, Ut, hx0, ldhx, dims.H, class_csr=attn)
    apd = dims.N * dims.A + dims.N * dims.C + 1
    out = torch.empty((2 * B if drop else B, apd), dtype=torch.float32, device=dev)
    gs = graph.c_struct()
    if p0_cache is not None and not drop:
        gs.p0_cache = p0_cache.data_ptr()
    if guard is not None:
        gs.x2_guard, gs.x2_guard_host = guard[0].data_ptr(), guard[1]
        graph.x2_guard = guard[0]                           # (the backward counts its dZ rows there too)
    flags = _L.RUN_NO_X2 if no_x2 else 0
    side = _side_stream(dev) if (PREPACK_SIDE and not drop) else 0
    if want_backward and not drop:
        flags |= _L.RUN_PREPACK_BWD
    if wcache is not None and not drop and not want_backward:
        gs.wcache, gs.wcache_valid = wcache["buf"].data_ptr(), int(bool(wcache["valid"]))
    _L.check(lib.gi_ggnn_forward_ex(C.byref(dims), _ptr_table(params), C.byref(gs), ws.data_ptr(),
                                    out.data_ptr(), apd, torch.cuda.current_stream(dev).cuda_stream, side, flags),
             "gi_ggnn_forward")
    if gs.wcache:
        wcache["valid"] = True                              # (derived on this stream, in order before any later forward)
        graph.no_backward = True
    # (no ws.record_stream(side stream): gi_ggnn_forward_ex orders the side stream's packs into ws before everything
    # enqueued on the current stream after it, so a tape dropped without a backward may free ws at once)
    # what the backward of THIS tape must repeat: the run flags and the process-wide arithmetic switches of the forward
    graph.run_flags = flags
    graph.modes = (lib.gi_bf3_enable(-1), lib.gi_x2_enable(-1), lib.gi_b3p_enable(-1))
    return (out[:B] if drop else out), (dims, graph, ws)


def _forward_bounded(lib, consts, nodes, edges, params, kind, bounds, p0_cache=None, sticky_err=None, guard=None,
                     no_x2=False, wcache=None):
    """The host-sync-free forward.  Runs under the SAME fp16x2 dynamic-range guard and the same trip state as the
    blocking forward (round-5 advisor: it used to call plain gi_ggnn_forward, so after a trip the blocking forwards ran
    bf16x3 while these kept running unguarded fp16x2 — and the two are documented to agree bit for bit)."""
    if nodes.dim() != 3:
        raise ValueError("nodes must be [B, N, Fn]")
    B, N = nodes.shape[0], nodes.shape[1]
    attn = kind != _L.KIND_GGNN
    dims = _dims_from_constants(consts, B, kind)
    if lib.gi_ggnn_num_params(C.byref(dims)) != len(params):
        raise RuntimeError("parameter table does not match the model dimensions")
    e_bound, d0_bound = bounds
    dev = nodes.device
    box = {}

    def lay_ws(S_b, E_b, U_b, D0_b):
        n_ws = lib.gi_ggnn_workspace_floats(C.byref(dims), S_b, E_b, U_b, D0_b)
        if n_ws < 0:
            _L.check(int(n_ws), "gi_ggnn_workspace_floats")
        ws = box["ws"] = torch.empty(n_ws, dtype=torch.float32, device=dev)
        return ws[lib.gi_ggnn_hx0_offset(C.byref(dims), S_b, E_b, U_b, D0_b):], \
            lib.gi_ggnn_ldhx(C.byref(dims)), dims.H

    graph, _ = _ops.compact_bounded(nodes, edges, lay_ws, e_bound, d0_bound, class_csr=attn, sticky_err=sticky_err)
    apd = dims.N * dims.A + dims.N * dims.C + 1
    out = torch.empty((B, apd), dtype=torch.float32, device=dev)
    gs = graph.c_struct()
    if p0_cache is not None:
        gs.p0_cache = p0_cache.data_ptr()
    if guard is not None:
        gs.x2_guard, gs.x2_guard_host = guard[0].data_ptr(), guard[1]
    if wcache is not None:
        gs.wcache, gs.wcache_valid = wcache["buf"].data_ptr(), int(bool(wcache["valid"]))
        wcache["valid"] = True
    # (no side stream: everything in line on the caller's stream — this forward may be recorded into a hipGraph)
    _L.check(lib.gi_ggnn_forward_ex(C.byref(dims), _ptr_table(params), C.byref(gs), box["ws"].data_ptr(),
                                    out.data_ptr(), apd, torch.cuda.current_stream(dev).cuda_stream, 0,
                                    _L.RUN_NO_X2 if no_x2 else 0),
             "gi_ggnn_forward (bounded)")
    return out, (dims, graph, box["ws"])


_SIDE_STREAMS = {}
_SIDE_STREAM_OBJS = {}
#: False keeps the whole backward on one stream (bench.py's one_stream measurement; no env knob)
WGRAD_SIDE_STREAM = True
#: False (environment GI_PREPACK=0): the forward packs nothing ahead and enqueues the weights' amax pass on its own
#: stream — the round-4 schedule (A/B aid)
import os as _os_env
PREPACK_SIDE = _os_env.environ.get("GI_PREPACK", "1") != "0"


def _side_stream_obj(device: torch.device, handle: int):
    """torch's view of the side stream (for ``Tensor.record_stream``)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), handle)
    st = _SIDE_STREAM_OBJS.get(key)
    if st is None:
        st = _SIDE_STREAM_OBJS[key] = torch.cuda.ExternalStream(handle, device=device)
    return st


def _side_stream(device: torch.device) -> int:
    """Second HIP stream (one per device) on which gi_ggnn_backward runs the weight-gradient GEMMs
    concurrently with the dZ chain; ``mpnn.WGRAD_SIDE_STREAM = False`` keeps everything on one stream."""
    if not WGRAD_SIDE_STREAM:
        return 0
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        handle = C.c_void_p()
        with torch.cuda.device(key):
            _L.check(_L.load().gi_side_stream_create(C.byref(handle)), "gi_side_stream_create")
        st = _SIDE_STREAMS[key] = handle.value      # lives as long as the process
    return st


def grad_bucket_layout(params):
    """(offsets, total) of the flat gradient bucket: state_dict order, every segment 16-byte aligned."""
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) & ~3
    return offs, total


def new_grad_bucket(params, device):
    """A flat fp32 gradient bucket and its per-parameter views."""
    offs, total = grad_bucket_layout(params)
    gflat = torch.zeros(total, dtype=torch.float32, device=device)   # (alignment gaps stay 0, not garbage)
    grads = [gflat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]
    return gflat, grads, offs


def ggnn_backward_raw(tape, out, d_out, params, early_hook=None, bucket=None):
    """The fused backward; consumes the tape's activations in place.  Returns (grads, gflat):
    per-parameter gradient views into ONE flat fp32 buffer (state_dict order, 16-byte aligned
    segments) — the bucket a data-parallel all-reduce operates on.

    ``early_hook(gflat, split, ready_event)`` (optional, set by ``dp.DataParallel``): the backward is
    issued in two calls; after the first, ``gflat[split:]`` — the readout's gradients, ~86 % of the
    bucket — is complete once ``ready_event`` fires, and the hook may start exchanging it while the
    second call differentiates the message passes."""
    lib = _L.load()
    dims, graph, ws = tape
    if getattr(graph, "no_backward", False):
        raise RuntimeError("this tape comes from a forward that used the weights cache (inference): it cannot feed a backward")
    d_out = d_out.contiguous().float()
    dev = out.device
    if dev.index != torch.cuda.current_device():     # autograd may run backward on another device
        with torch.cuda.device(dev):
            return ggnn_backward_raw(tape, out, d_out, params, early_hook, bucket)
    gs = graph.c_struct()
    n_slab = lib.gi_ggnn_slab_floats(C.byref(dims), graph.S, graph.U, gs.Ut)
    if n_slab < 0:
        _L.check(int(n_slab), "gi_ggnn_slab_floats")
    slabs = torch.empty(max(int(n_slab), 4), dtype=torch.float32, device=dev)
    gflat, grads, offs = bucket if bucket is not None else new_grad_bucket(params, dev)
    main = torch.cuda.current_stream(dev)
    # The weight-gradient GEMMs overlap the dZ chain on a second stream (gi_ggnn_backward holds them back
    # while the node-level dgrad launches fill the device by themselves).
    side = _side_stream(dev)
    args = (C.byref(dims), _ptr_table(params), C.byref(gs), ws.data_ptr(), slabs.data_ptr(),
            out.data_ptr(), out.stride(0), d_out.data_ptr(), d_out.stride(0), _ptr_table(grads),
            main.cuda_stream, side)
    # the forward's arithmetic: its run flags, and the process-wide switches as they stood then (toggling one between
    # a forward and its backward would make the backward read amax cells / images nobody wrote)
    fl = getattr(graph, "run_flags", 0)
    bits = (_L.BWD_PREPACKED if fl & _L.RUN_PREPACK_BWD else 0) | (_L.BWD_NO_X2 if fl & _L.RUN_NO_X2 else 0)
    guard = getattr(graph, "x2_guard", None)
    if guard is not None:
        gs.x2_guard = guard.data_ptr()
    modes = getattr(graph, "modes", None)
    now = (lib.gi_bf3_enable(-1), lib.gi_x2_enable(-1), lib.gi_b3p_enable(-1))
    restore = modes is not None and tuple(modes) != now
    if restore:
        lib.gi_bf3_enable(modes[0]); lib.gi_x2_enable(modes[1]); lib.gi_b3p_enable(modes[2])
    try:
        if early_hook is None:
            _L.check(lib.gi_ggnn_backward_phase(*args, _L.BWD_ALL | bits), "gi_ggnn_backward")
            return grads, gflat
        _L.check(lib.gi_ggnn_backward_phase(*args, _L.BWD_READOUT | bits), "gi_ggnn_backward(readout)")
        ready = torch.cuda.Event()
        ready.record(_side_stream_obj(dev, side) if side else main)
        early_hook(gflat, offs[lib.gi_ggnn_first_readout_param(C.byref(dims))], ready)
        _L.check(lib.gi_ggnn_backward_phase(*args, _L.BWD_PASSES | bits), "gi_ggnn_backward(passes)")
        return grads, gflat
    finally:
        if restore:
            lib.gi_bf3_enable(now[0]); lib.gi_x2_enable(now[1]); lib.gi_b3p_enable(now[2])


class _GGNNFunction(torch.autograd.Function):
    """forward = gi_compact_* + gi_ggnn_forward; backward = gi_ggnn_backward."""

    @staticmethod
    def forward(ctx, owner, nodes, edges, *params):
        out, tape = ggnn_forward_raw(owner.constants, nodes, edges, params, owner._KIND,
                                     owner._next_dropout_seed(), want_backward=True,
                                     guard=owner._x2_guard_state(nodes.device), no_x2=owner._x2_off())
        ctx.owner = owner
        ctx.tape = tape
        ctx.save_for_backward(out, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.tape is None:
            raise RuntimeError("GGNN backward called twice: the HIP backward consumes the saved "
                               "activations in place (retain_graph is not supported)")
        tape, ctx.tape = ctx.tape, None
        out, *params = ctx.saved_tensors
        # no early exchange on this path: AccumulateGrad may clone these gradients (or run user
        # hooks on them) while a collective would still be reducing the bucket in place
        grads, gflat = ggnn_backward_raw(tape, out, d_out, params, None)
        ctx.owner._grad_bucket = gflat          # the flat bucket graphinvent_amd.dp all-reduces
        return (None, None, None, *grads)


class _GGNNDirect(torch.autograd.Function):
    """Same computation as ``_GGNNFunction`` with ONE autograd input instead of 104: the parameters
    do not travel through the autograd graph; ``backward`` writes their gradients straight into
    ``param.grad`` (views of the module's flat bucket), as 104 ``AccumulateGrad`` nodes would have.
    The host-side cost of a training step drops by ~0.7 ms (of ~2.6 ms: at 2.7 ms GPU time per step
    the HOST was the bottleneck).  ``anchor`` is a dummy scalar that makes autograd record the node."""

    @staticmethod
    def forward(ctx, owner, nodes, edges, anchor):
        params = owner._params()
        seed = owner._next_dropout_seed()
        cache = owner._pass0_cache(params, nodes) if anchor is None and seed is None else None
        wcache = owner._weights_cache(params, nodes) if anchor is None and seed is None else None
        out, tape = ggnn_forward_raw(owner.constants, nodes, edges, params, owner._KIND, seed, None, cache,
                                     want_backward=anchor is not None, guard=owner._x2_guard_state(nodes.device),
                                     no_x2=owner._x2_off(), wcache=wcache)
        ctx.owner = owner
        ctx.tape = tape
        # the parameters are not saved tensors here: remember their versions so that an in-place
        # update between forward and backward is detected like autograd would
        ctx.versions = [p._version for p in params] if anchor is not None else None
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.tape is None:
            raise RuntimeError("GGNN backward called twice: the HIP backward consumes the saved "
                               "activations in place (retain_graph is not supported)")
        tape, ctx.tape = ctx.tape, None
        (out,) = ctx.saved_tensors
        if ctx.versions is not None and \
                ctx.versions != [p._version for p in ctx.owner._params()]:
            raise RuntimeError("a GGNN parameter was modified in place between forward and backward: "
                               "the saved activations no longer match the weights")
        ctx.owner._backward_into_grads(tape, out, d_out)
        return None, None, None, None


class _FusedMPNN(torch.nn.Module):
    """Shared front end of the fused models: one HIP call per forward, one per backward."""

    _KIND = _L.KIND_GGNN
    #: False (default): ``loss.backward()`` fills ``param.grad`` directly from the fused backward —
    #: what the reference's training loops use (Workflow.py:785-796).  True: every parameter is an
    #: input of the autograd node, for ``torch.autograd.grad(.., model.parameters())``, parameter
    #: hooks and autograd's in-place-modification checks (slower host side).
    autograd_params = False
    #: True: forwards that need no gradient (``no_grad()`` / ``eval()`` inference, generation) run HOST-SYNC-FREE —
    #: no read-back of the graph sizes, buffers sized for ``sync_free_bounds`` (None: ``ops.default_bounds``:
    #: 4 B N directed edges, 64 feature classes per bond type), real sizes on the device.  For loops that mutate
    #: ``nodes`` / ``edges`` in place between forwards (GraphGenerator.build_graphs) and so cannot use
    #: ``ops.prefetch_compact``.  ``last_bounded_error()`` reports a violated bound / invalid input of ANY such
    #: forward since it was last called (a device-side accumulator; one read-back — call it where you synchronise
    #: anyway, e.g. once after a whole generation loop).
    sync_free = False
    sync_free_bounds = None
    _last_bounded_graph = None
    #: True (default): forwards that need no gradient keep the first message pass's rows — a function of (bond
    #: type, 0/1 feature pattern of the source node) and the weights only — in a device-side table
    #: (``gi_p0_cache_words``) and skip that pass's stack launch when every row of the batch is already there.
    #: The table is emptied when a parameter's version, the parameter objects or ``lib.WEIGHTS_EPOCH`` (bumped by
    #: ``optim.FusedAdam.step`` and ``dp.DataParallel.broadcast_parameters``, which write through raw pointers)
    #: change; after any OTHER write that bypasses the version counter (``p.data.copy_``) call
    #: ``reset_pass0_cache()``.
    cache_pass0 = True

    #: True (default): the fp16x2 launches of this model run under the dynamic-range guard (``gi_graph.x2_guard``): every
    #: forward counts, on the device, the activation rows and weight rows / columns that the per-TENSOR fp16x2 scale
    #: would leave with fewer than ~14 significant bits (largest magnitude more than 2^24 below the tensor's).  The
    #: first such row sets a host-visible flag; from the NEXT forward on the model runs those launches as bf16x3 splits
    #: (fp32's exponent range, 9 % slower) until ``x2_guard_reset()``.  ``x2_guard_stats()`` reports the counters.
    x2_guard = True

    def _x2_guard_state(self, device):
        if not self.x2_guard or device.type != "cuda":
            return None
        st = self.__dict__.get("_x2_guard")
        if st is None:
            st = self.__dict__["_x2_guard"] = {}
        g = st.get(device)
        if g is None:
            # (counters on the device, the flag's device pointer, the flag: ops.HostFlag releases the pinned int with
            # the model — round-5 advisor: the raw allocation was never freed and made the module unpicklable)
            flag = _ops.HostFlag(device)
            g = st[device] = (torch.zeros(_L.X2_GUARD_WORDS, dtype=torch.int32, device=device), flag.dev, flag)
        return g

    def _x2_off(self) -> bool:
        """True once the guard has tripped on any device (no synchronisation: reads the mapped host flags)."""
        if self.__dict__.get("_x2_forced_off"):
            return True
        for g in (self.__dict__.get("_x2_guard") or {}).values():
            if g[2].value != 0:
                self.__dict__["_x2_forced_off"] = True
                return True
        return False

    def x2_guard_stats(self) -> dict:
        """Counters of the fp16x2 dynamic-range guard since the last reset (one read-back per device):
        ``forward_rows`` / ``weight_lines`` trip the guard, ``dgrad_rows`` is informational, ``tripped`` = the
        model now runs bf16x3 splits."""
        tot = [0, 0, 0, 0]
        for g in (self.__dict__.get("_x2_guard") or {}).values():
            tot = [a + b for a, b in zip(tot, g[0].tolist())]
        return {"forward_rows": tot[0], "weight_lines": tot[1], "dgrad_rows": tot[2], "tripped": self._x2_off()}

    def x2_guard_reset(self) -> None:
        """Clear the counters and the trip flag: fp16x2 again from the next forward on."""
        for g in (self.__dict__.get("_x2_guard") or {}).values():
            g[0].zero_()
            torch.cuda.synchronize(g[0].device)             # (no launch may set the flag after it was cleared)
            g[2].value = 0
        self.__dict__["_x2_forced_off"] = False

    def reset_pass0_cache(self) -> None:
        """Forget the cached pass-0 rows and the cached weights-only data (next no-grad forward recomputes them)."""
        for name in ("_p0_state", "_w_state"):
            st = self.__dict__.get(name)
            if st is not None:
                st["key"] = None

    def pass0_cache_stats(self) -> dict:
        """{"forwards", "hits", "rows"} of the pass-0 row cache since it was last emptied (one read-back)."""
        st = self.__dict__.get("_p0_state")
        if st is None or st.get("buf") is None:
            return {"forwards": 0, "hits": 0, "rows": 0}
        h = st["buf"][:4].tolist()
        return {"forwards": h[2], "hits": h[3], "rows": h[1]}

    def _pass0_cache(self, params, nodes):
        if not self.cache_pass0 or not nodes.is_cuda:
            return None
        st = self.__dict__.get("_p0_state")
        if st is None:
            st = self.__dict__["_p0_state"] = {"buf": None, "key": None}
        # ... and the arithmetic the rows were computed in: a tripped fp16x2 guard (bf16x3 from then on) or a flipped
        # process-wide switch must not be served rows of the other arithmetic
        key = self._cache_key(params)
        buf = st["buf"]
        if buf is None or buf.device != nodes.device:
            dims = _dims_from_constants(self.constants, nodes.shape[0], self._KIND)
            n = _L.load().gi_p0_cache_words(C.byref(dims))
            if n < 0:
                _L.check(int(n), "gi_p0_cache_words")
            buf = st["buf"] = torch.zeros(n, dtype=torch.int32, device=nodes.device)
            st["key"] = key
        elif st["key"] != key:
            buf.zero_()                              # weights changed: empty table (stream-ordered)
            st["key"] = key
        return buf

    #: True (default): forwards that need no gradient keep what they derive from the weights alone — the fp16x2 chain
    #: image, the max |W| cells, the weights' dynamic-range check — in a device buffer across calls
    #: (``gi_graph.wcache``) and re-derive it only when the weights or the arithmetic change (the pass-0 row cache's key)
    cache_weights = True

    def _cache_key(self, params):
        lib = _L.load()
        return (_L.WEIGHTS_EPOCH[0], id(params), tuple((_ops._version(p), p.data_ptr()) for p in params),
                self._x2_off(), lib.gi_bf3_enable(-1), lib.gi_x2_enable(-1))

    def _weights_cache(self, params, nodes):
        if not self.cache_weights or not nodes.is_cuda:
            return None
        st = self.__dict__.get("_w_state")
        if st is None:
            st = self.__dict__["_w_state"] = {"buf": None, "key": None, "valid": False}
        key = self._cache_key(params)
        buf = st["buf"]
        if buf is None or buf.device != nodes.device:
            dims = _dims_from_constants(self.constants, nodes.shape[0], self._KIND)
            n = _L.load().gi_ggnn_wcache_floats(C.byref(dims))
            if n < 0:
                _L.check(int(n), "gi_ggnn_wcache_floats")
            st["buf"] = torch.empty(max(int(n), 4), dtype=torch.float32, device=nodes.device)
            st["key"], st["valid"] = key, False
        elif st["key"] != key:
            st["key"], st["valid"] = key, False
        return st

    _grad_ready_hook = None
    _early_exchange_pending = False     # set by dp.DataParallel while its early all-reduce runs
    def _bounded_err_word(self, device) -> torch.Tensor:
        """The persistent device-side error accumulator of this model's sync-free forwards (one per device)."""
        words = self.__dict__.get("_bounded_err")
        if words is None:
            words = self.__dict__["_bounded_err"] = {}
        w = words.get(device)
        if w is None:
            w = words[device] = torch.zeros(1, dtype=torch.int32, device=device)
        return w

    def last_bounded_error(self) -> int:
        """Error bits (``ops.bounded_error``) of EVERY sync-free forward since the previous call, OR-ed on the device
        (``gi_compact_bound``'s ``sticky_err``): 0 = all of them valid; raises on a violation.  Reads the
        accumulator back (one synchronisation) and clears it."""
        err = 0
        for w in (self.__dict__.get("_bounded_err") or {}).values():
            bits = int(w.item())
            if bits:
                w.zero_()
                err |= bits
        if self._KIND == _L.KIND_GGNN:
            err &= ~_ops.ERR_MULTI_BOND              # several bond types on a pair = parallel edges, like the reference
        if err:
            raise ValueError("sync-free forward: " + ", ".join(
                m for bit, m in ((1, "a bond-type entry of an edge is not 0 / 1"),
                                 (2, "more edges / feature classes than sync_free_bounds"),
                                 (4, "node features are not 0/1"),
                                 (8, "an atom pair carries several bond types (AttentionGGNN)")) if err & bit))
        return 0

    def _dropout_active(self) -> bool:
        flag = self.__dict__.get("_has_dropout")
        if flag is None:                         # dropout probabilities are fixed at construction
            flag = self.__dict__["_has_dropout"] = any(
                m.dropout_p > 0 for m in self.modules() if isinstance(m, _modules.MLP))
        return self.training and flag

    #: tests: fix the mask seed of the next training-mode forwards (None: drawn from torch's CPU
    #: generator, so ``torch.manual_seed`` makes runs repeatable); the seed used last is kept in
    #: ``last_dropout_seed``
    dropout_seed = None
    last_dropout_seed = None

    def _next_dropout_seed(self):
        """None outside AlphaDropout's training mode (eval(), or every dropout_p == 0)."""
        if not self._dropout_active():
            return None
        seed = self.dropout_seed
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self.last_dropout_seed = seed
        return seed

    def _params(self) -> List[torch.nn.Parameter]:
        cache = self.__dict__.get("_param_cache")
        term = self.APDReadout.fTermNet2.seq
        if cache is None or cache[0] is not getattr(self.msg_nns[0].seq, "0").weight or \
                cache[-1] is not getattr(term, str(3 * (len(term._modules) - 1))).bias:
            # (re)built when a Parameter object was replaced: load_state_dict(assign=True),
            # module surgery, parameter swapping on .to()
            cache = self.__dict__["_param_cache"] = list(self.parameters())
            self.__dict__.pop("_bucket", None)
        return cache

    #: runtime-only attributes: caches that refer to THIS module's tensors, device-side scratch, pinned host memory
    _RUNTIME_KEYS = ("_param_cache", "_bucket", "_anchor", "_grad_bucket", "_grad_ready_hook",
                     "_early_exchange_pending", "_last_bounded_graph", "_p0_state", "_w_state", "_bounded_err", "_x2_guard",
                     "_x2_forced_off")

    def __getstate__(self):
        """Whole-module pickling (``torch.save(model)``: the reference's v1.0 checkpoints are pickled modules,
        util.py:841-849) keeps parameters, buffers and configuration; the runtime state above is rebuilt on first use."""
        state = dict(self.__dict__)
        for k in self._RUNTIME_KEYS:
            state.pop(k, None)
        state["_grad_bucket"] = None
        return state

    def __deepcopy__(self, memo):
        # the caches below refer to THIS module's tensors: a copy (the RL agent / prior models,
        # Workflow.py:187-188) must rebuild its own
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        skip = self._RUNTIME_KEYS
        import copy as _copy
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in skip else _copy.deepcopy(v, memo)
        for k in ("_param_cache", "_bucket", "_anchor"):
            new.__dict__.pop(k, None)
        return new

    def forward(self, nodes: torch.Tensor, edges: torch.Tensor) -> torch.Tensor:
        params = self._params()
        if nodes.is_cuda and nodes.device.index != torch.cuda.current_device():
            # every launch goes to the current stream of the CURRENT device: make that the inputs'
            with torch.cuda.device(nodes.device):
                return self.forward(nodes, edges)
        if self.sync_free and nodes.is_cuda and not self._dropout_active() and \
                not (torch.is_grad_enabled() and any(p.requires_grad for p in params)):
            if torch.cuda.is_current_stream_capturing():         # the only forward that can be recorded (no read-back)
                import graphinvent_amd as _pkg                 # (works in the top-level `gnn.mpnn` layout too)
                _pkg.assert_graph_safe()
                # persistent device state must exist BEFORE the capture: allocated inside it, its zero-fill would be
                # recorded and every replay would clear what the previous ones accumulated (round-4 advisor finding)
                if nodes.device not in (self.__dict__.get("_bounded_err") or {}) or \
                        (self.cache_pass0 and (self.__dict__.get("_p0_state") or {}).get("buf") is None) or \
                        (self.x2_guard and nodes.device not in (self.__dict__.get("_x2_guard") or {})) or \
                        (self.cache_weights and not (self.__dict__.get("_w_state") or {}).get("valid")):
                    raise RuntimeError("run one sync-free forward of this model outside the capture first: its sticky "
                                       "error word, pass-0 row cache and fp16x2 guard counters are allocated (and "
                                       "zero-filled) on first use")
            bounds = self.sync_free_bounds or _ops.default_bounds(nodes.shape[0], nodes.shape[1], edges.shape[3])
            out, tape = ggnn_forward_raw(self.constants, nodes, edges, params, self._KIND, None, bounds,
                                         self._pass0_cache(params, nodes), self._bounded_err_word(nodes.device),
                                         guard=self._x2_guard_state(nodes.device), no_x2=self._x2_off(),
                                         wcache=self._weights_cache(params, nodes))
            self.__dict__["_last_bounded_graph"] = tape[1]
            return out
        if self.autograd_params:
            self._grad_bucket = None
            return _GGNNFunction.apply(self, nodes, edges, *params)
        anchor = None
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            anchor = self.__dict__.get("_anchor")
            if anchor is None:
                anchor = self.__dict__["_anchor"] = torch.zeros((), requires_grad=True)
        return _GGNNDirect.apply(self, nodes, edges, anchor)

    def _backward_into_grads(self, tape, out, d_out) -> None:
        """Run the fused backward and accumulate into ``param.grad`` like autograd would."""
        params = self._params()
        fresh = all(p.grad is None for p in params)
        bucket = None
        if fresh:       # nobody holds gradients: write into the module's persistent flat bucket
            bucket = self.__dict__.get("_bucket")
            if bucket is None or bucket[0].device != out.device or \
                    len(bucket[1]) != len(params) or \
                    any(g.shape != p.shape for g, p in zip(bucket[1], params)):
                bucket = self.__dict__["_bucket"] = new_grad_bucket(params, out.device)
        hook = self._grad_ready_hook if fresh else None      # early exchange needs the fresh bucket
        if not fresh and getattr(self, "_early_exchange_pending", False):
            raise RuntimeError("a second backward through the model while the data-parallel early "
                               "all-reduce of the first one is reducing the gradient bucket in place; "
                               "use DataParallel(overlap=False) for multiple backwards per step")
        grads, gflat = ggnn_backward_raw(tape, out, d_out, params, hook, bucket)
        with torch.no_grad():
            if fresh:
                for p, g in zip(params, grads):
                    if p.requires_grad:
                        p.grad = g
                self._grad_bucket = gflat    # the flat bucket graphinvent_amd.dp all-reduces
            else:                            # gradient accumulation across backward calls
                for p, g in zip(params, grads):
                    if p.requires_grad:
                        if p.grad is None:
                            p.grad = g
                        else:
                            p.grad.add_(g)
                self._grad_bucket = None

    def _build_update_and_readout(self, c) -> None:
        """GRU cell, graph gather and APD readout — identical in GGNN and AttentionGGNN
        (gnn/mpnn.py:249-282 and :337-368)."""
        self.gru = torch.nn.GRUCell(input_size=c.message_size, hidden_size=c.hidden_node_features,
                                    bias=True)
        self.gather = _modules.GraphGather(
            node_features=c.n_node_features, hidden_node_features=c.hidden_node_features,
            out_features=c.gather_width, att_depth=c.gather_att_depth,
            att_hidden_dim=c.gather_att_hidden_dim, att_dropout_p=c.gather_att_dropout_p,
            emb_depth=c.gather_emb_depth, emb_hidden_dim=c.gather_emb_hidden_dim,
            emb_dropout_p=c.gather_emb_dropout_p, big_positive=c.big_positive)
        self._grad_bucket = None    # flat gradient buffer of the latest backward (transient)
        self.APDReadout = _modules.GlobalReadout(
            node_emb_size=c.hidden_node_features, graph_emb_size=c.gather_width,
            mlp1_hidden_dim=c.mlp1_hidden_dim, mlp1_depth=c.mlp1_depth,
            mlp1_dropout_p=c.mlp1_dropout_p, mlp2_hidden_dim=c.mlp2_hidden_dim,
            mlp2_depth=c.mlp2_depth, mlp2_dropout_p=c.mlp2_dropout_p,
            f_add_elems=c.len_f_add_per_node, f_conn_elems=c.len_f_conn_per_node, f_term_elems=1,
            max_n_nodes=c.max_n_nodes, device=c.device)


class GGNN(_FusedMPNN):
    """The "gated-graph neural network" model (gnn/mpnn.py:229-303) on MI355X HIP kernels."""

    def __init__(self, constants: namedtuple) -> None:
        super().__init__()
        c = constants
        # attributes SummationMPNN.__init__ caches (gnn/summation_mpnn.py:14-22)
        self.hidden_node_features = c.hidden_node_features
        self.edge_features = c.n_edge_features
        self.message_size = c.message_size
        self.message_passes = c.message_passes
        self.constants = c

        # registration order of gnn/mpnn.py:238-282 (fixes state_dict order and RNG consumption)
        self.msg_nns = torch.nn.ModuleList()
        for _ in range(c.n_edge_features):
            self.msg_nns.append(_modules.MLP(c.hidden_node_features,
                                             [c.enn_hidden_dim] * c.enn_depth, c.message_size,
                                             c.enn_dropout_p))
        self._build_update_and_readout(c)


class AttentionGGNN(_FusedMPNN):
    """The "GGNN with attention" model (gnn/mpnn.py:306-398; ``AggregationMPNN.forward``,
    gnn/aggregation_mpnn.py:83-168) on MI355X HIP kernels: per bond type one message MLP and one
    energy MLP on the neighbour's hidden state, softmax over each node's incoming edges
    (``gi_seg_softmax_fwd``) instead of the reference's max-degree-padded masked softmax."""

    _KIND = _L.KIND_ATTGGNN

    def __init__(self, constants: namedtuple) -> None:
        super().__init__()
        c = constants
        # attributes AggregationMPNN.__init__ caches (gnn/aggregation_mpnn.py:17-21)
        self.hidden_node_features = c.hidden_node_features
        self.edge_features = c.n_edge_features
        self.message_size = c.message_size
        self.message_passes = c.message_passes
        self.constants = c

        # both ModuleLists are registered first (state_dict: every msg_nns.* before att_nns.*), the
        # MLPs are constructed interleaved (RNG order msg_0, att_0, msg_1, ...): gnn/mpnn.py:316-335
        self.msg_nns = torch.nn.ModuleList()
        self.att_nns = torch.nn.ModuleList()
        for _ in range(c.n_edge_features):
            self.msg_nns.append(_modules.MLP(c.hidden_node_features,
                                             [c.msg_hidden_dim] * c.msg_depth, c.message_size,
                                             c.msg_dropout_p))
            self.att_nns.append(_modules.MLP(c.hidden_node_features,
                                             [c.att_hidden_dim] * c.att_depth, c.message_size,
                                             c.att_dropout_p))
        self._build_update_and_readout(c)
