"""
bench.py — training graphs/sec of the MI355X-native GGNN hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full training step of the hot path over one minibatch of synthetic molecular
graphs already resident in HBM: forward (graph_compact + message passes + readout) -> KL loss ->
backward -> gradient all-reduce (N > 1) -> Adam (FusedAdam, same math) -> OneCycleLR, in the order of
Workflow.py:785-796.  Workload = BASELINE.json configs[1]: GGNN, hidden = message = 128, 3 message
passes, GDB-13-shaped graphs (max_n_nodes 13, 5 atom types x 3 charges, 3 bond types), batch 1000
PER GPU (weak scaling).  fp32 throughout (the reference's dtype and the parity bar).

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      dominant kernel family (fp32-accurate GEMM): useful FLOP/s measured with HIP events
                around every launch on the stream it runs on (gi_prof_*), over extra profiled steps
                of the same workload, against the 157.3 TFLOP/s fp32 matrix peak; achieved = FLOP /
                SUM of the launch durations = flop_per_launch / avg_launch_us, the per-launch figure
                the rocprofv3 kernel stats under profiles/ reproduce (the backward runs GEMMs on two
                streams at once: the FLOP / union-of-intervals figure is reported next to it).
                `frac` divides ALL useful flops by the fp32 MFMA peak, which flatters the launches that run as
                bf16x3 / fp16x2 splits on the 16-bit pipe; `frac_own_pipe` prices every launch against the pipe it
                ran on (157.3; 2382 / 6 = 397; 2382 / 3 = 794 fp32-equivalent TFLOP/s) = sum of the ideal launch
                times / sum of the measured ones, and `pipes` lists the three groups
  extra_configs the non-headline BASELINE configurations on this GPU (configs[2]: GGNN on
                ZINC-shaped graphs B=1000; configs[4] per-GPU work: AttentionGGNN on ChEMBL-shaped
                graphs B=250), a few timed steps each
  aggregation   the segmented-sum kernel: GB/s inside the training step (L2/MALL resident at this
                batch size) and on a 128x replicated graph batch whose message rows exceed the
                256 MB Infinity Cache, against 8 TB/s HBM3E
  forward_only  no_grad forward rate on the same batches
  generation_loop  forward + sampling step (softmax + draw + decode + validity: gi_sample_actions) per round on
                ONE pair of input tensors rewritten in place between rounds — the GraphGenerator.build_graphs
                pattern (GraphGenerator.py:118-157), which cannot prefetch graph_compact: host-sync-free
                forward (model.sync_free) against the forward with the blocking 24-int read-back
  loader_inclusive  the same step fed from host memory through BlockStreamLoader (PCIe-inclusive rate; N == 1 only)
  cpu_baseline  the unmodified reference model when a checkout is visible (kind "reference"), else the oracle
                (CPU restatement of the reference algorithm, kind "port", + the committed port/reference ratio) on the host
                cores at their best thread count, same workload, bounded sample; N == 1 only
config.fuse_flags = the GI_FUSE launch-count reductions in use (include/graphinvent_amd.h, default 15).
config.host = what the timed loop cost the HOST per step: enqueue_ms_per_step (Python + torch + the library's launches)
and wait_for_device_ms_per_step (blocked on the prefetched counts, i.e. the device is the bound when this is > 0); the
counts of the next TWO batches are prefetched, so the host may run two steps ahead of the device.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from graphinvent_amd import dp, lib, ops, synthetic          # noqa: E402
from graphinvent_amd.optim import FusedAdam                  # noqa: E402
from graphinvent_amd.gnn import mpnn                         # noqa: E402
from graphinvent_amd.loss import apd_kl_loss                 # noqa: E402

BATCH = 1000
N_BATCHES = 4                  # distinct resident minibatches cycled through
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# fp32-EQUIVALENT peaks of the split launches: the 16-bit MFMA's dense peak (MI355X_MICROARCH.md: 2 382 TFLOP/s for
# v_mfma_f32_32x32x16_bf16 / _f16) over the products one fp32 product costs (6 bf16, 3 f16)
PEAK_16BIT_MFMA_TFLOPS = 2382.0
PIPES = (("fp32_mfma", PEAK_FP32_MFMA_TFLOPS), ("bf16x3", PEAK_16BIT_MFMA_TFLOPS / 6), ("fp16x2", PEAK_16BIT_MFMA_TFLOPS / 3))
PROFILE_DIR = "r06"            # profiles/<dir>/: rocprofv3 kernel stats + PMC passes of this command
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)


SHAPE = "gdb13"                # --shape zinc runs BASELINE configs[2] (not the headline metric)
MODEL = "ggnn"                 # --model attggnn --shape chembl --batch 250 = configs[4]'s per-GPU work


def workload_constants(device: str, shape: str = None):
    from collections import namedtuple
    shape = shape or SHAPE
    sh = synthetic.SHAPES[shape]
    na, nc, N, Fe = sh["n_atom_types"], sh["n_formal_charge"], sh["max_n_nodes"], sh["n_edge_features"]
    cfg = dict(  # parameters/defaults.py:280-300 with hidden/message 128 (BASELINE configs[1])
        device=device, big_positive=1e6, big_negative=-1e6, n_node_features=na + nc,
        n_edge_features=Fe, max_n_nodes=N, len_f_add_per_node=na * nc * Fe, len_f_conn_per_node=Fe,
        hidden_node_features=128 if shape == "gdb13" else 100,
        message_size=128 if shape == "gdb13" else 100, message_passes=3, enn_depth=4,
        enn_hidden_dim=250, enn_dropout_p=0.0, gather_width=100, gather_att_depth=4,
        gather_att_hidden_dim=250, gather_att_dropout_p=0.0, gather_emb_depth=4,
        gather_emb_hidden_dim=250, gather_emb_dropout_p=0.0, mlp1_depth=4, mlp1_hidden_dim=500,
        mlp1_dropout_p=0.0, mlp2_depth=4, mlp2_hidden_dim=500, mlp2_dropout_p=0.0,
        # AttGGNN only, parameters/defaults.py:340-363
        msg_depth=4, msg_hidden_dim=250, msg_dropout_p=0.0, att_depth=4, att_hidden_dim=250,
        att_dropout_p=0.0)
    return cfg, namedtuple("CONSTANTS", sorted(cfg))(**cfg)


def make_batches(rank: int, device, shape: str = None, batch: int = None, n_batches: int = N_BATCHES):
    sh = synthetic.SHAPES[shape or SHAPE]
    out = []
    for i in range(n_batches):
        n8, e8, a8 = synthetic.make_batch(batch or BATCH, **sh, seed=1000 * rank + i)
        out.append(tuple(torch.from_numpy(x).float().to(device) for x in (n8, e8, a8)))
    return out


class Workload:
    """model + optimizer + scheduler + data-parallel trainer + resident synthetic minibatches of one
    configuration; `run_step()` is one training step in the order of Workflow.py:785-796."""

    def __init__(self, shape, model_name, batch, rank, device, total_steps, prefetch=True,
                 n_batches=N_BATCHES):
        self.shape, self.model_name, self.batch, self.prefetch = shape, model_name, batch, prefetch
        self.cfg, constants = workload_constants("cuda", shape)
        torch.manual_seed(0)                               # identical initial weights on every rank
        cls = mpnn.GGNN if model_name == "ggnn" else mpnn.AttentionGGNN
        self.model = cls(constants).to(device).train()
        self.batches = make_batches(rank, device, shape, batch, n_batches)
        self.opt = FusedAdam(self.model.parameters(), lr=1e-4)   # defaults.py:120 init_lr; one launch/step
        self.sched = torch.optim.lr_scheduler.OneCycleLR(self.opt, max_lr=1e-4,
                                                         total_steps=total_steps + 1)
        self.trainer = dp.DataParallel(self.model, self.opt, self.sched, loss_fn=apd_kl_loss)
        self.trainer.broadcast_parameters()
        self.i = 0
        self.prefetched_upto = 0

    # Like the block loader (graphinvent_amd/loader.py), the loop hands the NEXT batch to
    # ops.prefetch_compact before stepping on the current one: the counting phase of graph_compact
    # (and its host read-back) for batch k+1 runs on a side stream during step k.  Every step still
    # compacts its own batch — the result is consumed once, nothing is cached across steps.
    # PREFETCH_AHEAD = 2: the count of batch k needs the device to have finished step k - 3 (its prefetch was enqueued
    # behind step k - 3's launches), so the host may run up to two steps ahead of the device and a host hiccup of a
    # millisecond or two (a shared box) does not drain the queue; with 1 it could be one step ahead (1.0-1.4 ms of host
    # work per 1.85-ms step: profiles/r06/README.md).
    PREFETCH_AHEAD = int(os.environ.get("GI_BENCH_PREFETCH_AHEAD", "2"))

    def run_step(self):
        nb = len(self.batches)
        if self.prefetch:
            # (a batch's pending count is keyed by the batch: with nb resident batches at most nb - 1 can be pending)
            for a in range(1, min(self.PREFETCH_AHEAD, nb - 1) + 1):
                if self.i + a > self.prefetched_upto:
                    nxt = self.batches[(self.i + a) % nb]
                    ops.prefetch_compact(nxt[0], nxt[1])
                    self.prefetched_upto = self.i + a
        loss = self.trainer.step(*self.batches[self.i % nb])
        self.i += 1
        return loss


def timed_steps(wl: Workload, steps: int, warmup: int, world: int, device):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; returns
    (seconds: max over ranks, last loss)."""
    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(warmup):
        wl.run_step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    rb0 = dict(ops.READBACKS)
    wait0 = ops.HOST_WAIT[0]
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = wl.run_step()
    t_enq = time.perf_counter() - t0                               # the host has enqueued everything
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    wl.readbacks = {k: ops.READBACKS[k] - rb0[k] for k in rb0}     # of the timed steps' forwards
    # host side of the loop: what it costs to enqueue a step (the loop's wall time up to its last enqueue minus the time
    # spent waiting for the device at the prefetched counts) — the margin by which the host stays ahead of the device
    wl.host = {"enqueue_ms_per_step": round((t_enq - (ops.HOST_WAIT[0] - wait0)) / steps * 1e3, 4),
               "wait_for_device_ms_per_step": round((ops.HOST_WAIT[0] - wait0) / steps * 1e3, 4)}
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_val = float(loss)
    if not np.isfinite(loss_val):
        raise SystemExit("non-finite loss in the timed region")
    return dt, loss_val


def workload_text(shape, model_name, cfg):
    name = {"ggnn": "GGNN", "attggnn": "AttentionGGNN"}[model_name]
    return (f"{name} hidden={cfg['hidden_node_features']} message={cfg['message_size']} 3 MP steps, "
            f"{shape}-shaped synthetic graphs max_n_nodes={synthetic.SHAPES[shape]['max_n_nodes']}, "
            "train step fwd+KL+bwd+allreduce+Adam")


EXTRA_CONFIGS = (  # (BASELINE.json entry, shape, model, graphs per GPU per step)
    ("configs[2]: GGNN on ZINC-250k-shaped graphs, batch=1000", "zinc", "ggnn", 1000),
    ("configs[4] per-GPU work: AttGGNN on ChEMBL-shaped graphs max_n_nodes=88", "chembl", "attggnn", 250),
    ("NOT a BASELINE config: the headline model and graphs at batch=4000 (where the launches fill the device)",
     "gdb13", "ggnn", 4000),
)


def seg_sum_hbm_probe(device, M=128, replicas=128):
    """seg_sum (the aggregation a_v = sum of incoming messages) on a graph batch replicated until its
    message rows exceed the 256 MB Infinity Cache.  Bytes counted = what must cross HBM once: every
    message row once (rows read by several edges are re-read from cache), the index, the offsets and
    the output — i.e. the SURVEY.md §8d form with the de-duplicated row count."""
    sh = synthetic.SHAPES["gdb13"]
    n8, e8, _ = synthetic.make_batch(1000, **sh, seed=77)
    g, _ = ops.compact(torch.from_numpy(n8).float().to(device),
                       torch.from_numpy(e8).float().to(device), M)
    S, E, U = g.S, g.E, g.U
    perm = torch.cat([g.in_perm + r * U for r in range(replicas)])
    seg = g.seg_off[:S + 1]
    off = torch.cat([seg[:-1] + r * E for r in range(replicas)] +
                    [torch.tensor([replicas * E, replicas * E], dtype=torch.int32, device=device)])
    rows, nnz, mrows = replicas * S, replicas * E, replicas * U
    vals = torch.randn(mrows, M, device=device)
    out = torch.empty(rows + 1, M, device=device)
    for _ in range(3):
        ops.seg_sum(vals, perm, off, rows, M, out)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    ev0.record()
    for _ in range(reps):
        ops.seg_sum(vals, perm, off, rows, M, out)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    nbytes = mrows * M * 4 + nnz * 4 + (rows + 1) * 4 + rows * M * 4
    return dict(rows=rows, nnz=nnz, message_rows=mrows, working_set_MB=round(mrows * M * 4 / 1e6, 1),
                us=round(ms * 1e3, 2), GBps=round(nbytes / ms / 1e6, 1))


def generation_loop(model, cfg, batches, device, rounds: int):
    """One generation round = forward of the whole batch + one sampling launch, with the inputs MUTATED IN PLACE
    between rounds (the generator keeps one `nodes` / `edges` pair and edits it, GraphGenerator.py:118-157), so
    graph_compact cannot run one batch ahead.  Timed twice on the same rounds: model.sync_free (no read-back,
    sizes on the device) and the ordinary forward (blocking read-back of graph_compact's 24 ints per round)."""
    from graphinvent_amd.sampler import sample_actions_raw
    nodes = batches[0][0].clone()
    edges = batches[0][1].clone()
    A = cfg["len_f_add_per_node"]
    out = {}
    model.eval()
    with torch.no_grad():
        for mode in ("sync_free", "blocking_readback"):
            model.sync_free = mode == "sync_free"
            rb0 = dict(ops.READBACKS)
            for i in range(rounds + 3):
                if i == 3:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                src = batches[i % len(batches)]
                nodes.copy_(src[0]); edges.copy_(src[1])          # in place: the generator's own tensors
                logits = model(nodes, edges)
                n_nodes = (nodes.sum(2) != 0).sum(1).int()
                sample_actions_raw(logits, n_nodes, edges, A)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            err = model.last_bounded_error() if model.sync_free else 0
            out[mode] = {"ms_per_round": round(dt / rounds * 1e3, 3),
                         "graphs_per_s": round(nodes.shape[0] * rounds / dt, 1),
                         "readbacks": {k: ops.READBACKS[k] - rb0[k] for k in rb0}, "input_error_bits": err}
        # the same round recorded ONCE as a hipGraph (possible because nothing in it waits for the device) and
        # replayed: the host cost of a round drops to one graph launch
        try:
            model.sync_free = True
            src_n, src_e = batches[0][0], batches[0][1]
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                n_nodes = (nodes.sum(2) != 0).sum(1).int()
                sample_actions_raw(model(nodes, edges), n_nodes, edges, A)       # warm-up on the capture stream
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                logits = model(nodes, edges)
                n_nodes = (nodes.sum(2) != 0).sum(1).int()
                action, like, flags = sample_actions_raw(logits, n_nodes, edges, A)
            for i in range(rounds + 3):
                if i == 3:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                src = batches[i % len(batches)]
                nodes.copy_(src[0]); edges.copy_(src[1])
                g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["sync_free_hipgraph"] = {"ms_per_round": round(dt / rounds * 1e3, 3),
                                         "graphs_per_s": round(nodes.shape[0] * rounds / dt, 1),
                                         "finite": bool(torch.isfinite(logits).all())}
        except Exception as e:                                   # capture support differs between ROCm builds
            out["sync_free_hipgraph"] = {"error": repr(e)[:200]}
    model.sync_free = False
    model.train()
    out["rounds"] = rounds
    out["note"] = ("forward (no_grad) + gi_sample_actions per round, B=%d, inputs rewritten in place between "
                   "rounds; this rank only.  sync_free: no read-back, buffers sized for 4 B N directed edges "
                   "(1.8x / 4x the real node / edge rows of these batches)" % nodes.shape[0])
    return out


def loader_inclusive(trainer, device, n_graphs: int = 16000, block: int = 8000):
    """The same training step fed from HOST memory through the replacement of the reference's input pipeline
    (BlockDatasetLoader.py:32-63 -> graphinvent_amd.loader.BlockStreamLoader: int8 blocks into pinned memory on a
    background thread, vectorised row gather, async H2D one batch ahead, graph compaction of that batch on the copy
    stream, int8 straight into the model and the fused loss): graphs/s over one epoch of a synthetic int8 dataset
    in host memory.  NOT `value` (whose inputs are resident in HBM): the PCIe-inclusive rate next to it."""
    from graphinvent_amd.loader import ArraySource, BlockStreamLoader
    sh = synthetic.SHAPES["gdb13"]
    parts = [synthetic.make_batch(4000, **sh, seed=100 + s) for s in range(n_graphs // 4000)]
    nodes, edges, apds = (np.concatenate([p[i] for p in parts]) for i in range(3))
    loader = BlockStreamLoader(ArraySource(nodes, edges, apds), BATCH, block_size=block, seed=0, device=device)
    rates = []
    for epoch in range(2):                                     # (epoch 0 warms the pinned buffers and the thread up)
        loader.set_epoch(epoch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 0
        for nb, eb, ab in loader:
            trainer.step(nb, eb, ab)
            n += nb.shape[0]
        torch.cuda.synchronize()
        rates.append((n, time.perf_counter() - t0))
    n, dt = rates[-1]
    return {"value": round(n / dt, 1), "unit": "graphs/s", "ms_per_step": round(dt / (n / BATCH) * 1e3, 3),
            "graphs": n, "bytes_per_graph": int((nodes.nbytes + edges.nbytes + apds.nbytes) / n_graphs),
            "pinned_MB": round(loader.pinned_bytes / 1e6, 1),
            "note": "one epoch of a 16 000-graph int8 dataset in host memory through BlockStreamLoader (blocks of 8 000 rows, "
                    "shuffled, H2D + compaction one batch ahead), same model / optimizer as the timed region; this rank only"}


def own_pipe(handle):
    """(frac_own_pipe, per-pipe table) of the last gi_prof_collect: every GEMM launch against the matrix pipe it ran on."""
    ms = (C.c_double * 3)(); work = (C.c_double * 3)(); n = (C.c_int * 3)()
    lib.check(handle.gi_prof_pipes(ms, work, n), "gi_prof_pipes")
    ideal = sum(work[i] / (peak * 1e12) for i, (_, peak) in enumerate(PIPES))          # seconds at peak
    total = sum(ms) * 1e-3
    table = {}
    for i, (name, peak) in enumerate(PIPES):
        tf = work[i] / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else 0.0
        table[name] = {"launches": n[i], "gflop": round(work[i] / 1e9, 2), "sum_of_launch_ms": round(ms[i], 3),
                       "achieved": round(tf, 2), "peak_fp32_equivalent": round(peak, 1), "frac": round(tf / peak, 4)}
    return (round(ideal / total, 4) if total > 0 else 0.0), table


def committed_traffic():
    """roofline.traffic: HBM-side bytes per launch of the GEMM family from the committed PMC passes of THIS
    command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction applied;
    tools/collect_profiles.sh -> profiles/<round>/traffic.json) — counters cannot be read from inside the
    process, so the figure is the committed one, null when the file is not there."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PROFILE_DIR, "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return {"traffic": int(t["hbm_side_bytes_per_launch"]),
                "traffic_note": f"bytes per launch, profiles/{PROFILE_DIR}/traffic.json: {t['source']}; "
                                f"{t['fetch_correction']}"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None,
                "traffic_note": f"not measurable inside bench.py; the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                f"passes of this command go to profiles/{PROFILE_DIR}/traffic.json "
                                "(tools/collect_profiles.sh)"}


def reference_checkout():
    """Directory of an UNMODIFIED reference checkout if one is visible ($GI_REFERENCE, else /root/reference — which
    exists in the build container only, never on the GPU box), else None."""
    for root in (os.environ.get("GI_REFERENCE"), "/root/reference"):
        if root and os.path.isfile(os.path.join(root, "graphinvent", "gnn", "mpnn.py")):
            return root
    return None


def port_over_reference():
    """The committed build-container measurement (tools/port_over_reference.py): oracle port / unmodified reference
    model, same cores, same batch — quoted next to a "port" baseline, which the GPU box cannot check itself."""
    path = os.path.join(ROOT, "profiles", PROFILE_DIR, "port_over_reference.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return {"port_over_reference": t["port_over_reference"],
                "port_over_reference_note": f"profiles/{PROFILE_DIR}/port_over_reference.json: build container, "
                                            f"{t['threads']} threads, B={t['batch']}: reference {t['reference_graphs_per_s']} "
                                            f"graphs/s, port {t['port_graphs_per_s']} graphs/s (tools/port_over_reference.py)"}
    except (OSError, KeyError, ValueError):
        return {"port_over_reference": None}


def cpu_baseline(cfg, threads: int = 0, n_timed: int = 3):
    """The reference algorithm on torch-CPU ops, timed on the host cores: the UNMODIFIED reference model
    (gnn.mpnn.GGNN of a visible checkout, kind "reference") when there is one, else the oracle port (kind "port",
    with the committed port / reference ratio of the build container).  threads = 0: the
    thread count is chosen by a one-step probe of 8 / 16 / 32 threads — more threads make this
    workload SLOWER on the 256-CPU box (16: ~1130 graphs/s, 64: ~540, 256: 6; the GEMMs are small),
    and the baseline should be the CPU's best."""
    from oracle import ggnn_oracle as O
    ocfg = {k: cfg[k] for k in O.GDB13_DEFAULTS}
    ocfg["device"] = "cpu"
    model = O.OracleGGNN(ocfg, seed=0)
    kind, extra = "port", port_over_reference()
    ref_root = reference_checkout()
    if ref_root:
        try:
            sys.path.insert(0, os.path.join(ref_root, "graphinvent"))
            import gnn.mpnn as ref_mpnn                        # the unmodified reference package (torch only)
            ref = ref_mpnn.GGNN(O.as_constants(ocfg))
            ref.load_state_dict({k: v.detach().clone() for k, v in model.named_oracle_params().items()})
            model, kind, extra = ref, "reference", {"reference_checkout": ref_root}
        except Exception as e:                                 # (an incomplete checkout: fall back to the port)
            extra = dict(extra, reference_import_error=repr(e)[:160])
        finally:
            sys.path.pop(0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    sh = synthetic.SHAPES["gdb13"]
    n8, e8, a8 = synthetic.make_batch(BATCH, **sh, seed=0)
    nodes, edges, tgt = (torch.from_numpy(x).float() for x in (n8, e8, a8))

    def one():
        out = model(nodes, edges)
        opt.zero_grad()
        loss = O.kl_loss(out, tgt)
        loss.backward()
        opt.step()

    def timed(n):
        t0 = time.perf_counter()
        for _ in range(n):
            one()
        return (time.perf_counter() - t0) / n

    probe = ""
    if threads <= 0:
        best = None
        for cand in (8, 16, 32):
            if cand > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(cand)
            one()                                            # warm-up at this thread count
            dt = timed(1)
            probe += f"{cand}: {BATCH / dt:.0f}  "
            if best is None or dt < best[1]:
                best = (cand, dt)
        threads = best[0]
        probe = f"; thread count picked by a 1-step probe (graphs/s at {probe.strip()})"
    torch.set_num_threads(threads)
    one()
    dt = timed(n_timed)
    return dict(value=round(BATCH / dt, 1), unit="graphs/s", cores=threads, kind=kind,
                sample=f"{n_timed} timed steps (+1 warm-up) of the same B={BATCH} GGNN "
                       f"training step (fwd+KL+bwd+Adam), "
                       + ("the unmodified reference gnn.mpnn.GGNN" if kind == "reference" else "torch-CPU oracle")
                       + f", {threads} threads of {os.cpu_count()} host CPUs{probe}", **extra)


def main():
    global SHAPE, MODEL, BATCH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline; 0 = pick the fastest of 8 / 16 / 32 by a probe")
    ap.add_argument("--shape", default="gdb13", choices=["gdb13", "zinc", "chembl"],
                    help="gdb13 = BASELINE configs[1] (the metric); zinc = configs[2]; chembl = the "
                         "graph shape of configs[4] — both for reference, not the headline")
    ap.add_argument("--model", default="ggnn", choices=["ggnn", "attggnn"],
                    help="attggnn = gnn.mpnn.AttentionGGNN (configs[4]'s model class), for reference")
    ap.add_argument("--no-one-stream", action="store_true",
                    help="skip the extra roofline leg with the weight-gradient side stream off")
    ap.add_argument("--one-stream", action="store_true",
                    help="measurement: the WHOLE run with the weight gradients on the main stream (every launch "
                         "alone on the device; tools/collect_traces.sh); not the product schedule")
    ap.add_argument("--no-prefetch-compact", action="store_true",
                    help="run graph_compact's counting phase inside the step instead of one batch ahead")
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the beyond-Infinity-Cache seg_sum probe (keeps kernel traces clean)")
    ap.add_argument("--probe-only", action="store_true",
                    help="run ONLY the beyond-Infinity-Cache seg_sum probe (for rocprofv3 --pmc passes "
                         "over the aggregation kernel) and print its JSON")
    ap.add_argument("--no-loader", action="store_true", help="skip the host-fed (loader_inclusive) leg")
    ap.add_argument("--no-forward-only", action="store_true",
                    help="skip the no_grad forward-rate leg (keeps kernel traces to training steps only)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the non-headline BASELINE configurations (extra_configs)")
    ap.add_argument("--batch", type=int, default=BATCH, help="graphs per GPU per step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path).  gloo + ranks sharing one GPU is "
                         "a control-flow smoke test of the multi-rank path on a 1-GPU box; its "
                         "numbers are not a measurement")
    args = ap.parse_args()
    SHAPE, MODEL, BATCH = args.shape, args.model, args.batch
    if args.one_stream:
        mpnn.WGRAD_SIDE_STREAM = False
        args.no_one_stream = True
    headline = SHAPE == "gdb13" and MODEL == "ggnn" and BATCH == 1000

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with python -m torch.distributed.run "
                         "--nproc-per-node N (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    lib.load()
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.probe_only:
        print(json.dumps({"aggregation_probe": seg_sum_hbm_probe(device)}), flush=True)
        return
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    total_steps = args.steps + args.warmup + 160          # (+ the profiled / one-stream / arithmetic-mode / loader legs)
    wl = Workload(SHAPE, MODEL, BATCH, rank, device, total_steps,
                  prefetch=not args.no_prefetch_compact)
    cfg, model, trainer, batches = wl.cfg, wl.model, wl.trainer, wl.batches
    dt, loss_val = timed_steps(wl, args.steps, args.warmup, world, device)
    first_readbacks = dict(wl.readbacks)
    first_host = dict(wl.host)
    if world > 1:
        # every rank stepped on different graphs: the weights can only still be identical if the
        # gradient exchange (incl. its overlap with the backward) delivered the same mean everywhere
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum()
        both = torch.stack([chk, -chk])
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        spread = float(both[0] + both[1])                      # max - min over ranks
        if not spread <= 1e-9 * max(abs(float(chk)), 1.0):
            raise SystemExit(f"ranks diverged: parameter checksum spread {spread}")

    result = {
        "metric": "training graphs/sec (GGNN, GDB-13 max_n_nodes=13)" if headline else
                  f"training graphs/sec ({MODEL}, {SHAPE} shape)",
        "value": round(BATCH * world * args.steps / dt, 1), "unit": "graphs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: GGNN hidden=128 message=128 3 MP steps, "
                                "GDB-13-shaped synthetic graphs max_n_nodes=13, train step "
                                "fwd+KL+bwd+allreduce+Adam") if headline else
                               "NOT the headline config: " + workload_text(SHAPE, MODEL, cfg),
                   "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                   "parallelism": f"dp{world}", "loss": round(loss_val, 5),
                   "allreduce": ("none (1 rank)" if world == 1 else
                                 "flat fp32 bucket, readout tail overlapped with the backward"
                                 if trainer.overlap else "flat fp32 bucket after the backward")},
    }
    result["config"]["fuse_flags"] = int(lib.load().gi_fuse_flags())     # GI_FUSE_* variants in use
    result["config"]["gemm_arithmetic"] = (
        "fp32 operands, fp32 accumulate everywhere; the node-level readout layers >= 192 wide (forward, dgrad and weight "
        "gradients) and the message stacks' chain launches, forward and dZ (roofline.pipes counts the launches) split every fp32 operand into "
        + ("two scaled fp16 values (three f16 MFMA products per fp32 product, power-of-two scale per tensor — per ROW for the forward chains' activations — from the "
           "tensor's largest magnitude)" if lib.load().gi_x2_enable(-1) else
           "three bf16 values (six bf16 MFMA products per fp32 product)")
        + ": max error against the fp64 product 3e-7 relative, the fp32 MFMA chain's own: 5e-7 "
          "(tests/test_kernels_gpu.py); everything else on v_mfma_f32_32x32x2_f32"
        if lib.load().gi_bf3_enable(-1) else "fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere")
    # fp16x2 dynamic-range guard (gi_graph.x2_guard): counted on the device during every warm-up and timed step
    gs = model.x2_guard_stats()
    # dZ rows the counted launches saw: every fp16x2 dgrad problem of the node-level stacks (hidden layers with an amax
    # producer on both sides) reads R compact rows per step
    try:
        _, _, _, S0, _, _, _, _ = ops.compact_count(batches[0][0], batches[0][1])
        n_x2_dgrad = sum(max(0, depth - 1) for depth, hid in ((cfg["gather_att_depth"], cfg["gather_att_hidden_dim"]),
                                                               (cfg["gather_emb_depth"], cfg["gather_emb_hidden_dim"]),
                                                               (cfg["mlp1_depth"], cfg["mlp1_hidden_dim"]),
                                                               (cfg["mlp1_depth"], cfg["mlp1_hidden_dim"])) if hid >= 192)
        seen = n_x2_dgrad * (S0 + 1) * (args.steps + args.warmup)
        gs["dgrad_rows_checked"] = seen
        gs["dgrad_rows_frac"] = round(gs["dgrad_rows"] / seen, 7) if seen else None
    except Exception:
        pass
    result["x2_guard"] = dict(gs, steps_observed=args.steps + args.warmup,
                              note="rows of a forward fp16x2 launch's activations / rows+columns of its weights more "
                                   "than 2^24 below the tensor's largest magnitude (fewer than ~14 bits left): any "
                                   "such line trips the model into bf16x3 splits from the next forward on; "
                                   "dgrad_rows is informational (the backward's outputs are sums over rows)")
    # graph_compact's sizes: found on the host (counting phase one batch ahead) / read back behind the stream
    result["config"]["compact_readbacks_timed_steps"] = first_readbacks
    result["config"]["host"] = first_host
    if args.backend != "nccl":
        result["config"]["backend"] = args.backend + " (control-flow smoke test, not a measurement)"

    # ---- N > 1: what the exchange step costs (all ranks run the same extra steps) ----------------
    if world > 1:
        import socket
        seen = [None] * world
        dist.all_gather_object(seen, f"{socket.gethostname()}:cuda{local_rank}:"
                                     f"{torch.cuda.get_device_name(local_rank)}")
        k = 8

        def ms_per_step_now():
            d, _ = timed_steps(wl, k, 2, world, device)
            return d / k * 1e3
        t_overlap = ms_per_step_now()
        had_overlap = trainer.overlap
        trainer.overlap = False
        t_serial = ms_per_step_now()                           # one all-reduce after the backward
        real_world, trainer.world_size = trainer.world_size, 1
        t_none = ms_per_step_now()                             # no exchange at all (ranks diverge: last)
        trainer.world_size, trainer.overlap = real_world, had_overlap
        result["allreduce"] = {
            "backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "ranks": seen,
            "bucket_MB": round(sum(p.numel() for p in model.parameters()) * 4 / 1e6, 2),
            "ms_per_step_overlapped": round(t_overlap, 3), "ms_per_step_after_backward": round(t_serial, 3),
            "ms_per_step_without_exchange": round(t_none, 3),
            "exposed_ms_overlapped": round(t_overlap - t_none, 3),
            "exposed_ms_after_backward": round(t_serial - t_none, 3),
            "method": f"{k} timed steps each (max over ranks): product path (early exchange of the readout "
                      "tail overlapped with the message passes' backward); DataParallel(overlap=False): one "
                      "all-reduce after the backward; exchange skipped"}
        trainer.broadcast_parameters()                         # re-align the ranks for what follows

    # ---- roofline leg: per-launch HIP-event timing of the GEMM family + seg_sum -----------------
    # Every rank runs the extra steps (they contain the gradient all-reduce, a collective); only
    # rank 0 records and reports.
    handle = lib.load()
    prof_steps = 3
    torch.cuda.synchronize()
    if rank == 0:
        handle.gi_prof_enable(1)
    seg_bytes = 0.0
    for i in range(prof_steps):
        b = batches[i % N_BATCHES]
        if rank == 0:
            _, _, _, S, E, U, D0, _ = ops.compact_count(b[0], b[1])
            R, Mm, H, P = S + 1, cfg["message_size"], cfg["hidden_node_features"], cfg["message_passes"]
            # algorithmic bytes of the segmented-sum LAUNCHES of this step (SURVEY.md §8d form: values
            # read through the index + index + offsets + output).  Pass 0 aggregates through the
            # edge-count matrix (a GEMM) when the pass-0 rows are on; the d h scatter that GI_FUSE folds
            # into the gate backward is no launch of its own.
            fuse = handle.gi_fuse_flags()
            Pm = P - 1 if D0 > 0 else P                       # passes that run on message rows
            seg_bytes += Pm * (E * Mm * 4 + E * 4 + (R + 1) * 4 + R * Mm * 4)             # aggregation
            seg_bytes += Pm * (E * Mm * 4 + E * 4 + (U + 1) * 4 + 2 * U * Mm * 4)         # its backward
            if not (fuse & lib.FUSE_DH_SCATTER):
                seg_bytes += (P - 1) * (U * H * 4 + U * 4 + (R + 1) * 4 + 2 * R * H * 4)  # d h scatter
        trainer.step(*b)
    torch.cuda.synchronize()
    if rank == 0:
        ms = (C.c_double * 2)(); busy = (C.c_double * 2)(); work = (C.c_double * 2)(); n = (C.c_int * 2)()
        lib.check(handle.gi_prof_collect(ms, busy, work, n), "gi_prof_collect")
        handle.gi_prof_enable(0)
        # Headline: FLOP / SUM of the launch durations (= flop_per_launch / avg_launch_us), the
        # per-launch figure that `rocprofv3 --kernel-trace --stats` of this command reproduces
        # (profiles/<round>/).  The backward runs its weight-gradient GEMMs on a second stream
        # concurrently with the dZ chain, so FLOP / union-of-intervals is higher; reported next to it.
        tf_union = work[0] / (busy[0] * 1e-3) / 1e12 if busy[0] > 0 else 0.0
        tf = work[0] / (ms[0] * 1e-3) / 1e12 if ms[0] > 0 else 0.0
        own, pipes = own_pipe(handle)
        result["roofline"] = {
            "bound": "mfma", "kernel": "gi_gemm*/gi_chain*/gi_b3p* (fp32-accurate GEMM family: fp32 MFMA, bf16x3 and fp16x2 splits)",
            "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "frac_own_pipe": own, "pipes": pipes, **committed_traffic(),
            "launches_per_step": n[0] // prof_steps,
            "avg_launch_us": round(ms[0] * 1e3 / max(n[0], 1), 2),
            "flop_per_launch": round(work[0] / max(n[0], 1)),
            "gemm_sum_of_launch_ms_per_step": round(ms[0] / prof_steps, 3),
            "gemm_busy_ms_per_step": round(busy[0] / prof_steps, 3),
            "achieved_union_of_intervals": round(tf_union, 2),
            "frac_union_of_intervals": round(tf_union / PEAK_FP32_MFMA_TFLOPS, 4),
            "useful_gflop_per_step": round(work[0] / prof_steps / 1e9, 2),
            "step_useful_tflops": round(work[0] / prof_steps / 1e12 / (dt / args.steps), 2),
            "method": f"hipEvent pair around every launch on the stream it runs on, {prof_steps} extra "
                      "steps; achieved = useful FLOP / sum of the per-launch durations",
        }
        agg_gbs = seg_bytes / (busy[1] * 1e-3) / 1e9 if busy[1] > 0 else 0.0
        probe = dict(GBps=0.0, skipped=True) if args.no_probe else seg_sum_hbm_probe(device)
        result["aggregation"] = {
            "bound": "hbm", "kernel": "seg_sum_kernel", "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "in_step": {"achieved": round(agg_gbs, 1), "frac": round(agg_gbs / PEAK_HBM_GBS, 4),
                        "avg_launch_us": round(ms[1] * 1e3 / max(n[1], 1), 2),
                        "note": "working set ~10 MB: L2 / Infinity-Cache resident"},
            "beyond_infinity_cache": {"achieved": probe["GBps"],
                                      "frac": round(probe["GBps"] / PEAK_HBM_GBS, 4), **probe},
        }
    # The same per-launch figure with every launch ALONE on the device: the product path overlaps the
    # weight-gradient GEMMs (second stream) with the dZ chain, so its launch durations above include
    # the time two kernels share the CUs.  mpnn.WGRAD_SIDE_STREAM=False serialises the step (slower step,
    # shorter launches) — reported next to the headline, never instead of it.
    if not args.no_one_stream:
        mpnn.WGRAD_SIDE_STREAM = False
        try:
            k1 = 8
            d1, _ = timed_steps(wl, k1, 2, world, device)
            torch.cuda.synchronize()
            if rank == 0:
                handle.gi_prof_enable(1)
            for i in range(prof_steps):
                trainer.step(*batches[i % N_BATCHES])
            torch.cuda.synchronize()
            if rank == 0:
                ms1 = (C.c_double * 2)(); busy1 = (C.c_double * 2)(); work1 = (C.c_double * 2)(); n1 = (C.c_int * 2)()
                lib.check(handle.gi_prof_collect(ms1, busy1, work1, n1), "gi_prof_collect")
                handle.gi_prof_enable(0)
                tf1 = work1[0] / (ms1[0] * 1e-3) / 1e12 if ms1[0] > 0 else 0.0
                own1, pipes1 = own_pipe(handle)
                result["roofline"]["one_stream"] = {
                    "achieved": round(tf1, 2), "frac": round(tf1 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "frac_own_pipe": own1, "pipes": pipes1,
                    "avg_launch_us": round(ms1[0] * 1e3 / max(n1[0], 1), 2),
                    "ms_per_step": round(d1 / k1 * 1e3, 3),
                    "note": "mpnn.WGRAD_SIDE_STREAM=False: no two GEMM launches share the device; the step is "
                            "slower than the product path (ms_per_step above), the launches are shorter"}
        finally:
            mpnn.WGRAD_SIDE_STREAM = True
    # The same step with every GEMM on the fp32 MFMA (gi_bf3_enable(0)), and with the split launches as bf16x3 instead
    # of fp16x2 (gi_x2_enable(0)): by default the node-level hidden-layer launches of a step (forward, dgrad, weight
    # gradients, K = N = 250 / 500) run as fp16x2 splits on the 16-bit MFMA pipe — fp32 operands, fp32 accumulate, the
    # same result to ~3e-7 (tests/test_kernels_gpu.py) — reported next to the headline so that the effect of that
    # choice is visible, never instead of it.
    if not args.no_one_stream:
        was = handle.gi_bf3_enable(0)
        try:
            d0, _ = timed_steps(wl, 8, 2, world, device)
        finally:
            handle.gi_bf3_enable(was)
        if rank == 0:
            result["fp32_mfma_only"] = {"ms_per_step": round(d0 / 8 * 1e3, 3),
                                        "value": round(BATCH * world * 8 / d0, 1), "unit": "graphs/s",
                                        "note": "gi_bf3_enable(0) / GI_BF3=0: no launches on the 16-bit pipe"}
        if handle.gi_x2_enable(-1):
            handle.gi_x2_enable(0)
            try:
                d3, _ = timed_steps(wl, 8, 2, world, device)
            finally:
                handle.gi_x2_enable(1)
            if rank == 0:
                result["bf16x3_only"] = {"ms_per_step": round(d3 / 8 * 1e3, 3),
                                         "value": round(BATCH * world * 8 / d3, 1), "unit": "graphs/s",
                                         "note": "gi_x2_enable(0) / GI_X2=0: the split launches as three bf16 planes"}
    if rank == 0 and not args.no_forward_only:
        # forward-only rate: inference on the same resident batches, no_grad
        model.eval()
        with torch.no_grad():
            for i in range(3):
                model(*batches[i % N_BATCHES][:2])
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for i in range(args.steps):
                ops.prefetch_compact(*batches[(i + 1) % N_BATCHES][:2])
                model(*batches[i % N_BATCHES][:2])
            torch.cuda.synchronize()
            fdt = time.perf_counter() - tf0
        model.train()
        result["forward_only"] = {"value": round(BATCH * args.steps / fdt, 1), "unit": "graphs/s",
                                  "ms_per_step": round(fdt / args.steps * 1e3, 3),
                                  "note": "this rank only, no_grad forward of the same batches"}
    if rank == 0 and not args.no_forward_only:
        result["generation_loop"] = generation_loop(model, cfg, batches, device, rounds=max(args.steps, 10))
    if rank == 0 and world == 1 and headline and not args.no_loader and not args.no_forward_only:
        result["loader_inclusive"] = loader_inclusive(trainer, device)
    barrier()

    # ---- the non-headline BASELINE configurations, observed by the same run ----------------------
    if headline and not args.no_extra_configs:
        del wl, model, trainer, batches
        torch.cuda.empty_cache()
        extra = []
        for label, shape, model_name, batch in EXTRA_CONFIGS:
            k, w = 10, 3
            ewl = Workload(shape, model_name, batch, rank, device, k + w + 8, n_batches=2)
            edt, eloss = timed_steps(ewl, k, w, world, device)
            extra.append({"baseline_config": label,
                          "workload": workload_text(shape, model_name, ewl.cfg),
                          "batch_per_gpu": batch, "steps": k, "warmup": w,
                          "ms_per_step": round(edt / k * 1e3, 3),
                          "value": round(batch * world * k / edt, 1), "unit": "graphs/s",
                          "loss": round(eloss, 5)})
            del ewl
            torch.cuda.empty_cache()
        result["extra_configs"] = extra

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and headline:
            result["cpu_baseline"] = cpu_baseline(cfg, args.cpu_threads)
            result["speedup_vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        print(json.dumps(result), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
