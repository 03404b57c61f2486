"""-m gpu: the data-parallel step on the REAL HIP model with two ranks.  A 1-GPU box cannot run two
RCCL ranks (one rank per device), so there the two ranks share cuda:0 and talk through gloo: this
exercises everything the 8-GPU run does above the transport — per-rank batch slices, the two-call
backward, the early (overlapped) all-reduce of the readout tail of the flat gradient bucket on a
communication stream, the head exchange, FusedAdam on every rank — and checks it against one process
stepping on the global batch.  As soon as >= 2 GPUs are visible the SAME scenario also runs with
backend nccl (= RCCL over xGMI), one rank per device — the product transport; it is skipped, not
faked, on a 1-GPU box."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup(seed):
    from graphinvent_amd import synthetic
    from graphinvent_amd.gnn import mpnn
    from oracle import ggnn_oracle as O
    cfg = O.make_config(device="cuda")
    model = mpnn.GGNN(O.as_constants(cfg))
    model.load_state_dict(O.init_params(cfg, seed=seed))
    n8, e8, a8 = synthetic.make_batch(128, **synthetic.SHAPES["gdb13"], seed=3, frac_empty=0.0,
                                      frac_single=0.0)
    data = tuple(torch.from_numpy(x).float().cuda() for x in (n8, e8, a8))
    return model.cuda().train(), data


def _worker(rank, world, port, out_dir, overlap, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC (RCCL across processes)
    from graphinvent_amd import dp
    from graphinvent_amd.loss import apd_kl_loss
    from graphinvent_amd.optim import FusedAdam
    if backend == "nccl":                                       # RCCL: one rank per device
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    model, (nodes, edges, tgt) = _setup(seed=10 + rank)         # deliberately different per rank
    opt = FusedAdam(model.parameters(), lr=1e-3)
    tr = dp.DataParallel(model, opt, loss_fn=apd_kl_loss, overlap=overlap)
    tr.broadcast_parameters(src=0)
    sampler = dp.ShardedBatchSampler(128, 32, rank, world, seed=9)
    losses, flags = [], []
    for idx in sampler:                                         # 2 steps per rank, lock-step
        i = torch.from_numpy(idx).cuda()
        losses.append(float(tr.step(nodes[i], edges[i], tgt[i])))
        flags.append((tr.last_bucket_zero_copy, tr.last_overlapped))
    torch.save(dict(params=[p.detach().cpu() for p in model.parameters()], losses=losses,
                    flags=flags, idx=[b.tolist() for b in sampler]),
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_on_the_hip_model_equal_one_process_on_the_global_batch(tmp_path, overlap, backend):
    from graphinvent_amd.loss import apd_kl_loss
    from graphinvent_amd.optim import FusedAdam
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: < 2 GPUs visible")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), overlap, backend), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in (0, 1))
    for a, b in zip(r0["params"], r1["params"]):                # ranks stay bit-identical
        assert torch.equal(a, b)
    assert all(f == (True, overlap) for f in r0["flags"] + r1["flags"])
    model, (nodes, edges, tgt) = _setup(seed=10)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    for step in range(2):
        i = torch.tensor(r0["idx"][step] + r1["idx"][step]).cuda()
        out = model(nodes[i], edges[i])
        opt.zero_grad(set_to_none=True)
        apd_kl_loss(out, tgt[i]).backward()
        opt.step()
    # Adam turns a ~0 gradient into an O(lr) step of arbitrary sign: bound single elements by
    # 2 * lr * steps, and the bulk of every sizeable tensor tightly
    for (k, a), b in zip(model.named_parameters(), r0["params"]):
        diff = (a.detach().cpu() - b).double()
        assert float(diff.abs().max()) <= 2 * 1e-3 * 2, k
        if b.numel() >= 1000:
            assert float(diff.norm() / b.double().norm().clamp_min(1e-12)) < 5e-4, k


def _worker_replaced_grads(rank, world, port, out_dir):
    """A training loop that replaces param.grad between backward and the exchange while the early
    all-reduce is pending must fail loudly (it used to double-count the readout gradients)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from graphinvent_amd import dp
    from graphinvent_amd.loss import apd_kl_loss
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, (nodes, edges, tgt) = _setup(seed=10)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    tr = dp.DataParallel(model, opt, loss_fn=apd_kl_loss, overlap=True)
    out = model(nodes[:16], edges[:16])
    loss = apd_kl_loss(out, tgt[:16])
    model._grad_ready_hook = tr._early_allreduce
    loss.backward()
    assert model._grad_ready_hook is None and model._early_exchange_pending      # single shot
    for p in model.parameters():
        p.grad = p.grad.clone()                                  # no longer views of the bucket
    msg = ""
    try:
        tr.allreduce_gradients()
    except RuntimeError as e:
        msg = str(e)
    torch.save(dict(msg=msg), os.path.join(out_dir, f"e{rank}.pt"))
    dist.destroy_process_group()


def test_pending_early_exchange_with_replaced_gradients_fails_loudly(tmp_path):
    port = _free_port()
    mp.spawn(_worker_replaced_grads, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        assert "no longer views" in torch.load(tmp_path / f"e{r}.pt")["msg"]
