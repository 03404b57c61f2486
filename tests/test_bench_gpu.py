"""-m gpu: `bench.py --gpus N` as the driver launches it (python -m torch.distributed.run, one rank per GPU), so
that the first 8-GPU execution of the multi-rank branch is a re-run: the `allreduce` report, the cross-rank
weight checksum and the three extra timed legs that toggle the exchange.  On a 1-GPU box the two ranks share the
device over gloo (a control-flow test, not a measurement); with >= 2 devices the same scenario runs over nccl
(RCCL), one rank per device.  Reference counterpart: none (Workflow.py:289-290 trains in one process)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(world: int, backend: str, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", backend, "--steps", "3",
           "--warmup", "1", "--no-extra-configs", "--no-one-stream", "--no-cpu-baseline", "--no-probe",
           "--no-forward-only", *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]                # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _check(d, world: int, backend: str, batch: int = 1000):
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["unit"] == "graphs/s" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == batch * world and d["config"]["parallelism"] == f"dp{world}"
    assert d["value"] > 0 and abs(d["value"] - batch * world * 3 / (d["ms_per_step"] * 3e-3)) < 0.01 * d["value"]
    loss = d["config"]["loss"]
    assert loss == loss and 0 < loss < 100                   # finite (the run aborts on a diverged checksum)
    ar = d["allreduce"]
    assert ar["ranks_seen"] == world and len(ar["ranks"]) == world and ar["backend"] == backend
    for k in ("ms_per_step_overlapped", "ms_per_step_after_backward", "ms_per_step_without_exchange",
              "exposed_ms_overlapped", "exposed_ms_after_backward"):
        assert isinstance(ar[k], float) and ar[k] == ar[k], k
    assert ar["bucket_MB"] > 20
    rf = d["roofline"]
    assert rf["frac"] > 0 and rf["bound"] == "mfma"
    # every launch priced against the pipe it ran on: the three groups add up to the family, and the own-pipe
    # fraction can only be below the all-against-fp32 one when launches ran on the faster 16-bit pipe
    pipes = rf["pipes"]
    assert set(pipes) == {"fp32_mfma", "bf16x3", "fp16x2"}
    assert sum(p["launches"] for p in pipes.values()) == rf["launches_per_step"] * 3
    assert abs(sum(p["gflop"] for p in pipes.values()) - rf["useful_gflop_per_step"] * 3) < 0.02 * rf["useful_gflop_per_step"] * 3
    assert 0 < rf["frac_own_pipe"] <= rf["frac"] + 1e-4
    assert "overlapped" in d["config"]["allreduce"]
    # host side of the timed loop: what enqueuing a step costs, and how long the host waited for the device
    host = d["config"]["host"]
    assert host["enqueue_ms_per_step"] > 0 and host["wait_for_device_ms_per_step"] >= 0
    assert host["enqueue_ms_per_step"] + host["wait_for_device_ms_per_step"] <= d["ms_per_step"] * 1.05
    assert d["config"]["compact_readbacks_timed_steps"]["blocking"] == 0     # every count was prefetched


def test_bench_two_ranks_gloo_sharing_the_gpu():
    d = _run_bench(2, "gloo")
    _check(d, 2, "gloo")
    assert "smoke test" in d["config"]["backend"]


def test_bench_eight_ranks_gloo_sharing_the_gpu():
    """The world size the driver's scaling run ends at (`bench.py --gpus 8`), on ONE device over gloo with a small
    per-rank batch: eight ranks through the rendezvous, the weight broadcast, the overlapped two-call backward with
    the early exchange of the readout tail, FusedAdam on every rank and the cross-rank weight checksum (a diverged
    checksum aborts the run) — so that the first real 8-GPU execution (backend nccl = RCCL, one rank per device) only
    swaps the transport.  Not a measurement."""
    d = _run_bench(8, "gloo", extra=("--batch", "96"))
    _check(d, 8, "gloo", batch=96)
    assert sorted(d["allreduce"]["ranks"]) == sorted(set(d["allreduce"]["ranks"])) or len(d["allreduce"]["ranks"]) == 8


def test_scale_script_dry_run_over_gloo(tmp_path):
    """tools/scale.sh — what produces the N = 1, 2, 4, 8 lines on the first multi-GPU box — run end to end on ONE device
    with the ranks sharing it over gloo (SCALE_BACKEND=gloo, small batch, 3 steps): four JSON lines that parse, the
    rank counts right, `allreduce.exposed_ms_*` present for N > 1, and the summary lines the script prints.  A dry run
    of the script's control flow, not a measurement: scaling on RCCL / xGMI is UNMEASURED (README)."""
    out = tmp_path / "scale"
    env = dict(os.environ, SCALE_BACKEND="gloo", SCALE_PORT=str(_free_port() - 8), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale.sh"), str(out), "--batch", "96", "--steps", "3",
                        "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    summary = [ln for ln in r.stdout.splitlines() if ln.startswith("N=")]
    assert len(summary) == 4 and all("graphs/s" in ln for ln in summary), r.stdout[-2000:]
    for n in (1, 2, 4, 8):
        d = json.loads(open(out / f"scale_{n}.json").read())
        assert d["n_gpus"] == n and d["config"]["global_batch"] == 96 * n and d["value"] > 0
        if n > 1:
            ar = d["allreduce"]
            assert ar["ranks_seen"] == n and ar["backend"] == "gloo"
            assert all(isinstance(ar[k], float) for k in ("exposed_ms_overlapped", "exposed_ms_after_backward"))
            assert "exchange exposed" in summary[(1, 2, 4, 8).index(n)]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL, one rank per device)")
def test_bench_two_ranks_rccl():
    _check(_run_bench(2, "nccl"), 2, "nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 8, reason="needs 8 GPUs (RCCL over xGMI, one rank per device)")
def test_bench_eight_ranks_rccl():
    _check(_run_bench(8, "nccl"), 8, "nccl")
